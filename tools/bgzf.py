#!/usr/bin/env python3
"""Writes BGZF (bgzip's format, SAM specification 4.1): gzip members of at most 64 KiB of input each, the compressed
block size in a 'BC' extra field, an empty block at the end.  There is no bgzip in the image; this is what the tests
and tools/e2e_bench.py use to make block-compressed FASTQ for the block-parallel inflate of chromap-amd."""
import struct
import sys
import zlib
from concurrent.futures import ThreadPoolExecutor

BLOCK = 0xff00  # input bytes per block (bgzip's default)


def _block(data, level):
    c = zlib.compressobj(level, zlib.DEFLATED, -15)
    body = c.compress(data) + c.flush()
    bsize = len(body) + 25  # total block size - 1
    assert bsize < 65536
    hdr = b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize)
    return hdr + body + struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data))


def compress_file(src, dst, level=1, threads=16):
    with open(src, "rb") as f, open(dst, "wb") as out, ThreadPoolExecutor(threads) as ex:
        while True:
            slab = f.read(BLOCK * 4096)
            if not slab:
                break
            parts = [slab[i:i + BLOCK] for i in range(0, len(slab), BLOCK)]
            for blk in ex.map(lambda d: _block(d, level), parts):  # zlib releases the GIL
                out.write(blk)
        out.write(_block(b"", level))


if __name__ == "__main__":
    compress_file(sys.argv[1], sys.argv[2])
