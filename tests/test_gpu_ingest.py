"""FASTQ ingest on the device (cm_ingest.hip): chunks of raw FASTQ text are
split into lines and records in HBM and become the resident SoA batch.  The batch must equal what
the oracle's kseq-style reader produces from the same files, for any chunking of the text."""
import numpy as np
import pytest

import datasets
import oracle_lib as ol

pytestmark = pytest.mark.gpu


def _gpu():
    from chromap_amd import ChromapGPU
    fa, _, _ = datasets.case_inputs("toy_chip")
    return ChromapGPU(datasets.case_index("toy_chip"), fa, preset="chip")


def _ingest_stream(g, text, chunk, stream=0, limit=None):
    """feeds `text` in chunks of `chunk` bytes the way the CLI does; returns (bases, offsets)"""
    bases, lens = [], []
    pos, carry = 0, b""
    while True:
        piece = text[pos:pos + chunk]
        pos += len(piece)
        final = pos >= len(text)
        buf = carry + piece
        n = g.fastq_scan(stream, buf, final)
        if limit:
            n = min(n, limit)
        used = g.fastq_take(stream, n)
        if n:
            g.fastq_commit(n, paired=False)
            b1, o1, _, _ = g.download_batch(n)
            bases.append(b1.copy())
            lens.append(np.diff(o1))
        carry = buf[used:]
        if final and n == 0:
            assert carry.strip() == b""
            break
        if final and not carry.strip():
            break
    b = np.concatenate(bases) if bases else np.zeros(0, np.uint8)
    ln = np.concatenate(lens) if lens else np.zeros(0, np.uint32)
    off = np.zeros(len(ln) + 1, np.uint32)
    off[1:] = np.cumsum(ln)
    return b, off


@pytest.mark.parametrize("chunk,limit", [(1 << 30, None), (100003, None), (4096, None), (1 << 20, 777)])
def test_ingest_equals_host_reader(chunk, limit):
    g = _gpu()
    _, r1, _ = datasets.case_inputs("s2_atac_q0")  # variable read lengths
    text = open(r1, "rb").read()
    want_b, want_o = ol.read_fastx(r1)
    b, off = _ingest_stream(g, text, chunk, limit=limit)
    assert np.array_equal(off, want_o)
    assert np.array_equal(b, want_b)
    g.close()


def test_ingest_line_ending_variants(tmp_path):
    g = _gpu()
    recs = [(b"r%d extra comment" % i, b"ACGTN"[i % 5:i % 5 + 1] * (20 + i % 37), b"I" * (20 + i % 37)) for i in range(500)]
    recs[7] = (b"empty", b"", b"")          # skipped like kseq length 0 (sequence_batch.cc:27-30)
    recs[499] = (b"lastempty", b"", b"")
    def render(nl, trailing):
        t = b"".join(b"@" + n + nl + s + nl + b"+" + nl + q + nl for n, s, q in recs)
        return t[:-len(nl)] + trailing
    want = [s for _, s, _ in recs if s]
    for nl, trailing in ((b"\n", b"\n"), (b"\r\n", b"\r\n"), (b"\n", b""), (b"\n", b"\n\n\n")):
        b, off = _ingest_stream(g, render(nl, trailing), 1500)
        assert len(off) - 1 == len(want)
        assert b.tobytes() == b"".join(want)
    g.close()


def test_ingest_rejects_multiline_fastq():
    from chromap_amd import ChromapError
    g = _gpu()
    text = b"@a\nACGT\nACGT\n+\nIIIIIIII\n@b\nAC\n+\nII\n"
    with pytest.raises(ChromapError, match="4-line FASTQ"):
        g.fastq_scan(0, text, True)
    g.close()


def test_ingest_pairs_and_barcodes_map_like_host_buffers():
    """reads + barcodes ingested on the device give the same records as the host-buffer entry"""
    from chromap_amd import ChromapGPU
    case = "b1_atac_bc"
    meta = datasets.case_meta(case)
    fa, r1, r2 = datasets.case_inputs(case)
    bcf, wlf = datasets.case_barcode_inputs(case)
    preset, kw = datasets.flags_to_params(meta["chromap_flags"])
    g = ChromapGPU(datasets.case_index(case), fa, preset=preset, **kw)
    b1, o1 = ol.read_fastx(r1)
    b2, o2 = ol.read_fastx(r2)
    bc, bcq, bco = ol.read_fastq_qual(bcf)
    g.set_whitelist_file(wlf, int(bco[1] - bco[0]))
    g.compute_barcode_abundance(bc, bco)
    rec, k = g.map_pairs_barcoded(b1, o1, b2, o2, bc, bcq, bco)
    want = sorted((rec[i].r.read_id, rec[i].r.rid, rec[i].r.fragment_start, rec[i].r.fragment_length, rec[i].r.mapq, rec[i].barcode)
                  for i in range(k))
    ns = [g.fastq_scan(s, open(f, "rb").read(), True) for s, f in ((0, r1), (1, r2), (2, bcf))]
    assert ns == [len(o1) - 1] * 3
    for s in range(3):
        g.fastq_take(s, ns[0])
    g.fastq_commit(ns[0], paired=True, barcoded=True)
    from chromap_amd import Stats
    k2 = g.map_resident(Stats())
    assert k2 == k
    g.store_clear()
    assert g.store_append_resident() == k
    lines, _ = g.store_format(2, barcode_length=g.barcode_length)
    import hashlib
    assert hashlib.md5(g.store_text()).hexdigest() == meta["bed_md5"]
    assert len(want) == k
    g.close()


# ---- BGZF inflated on the device (cmgpu_fastq_scan_bgzf): the host hands over whole compressed blocks only ----
def _bgzf_blocks(text, level=1, block=0xff00):
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tools"))
    import bgzf
    return [bgzf._block(text[i:i + block], level) for i in range(0, len(text), block)] + [bgzf._block(b"", level)]


def _ingest_bgzf(g, blocks, per_call, stream=0, limit=None):
    """feeds the blocks `per_call` at a time; the device keeps what a take leaves; returns (bases, offsets)"""
    bases, lens = [], []
    at = 0
    while True:
        piece = b"".join(blocks[at:at + per_call])
        at += per_call
        final = at >= len(blocks)
        n = g.fastq_scan(stream, piece, final, bgzf=True)
        if limit:
            n = min(n, limit)
        g.fastq_take(stream, n)
        if n:
            g.fastq_commit(n, paired=False)
            b1, o1, _, _ = g.download_batch(n)
            bases.append(b1.copy())
            lens.append(np.diff(o1))
        if final and n == 0:
            break
    b = np.concatenate(bases) if bases else np.zeros(0, np.uint8)
    ln = np.concatenate(lens) if lens else np.zeros(0, np.uint32)
    off = np.zeros(len(ln) + 1, np.uint32)
    off[1:] = np.cumsum(ln)
    return b, off


@pytest.mark.parametrize("per_call,limit,level,block", [(1 << 20, None, 1, 0xff00), (7, None, 6, 0xff00), (1, None, 1, 5000), (50, 333, 9, 30011),
                                                        (3, None, 0, 0xff00)])
def test_bgzf_inflated_on_the_device_equals_host_reader(per_call, limit, level, block):
    g = _gpu()
    _, r1, _ = datasets.case_inputs("s2_atac_q0")
    text = open(r1, "rb").read()
    want_b, want_o = ol.read_fastx(r1)
    b, off = _ingest_bgzf(g, _bgzf_blocks(text, level, block), per_call, limit=limit)
    assert np.array_equal(off, want_o)
    assert np.array_equal(b, want_b)
    # a second file on the same stream starts clean (the blank rest of the first one is dropped)
    b, off = _ingest_bgzf(g, _bgzf_blocks(text + b"\n\n", level, block), per_call, limit=limit)
    assert np.array_equal(off, want_o) and np.array_equal(b, want_b)
    g.close()


def test_bgzf_file_ending_in_many_blank_lines():
    """more than 64 KiB of blank lines behind the last record (the host reader's only_whitespace() accepts any amount): accepted,
    counted as consumed, and the stream starts clean for the next file; text that is not blank there is refused"""
    from chromap_amd import ChromapError
    g = _gpu()
    _, r1, _ = datasets.case_inputs("s2_atac_q0")
    text = open(r1, "rb").read()
    want_b, want_o = ol.read_fastx(r1)
    for tail in (b"\n" * 200000, b" \r\n\t\n" * 40000):
        b, off = _ingest_bgzf(g, _bgzf_blocks(text + tail, 1), 1 << 20)
        assert np.array_equal(off, want_o) and np.array_equal(b, want_b)
    blocks = _bgzf_blocks(text + b"\n" * 100000 + b"x\n", 1)
    n = g.fastq_scan(0, b"".join(blocks), True, bgzf=True)
    with pytest.raises(ChromapError, match="after the last whole FASTQ record"):
        g.fastq_take(0, n)
    g.close()


def test_bgzf_damaged_blocks_are_rejected():
    import struct, zlib
    from chromap_amd import ChromapError
    g = _gpu()
    _, r1, _ = datasets.case_inputs("s2_atac_q0")
    text = open(r1, "rb").read()[:400000]
    text = text[:text.rindex(b"\n@") + 1]  # whole records
    blocks = _bgzf_blocks(text, 6)
    assert len(blocks) >= 5
    good = b"".join(blocks)
    assert g.fastq_scan(0, good, True, bgzf=True) > 0
    g.fastq_take(0, 0)

    def fresh():  # the text kept from the attempt before is dropped by a plain-text scan (leaves device mode)
        g.fastq_scan(0, b"@a\nAC\n+\nII\n", True)
        g.fastq_take(0, 1)

    def damaged(i, f):
        b = list(blocks)
        b[i] = f(bytearray(b[i]))
        return b"".join(bytes(x) for x in b)

    def flip_crc(b):
        b[-8] ^= 1
        return b

    def wrong_isize(b):
        b[-4:] = struct.pack("<I", struct.unpack("<I", bytes(b[-4:]))[0] - 1)
        return b

    def payload_bits(b):
        for k in range(40, 60):
            b[k] ^= 0xA5
        return b

    def cut_payload(b):  # a shorter payload under the same trailer: the header's block size follows
        nb = b[:18] + b[18:len(b) - 8 - 200] + b[-8:]
        nb[16:18] = struct.pack("<H", len(nb) - 1)
        return nb

    for i, f, what in ((2, flip_crc, "CRC mismatch"), (1, wrong_isize, "ISIZE|past the block|invalid code|CRC"), (3, payload_bits, "damaged BGZF block 3"),
                       (0, cut_payload, "damaged BGZF block 0")):
        fresh()
        with pytest.raises(ChromapError, match=what):
            g.fastq_scan(0, damaged(i, f), True, bgzf=True)
    # a chunk that ends inside a block, or does not start with one
    fresh()
    with pytest.raises(ChromapError, match="truncated BGZF block"):
        g.fastq_scan(0, good[:len(blocks[0]) + 100], True, bgzf=True)
    fresh()
    with pytest.raises(ChromapError, match="not a BGZF block"):
        g.fastq_scan(0, good[5:], True, bgzf=True)
    # text after the last whole record of the file
    fresh()
    junk = b"".join(_bgzf_blocks(b"@a\nACGT\n+\nIIII\n@b\nAC", 6))
    assert g.fastq_scan(0, junk, False, bgzf=True) == 1
    g.fastq_take(0, 1)
    fresh()
    g.fastq_scan(0, b"".join(_bgzf_blocks(b"@a\nACGT\n+\nIIII\nxx", 6)), True, bgzf=True)
    with pytest.raises(ChromapError, match="after the last whole FASTQ record"):
        g.fastq_take(0, 1)
    # the context still works
    fresh()
    b, off = _ingest_bgzf(g, blocks, 2)
    assert len(off) - 1 == text.count(b"\n") // 4
    g.close()


def test_next_batch_taken_while_the_last_is_mapped():
    """the CLI's overlap (round 6): cmgpu_fastq_take gathers into staging buffers on the file's own stream, cmgpu_fastq_commit swaps them
    in -- so a batch may be scanned and taken while another thread maps the committed one.  The records of both batches equal those of the
    serial order."""
    import threading
    from chromap_amd import ChromapGPU, Stats
    case = "s1_atac"
    meta = datasets.case_meta(case)
    fa, r1, r2 = datasets.case_inputs(case)
    preset, kw = datasets.flags_to_params(meta["chromap_flags"])
    t1, t2 = open(r1, "rb").read(), open(r2, "rb").read()
    n_all = t1.count(b"\n") // 4
    half = n_all // 2

    def cut(t, n):  # the first n records / the rest
        pos = 0
        for _ in range(4 * n):
            pos = t.index(b"\n", pos) + 1
        return t[:pos], t[pos:]
    a1, b1 = cut(t1, half)
    a2, b2 = cut(t2, half)

    def records(g):
        k = g.store_append_resident()
        return k

    # serial order
    g = ChromapGPU(datasets.case_index(case), fa, preset=preset, **kw)
    g.store_clear()
    for (x1, x2, first) in ((a1, a2, 0), (b1, b2, half)):
        n = g.fastq_scan(0, x1, True)
        assert g.fastq_scan(1, x2, True) == n
        g.fastq_take(0, n); g.fastq_take(1, n)
        g.fastq_commit(n, first_read_id=first, paired=True)
        g.map_resident(Stats())
        g.store_append_resident()
    g.store_format()
    want = g.store_text()
    g.close()

    # the second batch scanned and taken while the first is being mapped
    g = ChromapGPU(datasets.case_index(case), fa, preset=preset, **kw)
    g.store_clear()
    n = g.fastq_scan(0, a1, True)
    assert g.fastq_scan(1, a2, True) == n
    g.fastq_take(0, n); g.fastq_take(1, n)
    g.fastq_commit(n, first_read_id=0, paired=True)
    err = []

    def work():
        try:
            for _ in range(3):  # (several times over: the mapping must not see the staging buffers change under it)
                g.map_resident(Stats())
            g.store_append_resident()
        except Exception as e:  # noqa: BLE001
            err.append(e)
    th = threading.Thread(target=work)
    th.start()
    m = g.fastq_scan(0, b1, True)
    assert g.fastq_scan(1, b2, True) == m
    g.fastq_take(0, m); g.fastq_take(1, m)
    th.join()
    assert not err, err
    g.fastq_commit(m, first_read_id=half, paired=True)
    g.map_resident(Stats())
    g.store_append_resident()
    g.store_format()
    got = g.store_text()
    g.close()
    assert got == want and len(want) > 1000


def test_warm_up_leaves_results_alone():
    """cmgpu_warm_up (a dry run of the whole path in a context of its own): callable more than once, before or after contexts exist;
    a golden case maps to its BED afterwards"""
    import hashlib
    from chromap_amd import ChromapGPU, Stats, lib
    L = lib()
    assert L.cmgpu_warm_up(0) == 0
    assert L.cmgpu_warm_up(0) == 0
    assert L.cmgpu_warm_up(99) != 0  # no such device
    case = "s1_atac"
    meta = datasets.case_meta(case)
    fa, r1, r2 = datasets.case_inputs(case)
    preset, kw = datasets.flags_to_params(meta["chromap_flags"])
    g = ChromapGPU(datasets.case_index(case), fa, preset=preset, **kw)
    n = g.fastq_scan(0, open(r1, "rb").read(), True)
    assert g.fastq_scan(1, open(r2, "rb").read(), True) == n
    g.fastq_take(0, n); g.fastq_take(1, n)
    g.fastq_commit(n, paired=True)
    g.map_resident(Stats())
    g.store_clear()
    g.store_append_resident()
    g.store_format()
    assert hashlib.md5(g.store_text()).hexdigest() == meta["bed_md5"]
    g.close()
