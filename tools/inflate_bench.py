#!/usr/bin/env python3
"""Times the device BGZF inflate (cmgpu_fastq_scan_bgzf: k_bgzf_tokens + k_bgzf_resolve, then the FASTQ scan) against the plain-text
scan of the same FASTQ text, and checks that both give the same batch.  Synthetic reads: random bases, qualities in runs (what
makes real FASTQ compress), fixed-width names.  Prints one JSON line.  Under rocprofv3 --kernel-trace --stats the two kernels'
durations are the numbers DESIGN.md section 12 quotes."""
import argparse
import json
import os
import sys
import time
import zlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bgzf  # noqa: E402


def fastq_text(n, readlen, seed):
    rng = np.random.default_rng(seed)
    name = np.frombuffer(b"@SRR0000000.", np.uint8)
    w = len(name) + 9 + 1 + readlen + 1 + 2 + readlen + 1
    rec = np.zeros((n, w), np.uint8)
    rec[:, :len(name)] = name
    ids = np.arange(n)
    for k in range(9):
        rec[:, len(name) + 8 - k] = 48 + (ids // 10 ** k) % 10
    at = len(name) + 9
    rec[:, at] = 10
    rec[:, at + 1:at + 1 + readlen] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, (n, readlen))]
    at += 1 + readlen
    rec[:, at] = 10
    rec[:, at + 1] = ord("+")
    rec[:, at + 2] = 10
    q = np.frombuffer(b"FFFF:F,F", np.uint8)[rng.integers(0, 8, (n, (readlen + 7) // 8))]
    rec[:, at + 3:at + 3 + readlen] = np.repeat(q, 8, axis=1)[:, :readlen]
    rec[:, at + 3 + readlen] = 10
    return rec.tobytes()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=4_000_000)
    ap.add_argument("--readlen", type=int, default=50)
    ap.add_argument("--level", type=int, default=1)
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    text = fastq_text(a.reads, a.readlen, 7)
    parts = [text[i:i + bgzf.BLOCK] for i in range(0, len(text), bgzf.BLOCK)]
    with ThreadPoolExecutor(32) as ex:
        blocks = b"".join(ex.map(lambda d: bgzf._block(d, a.level), parts)) + bgzf._block(b"", a.level)
    from chromap_amd import ChromapGPU
    g = ChromapGPU(synthetic=(2_000_000, 2, 5), preset="atac")
    res = {"reads": a.reads, "text_bytes": len(text), "bgzf_bytes": len(blocks), "blocks": len(parts), "level": a.level}
    for name, buf, kw in (("plain", text, {}), ("bgzf", blocks, {"bgzf": True})):
        ts = []
        for _ in range(a.reps):
            t0 = time.perf_counter()
            n = g.fastq_scan(0, buf, True, **kw)
            t1 = time.perf_counter()
            assert n == a.reads, (name, n)
            g.fastq_take(0, n)
            g.fastq_commit(n, paired=False)
            ts.append(t1 - t0)
        b1, o1, _, _ = g.download_batch(a.reads)
        res[name] = {"scan_s": [round(t, 4) for t in ts], "text_GB_per_s": round(len(text) / min(ts) / 1e9, 2), "crc": zlib.crc32(b1.tobytes()) ^ zlib.crc32(o1.tobytes())}
    res["same_batch"] = res["plain"]["crc"] == res["bgzf"]["crc"]
    print(json.dumps(res))


if __name__ == "__main__":
    main()
