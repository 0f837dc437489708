#!/usr/bin/env python3
"""Host-buffer boundary of the mapping path (cmgpu_submit_pairs / cmgpu_map_submitted) under bench.py's conditions:
GRCh38-sized synthetic index, 4 M pairs of 2 x 50 per batch, page-locked host buffers.  Several upload / download
settings are timed in ONE process (the index is built once).  One JSON line per setting.

    python tools/pcie_bench.py [--pairs N] [--settings "h2d_copy_blocks=0" "h2d_copy_blocks=64,d2h_copy_blocks=32" ...]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=4_000_000)
    ap.add_argument("--genome", type=int, default=3_100_000_000)
    ap.add_argument("--nseq", type=int, default=24)
    ap.add_argument("--readlen", type=int, default=50)
    ap.add_argument("--repeats", type=int, default=6)
    ap.add_argument("--settings", nargs="*", default=["h2d_copy_blocks=0", "h2d_copy_blocks=64"])
    ap.add_argument("--pre", nargs="*", default=[], help="things bench.py does before its boundary measurement: lanes3, pageable, parked")
    args = ap.parse_args()
    import numpy as np
    import torch  # noqa: F401  (HIP runtime of the process)
    from chromap_amd import ChromapGPU, Stats
    g = ChromapGPU(synthetic=(args.genome, args.nseq, 12345, None), preset="atac", device=0)
    n = args.pairs
    g.generate_resident(n, args.readlen, 30, 600, 0.01, 1000)
    g.map_resident(Stats())
    o1 = np.zeros(n + 1, np.uint32)
    o2 = np.zeros(n + 1, np.uint32)
    b1 = np.zeros(n * args.readlen, np.uint8)
    b2 = np.zeros(n * args.readlen, np.uint8)
    assert g.L.cmgpu_download_batch(g.ctx, b1.ctypes.data, o1.ctypes.data, b2.ctypes.data, o2.ctypes.data) == 0
    if "parked" in args.pre:
        for b in range(1, 4):
            g.swap_resident(b)
            g.generate_resident(n, args.readlen, 30, 600, 0.01, 1000 + b)
            g.swap_resident(b)
    if "lanes3" in args.pre:
        g.set_option("lanes", 3)
        for _ in range(5):
            g.map_resident(Stats())
    if "pageable" in args.pre:
        g.set_option("lanes", 1)
        g.map_pairs(b1, o1, b2, o2)
        g.map_pairs(b1, o1, b2, o2)
    for setting in args.settings:
        opts = dict(kv.split("=") for kv in setting.split(",") if kv)
        g.set_option("lanes", 1)
        g.set_option("h2d_copy_blocks", 0)
        g.set_option("d2h_copy_blocks", 0)
        for k, v in opts.items():
            g.set_option(k, int(v))
        t0 = time.perf_counter()
        r = g.map_pairs_pipelined(b1, o1, b2, o2, repeats=args.repeats)
        r["setting"] = setting
        r["wall_s"] = round(time.perf_counter() - t0, 2)
        r.pop("note", None)
        print(json.dumps(r), flush=True)


if __name__ == "__main__":
    main()
