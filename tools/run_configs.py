#!/usr/bin/env python3
"""BASELINE.json's configurations at their FULL sizes on one MI355X (configs[1..4]; configs[0] is the CPU plumbing case):

  1  --preset chip,  10 M pairs 2x50            resident batches -> device store -> sort / dedup / BED text
  2  --preset atac, 100 M pairs 2x50            the same (the 1 -> 8 GPU scaling of this config is bench.py --gpus N)
  3  --preset atac, 200 M pairs 2x50 + 16-base cell barcodes, 737 280-entry whitelist (host buffers: cmgpu_map_pairs_barcoded)
  4  --preset hic,   50 M pairs 2x150, 0.1 % indels, split alignment (pairs records downloaded per batch)

Reads come from the device generator (seeded per batch), the index is the GRCh38-sized synthetic one of bench.py.  What is
checked here is what does not depend on the size: every batch maps, the counters add up (records in the store = records the
batches reported), the mapped fraction of the full run equals that of its first 16 M pairs, and the BED text of those first
16 M pairs has the same md5 on a second run (the pipeline places minimizers and list entries with atomics and maps three
ranges of a batch side by side: the text must not depend on that order).  The bit-for-bit comparison
with the reference binary on these presets is tools/ref_baseline.py (--preset chip at the full 10 M pairs; atac and hic on
slices) and the golden cases under tests/.  Prints one JSON object per configuration.

    python tools/run_configs.py [--only 1 2 3 4] [--scale 1.0]
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def stats_dict(st):
    return st.as_dict()


def run_bulk(cfg, args):
    import numpy as np  # noqa: F401
    from chromap_amd import ChromapGPU, Stats, _capi
    total = int(cfg["pairs"] * args.scale)
    bs = cfg["batch"]
    g = ChromapGPU(synthetic=(args.genome, args.nseq, 12345, None), preset=cfg["preset"])
    g.set_option("lanes", 3)
    out = {"config": cfg["name"], "pairs": total, "batch_pairs": bs}

    def run(n_total, batch, seed0, keep_text):
        g.store_clear()
        st = Stats()
        reported = 0
        t0 = time.perf_counter()
        done = 0
        b = 0
        while done < n_total:
            n = min(batch, n_total - done)
            g.generate_resident(n, cfg["readlen"], cfg["frag_min"], cfg["frag_max"], 0.01, seed0 + b, cfg.get("indel", 0.0))
            # read ids continue across batches, and every batch starts on a read_batch_size boundary of the "file"
            assert g.L.cmgpu_set_option(g.ctx, b"first_read_id", done) == 0
            reported += g.map_resident(st)
            if not cfg.get("split"):
                g.store_append_resident()
            done += n
            b += 1
        t_map = time.perf_counter() - t0
        res = {"batches": b, "map_s": round(t_map, 3), "M_pairs_per_s_incl_generation": round(n_total / t_map / 1e6, 1),
               "records_reported": int(reported), "stats": stats_dict(st)}
        if not cfg.get("split"):
            t1 = time.perf_counter()
            lines, nbytes = g.store_format(_capi.TEXT_BED_PE)
            res["post_s"] = round(time.perf_counter() - t1, 3)
            info = (C.c_uint64(0), C.c_uint64(0), C.c_uint64(0))
            g.L.cmgpu_store_info(g.ctx, C.byref(info[0]), C.byref(info[1]), C.byref(info[2]))
            res.update({"store_records": int(info[0].value), "bed_lines": int(lines), "bed_bytes": int(nbytes)})
            res["records_add_up"] = int(info[0].value) == int(reported)
            if keep_text:
                res["bed_md5"] = hashlib.md5(g.store_text()).hexdigest()
        return res

    prefix = min(total, 16_000_000)
    full = run(total, bs, 1000, False)
    out["full"] = full
    mapped_full = full["stats"]["num_mapped_reads"] / (2.0 * total)
    pa = run(prefix, bs, 1000, not cfg.get("split"))
    out["prefix"] = {"pairs": prefix, **pa}
    if not cfg.get("split"):
        out["prefix_md5_second_run"] = run(prefix, bs, 1000, True)["bed_md5"]
        out["deterministic"] = out["prefix_md5_second_run"] == pa["bed_md5"]
    out["mapped_fraction"] = {"full": round(mapped_full, 5), "prefix": round(pa["stats"]["num_mapped_reads"] / (2.0 * prefix), 5)}
    out["mapped_fraction_agrees"] = abs(out["mapped_fraction"]["full"] - out["mapped_fraction"]["prefix"]) < 2e-3
    g.close()
    return out


def run_barcoded(cfg, args):
    import numpy as np
    from chromap_amd import ChromapGPU, Stats, _capi
    from chromap_amd._capi import Batch, BarcodeBatch
    total = int(cfg["pairs"] * args.scale)
    bs = cfg["batch"]
    g = ChromapGPU(synthetic=(args.genome, args.nseq, 12345, None), preset="atac")
    g.set_option("lanes", 3)
    rng = np.random.default_rng(737)
    n_wl = 737_280
    wl = np.unique(rng.integers(0, 1 << 32, n_wl, dtype=np.uint64))  # 16-base barcodes as 2-bit keys
    keys = np.ascontiguousarray(wl)
    assert g.L.cmgpu_set_whitelist(g.ctx, keys.ctypes.data, len(keys), 16) == 0
    g.barcode_length = 16
    letters = np.frombuffer(b"ACGT", np.uint8)
    # four host batches (reads from the device generator, downloaded once) and four barcode batches take turns
    host = []
    for b in range(4):
        g.generate_resident(bs, 50, 30, 600, 0.01, 3000 + b)
        b1, o1, b2, o2 = g.download_batch(bs)
        pick = wl[rng.integers(0, len(wl), bs)]
        codes = ((pick[:, None] >> (2 * (15 - np.arange(16, dtype=np.uint64)))) & np.uint64(3)).astype(np.uint8)
        err = rng.random(bs) < 0.05  # one substitution in 5 % of the barcodes: the correction kernel has work
        pos = rng.integers(0, 16, bs)
        codes[err, pos[err]] = (codes[err, pos[err]] + 1 + rng.integers(0, 3, int(err.sum()))) % 4
        bc = letters[codes].reshape(-1).copy()
        bq = np.full(bs * 16, ord("I"), np.uint8)
        bo = (np.arange(bs + 1, dtype=np.uint32) * 16).astype(np.uint32)
        host.append((b1, o1, b2, o2, bc, bq, bo))
    g.compute_barcode_abundance(host[0][4], host[0][6])
    st = Stats()
    g.store_clear()
    reported, done, b = 0, 0, 0
    t0 = time.perf_counter()
    while done < total:
        n = min(bs, total - done)
        b1, o1, b2, o2, bc, bq, bo = host[b & 3]
        bt = Batch(n, done, b1.ctypes.data, o1.ctypes.data, b2.ctypes.data, o2.ctypes.data)
        bb = BarcodeBatch(bc.ctypes.data, bq.ctypes.data, bo.ctypes.data)
        k = C.c_uint64(0)
        rc = g.L.cmgpu_map_pairs_barcoded(g.ctx, C.byref(bt), C.byref(bb), None, 0, C.byref(k), C.byref(st))
        assert rc == 0, g.L.cmgpu_last_error(g.ctx)
        reported += int(k.value)
        g.store_append_resident()
        done += n
        b += 1
    t_map = time.perf_counter() - t0
    t1 = time.perf_counter()
    lines, nbytes = g.store_format(_capi.TEXT_BED_PE_BC, barcode_length=16)
    post = time.perf_counter() - t1
    info = (C.c_uint64(0), C.c_uint64(0), C.c_uint64(0))
    g.L.cmgpu_store_info(g.ctx, C.byref(info[0]), C.byref(info[1]), C.byref(info[2]))
    out = {"config": cfg["name"], "pairs": total, "batch_pairs": bs, "whitelist": int(len(wl)), "batches": b,
           "map_s_incl_upload_from_pageable_host_buffers": round(t_map, 3), "M_pairs_per_s": round(total / t_map / 1e6, 1),
           "post_s": round(post, 3), "records_reported": reported, "store_records": int(info[0].value),
           "records_add_up": int(info[0].value) == reported, "bed_lines": int(lines), "bed_bytes": int(nbytes), "stats": stats_dict(st),
           "note": "four distinct read batches x four distinct barcode batches take turns (host memory); duplicates across turns are "
                   "removed by the cell-level dedup, hence bed_lines << records"}
    g.close()
    return out


CONFIGS = {
    1: {"name": "1: --preset chip, 10 M pairs 2x50", "preset": "chip", "pairs": 10_000_000, "batch": 2_000_000, "readlen": 50,
        "frag_min": 30, "frag_max": 600},
    2: {"name": "2: --preset atac, 100 M pairs 2x50", "preset": "atac", "pairs": 100_000_000, "batch": 4_000_000, "readlen": 50,
        "frag_min": 30, "frag_max": 600},
    3: {"name": "3: --preset atac + 16-base barcodes, 737K whitelist, 200 M pairs 2x50", "pairs": 200_000_000, "batch": 4_000_000, "barcoded": True},
    4: {"name": "4: --preset hic, 50 M pairs 2x150, 0.1 % indels", "preset": "hic", "pairs": 50_000_000, "batch": 2_000_000, "readlen": 150,
        "frag_min": 300, "frag_max": 800, "indel": 0.001, "split": True},
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", type=int, nargs="*", default=[1, 2, 3, 4])
    ap.add_argument("--scale", type=float, default=1.0, help="fraction of each configuration's pairs (smoke runs)")
    ap.add_argument("--genome", type=int, default=3_100_000_000)
    ap.add_argument("--nseq", type=int, default=24)
    args = ap.parse_args()
    import torch  # noqa: F401
    for k in args.only:
        cfg = CONFIGS[k]
        t0 = time.time()
        try:
            res = run_barcoded(cfg, args) if cfg.get("barcoded") else run_bulk(cfg, args)
        except Exception as e:  # keep going: one JSON object per configuration
            res = {"config": cfg["name"], "error": repr(e)}
        res["wall_s"] = round(time.time() - t0, 1)
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
