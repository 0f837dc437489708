#!/usr/bin/env python3
"""Distribution of the list lengths the stages of one mapped batch saw (hit lists, rescue hits, merged / filtered candidates,
draft mappings, best pairings): which size classes hold the work on a given genome.  GPU tool (cmgpu_debug_array).
usage: python tools/list_hist.py --repeats profile:1 [--pairs 4000000]"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chromap_amd import ChromapGPU, Stats  # noqa: E402


def arr(g, name):
    n = C.c_uint64(0)
    cap = 2 * g.n_pairs_resident
    out = np.zeros(cap, np.uint32)
    rc = g.L.cmgpu_debug_array(g.ctx, name.encode(), out.ctypes.data, cap, C.byref(n))
    if rc != 0:
        raise RuntimeError(g.L.cmgpu_last_error(g.ctx).decode())
    return out[:n.value].astype(np.int64)


def describe(name, v, ths=(16, 48, 64, 256, 1024, 2048, 4096, 8192, 16384, 65535)):
    nz = v[v > 0]
    o = {"name": name, "n": int(v.size), "nonzero": int(nz.size), "sum": int(v.sum()), "max": int(v.max()) if v.size else 0}
    if nz.size:
        for q in (50, 90, 99, 99.9, 99.99):
            o["p%s" % q] = int(np.percentile(nz, q))
    o["above"] = {str(t): [int((v > t).sum()), int(v[v > t].sum())] for t in ths if (v > t).any()}
    return o


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--repeats", default="profile:1")
    ap.add_argument("--pairs", type=int, default=4_000_000)
    ap.add_argument("--genome", type=int, default=3_100_000_000)
    ap.add_argument("--nseq", type=int, default=24)
    ap.add_argument("--seed", type=int, default=5000)
    a = ap.parse_args()
    rep = a.repeats if a.repeats.startswith("profile:") else (tuple(float(x) if "." in x else int(x) for x in a.repeats.split(",")) if a.repeats else None)
    g = ChromapGPU(synthetic=(a.genome, a.nseq, 12345, rep), preset="atac", device=0)
    g.set_option("lanes", 1)
    g.set_option("probe_table_shift", 1)
    g.generate_resident(a.pairs, read_length=50, frag_min=30, frag_max=600, sub_rate=0.01, seed=a.seed)
    g.n_pairs_resident = a.pairs
    st = Stats()
    g.map_resident(st)
    res = {"workload": a.repeats, "pairs": a.pairs, "counters": st.as_dict(), "timings": g.timings()}
    A = {k: arr(g, k) for k in ("hit_tot", "ncp", "ncn", "resc_p", "resc_n", "mcp", "mcn", "fcp", "fcn", "nv", "ndp", "ndn", "pe_nbest")}
    lists = []
    lists.append(describe("hit_tot (S3b hit list per read)", A["hit_tot"]))
    lists.append(describe("cand per read after S3b (ncp+ncn)", A["ncp"] + A["ncn"]))
    lists.append(describe("rescue hits per read, larger strand (S4b)", np.maximum(A["resc_p"], A["resc_n"])))
    mc = np.maximum(A["mcp"], A["mcn"])
    lists.append(describe("merged candidates per pair, largest list (S4c)", np.maximum(mc[0::2], mc[1::2])))
    lists.append(describe("filtered candidates per read (S5: fcp+fcn)", A["fcp"] + A["fcn"]))
    nd = np.maximum(A["ndp"], A["ndn"])
    lists.append(describe("draft mappings per pair, largest list (S6a)", np.maximum(nd[0::2], nd[1::2])))
    # S6a's work: first-list entries x partner ranges is data dependent; the products bound it
    r1p, r1n, r2p, r2n = A["ndp"][0::2], A["ndn"][0::2], A["ndp"][1::2], A["ndn"][1::2]
    lists.append(describe("S6a first-list entries per pair (ndp1 + ndn1)", r1p + r1n))
    lists.append(describe("S6a product bound per pair (ndp1*ndn2 + ndn1*ndp2)", r1p * r2n + r1n * r2p,
                          ths=(10**3, 10**4, 10**5, 10**6, 10**7, 10**8)))
    lists.append(describe("pe_nbest per pair", A["pe_nbest"]))
    res["lists"] = lists
    print(json.dumps(res))
    for o in lists:
        print("%-60s nonzero %9d sum %12d max %9d  p50 %s p99 %s p99.9 %s" % (o["name"], o["nonzero"], o["sum"], o["max"], o.get("p50"), o.get("p99"), o.get("p99.9")), file=sys.stderr)
        print("      above (count, sum): %s" % o["above"], file=sys.stderr)
    g.close()


if __name__ == "__main__":
    main()
