"""Shared builder for the dense-whitelist barcode fuzz (short barcodes, nearly complete whitelist:
a barcode outside the whitelist has hundreds of whitelisted neighbours within two substitutions,
more than the device's candidate buffer -- the selection path of cm_s0b_barcode)."""
import itertools
import os

import numpy as np

import datasets
import oracle_lib as ol

BC_LEN = 8


def build(tmp, n_pairs=2500, seed=77):
    """returns (fa, idx, b1, o1, b2, o2, bc, bcq, bco, whitelist_path)"""
    case = "b2_atac_bc2_q0"
    fa, r1, r2 = datasets.case_inputs(case)
    idx = datasets.case_index(case)
    b1, o1 = ol.read_fastx(r1)
    b2, o2 = ol.read_fastx(r2)
    o1 = o1[:n_pairs + 1].copy()
    o2 = o2[:n_pairs + 1].copy()
    b1 = b1[:o1[-1]].copy()
    b2 = b2[:o2[-1]].copy()
    rng = np.random.default_rng(seed)
    allk = ["".join(p) for p in itertools.product("ACGT", repeat=BC_LEN)]
    missing = set(rng.choice(len(allk), 60, replace=False).tolist())
    wl = os.path.join(tmp, "wl.txt")
    with open(wl, "w") as f:
        for i, s in enumerate(allk):
            if i not in missing:
                f.write(s + "\n")
    miss = [allk[i] for i in sorted(missing)]
    bcs, quals = [], []
    for i in range(n_pairs):
        u = rng.random()
        if u < 0.45:
            s = miss[rng.integers(0, len(miss))]            # not whitelisted: 24 + 252 neighbours to rank
        elif u < 0.55:
            s = list(allk[rng.integers(0, len(allk))])       # one or two N
            for _ in range(1 + (rng.random() < 0.5)):
                s[rng.integers(0, BC_LEN)] = "N"
            s = "".join(s)
        else:
            s = allk[rng.integers(0, 400)]                    # skewed abundance
        bcs.append(s)
        quals.append("".join(chr(33 + int(q)) for q in rng.choice([2, 11, 25, 37, 41], BC_LEN)))
    bc = np.frombuffer("".join(bcs).encode(), np.uint8).copy()
    bcq = np.frombuffer("".join(quals).encode(), np.uint8).copy()
    bco = (np.arange(n_pairs + 1) * BC_LEN).astype(np.uint32)
    return fa, idx, b1, o1, b2, o2, bc, bcq, bco, wl


def oracle_result(fa, idx, b1, o1, b2, o2, bc, bcq, bco, wl_path, bc_err):
    o = ol.Oracle(idx, fa, ol.params("atac", mapq_threshold=0, bc_error_threshold=bc_err))
    wl = ol.Whitelist(wl_path, BC_LEN)
    wl.abundance(bc, bco)
    bcc = bc.copy()
    rec, k, st, n_in, n_corr = ol.map_pairs_bc(o, b1, o1, b2, o2, bcc, bcq, bco, wl)
    tup = sorted((rec[i].r.read_id, rec[i].r.rid, rec[i].r.fragment_start, rec[i].r.fragment_length, rec[i].r.mapq,
                  rec[i].r.direction, rec[i].barcode) for i in range(k))
    keys, _ = wl.export()
    o.close()
    return tup, n_in, n_corr, keys
