"""Small adversarial datasets for oracle-vs-device fuzzing (no reference needed: the oracle is
pinned by tests/test_oracle_golden.py).  Heavy repeat content pushes the code through the
paths typical data rarely reaches: seed-frequency caps (F0/F1, second round), long hit lists
(heap sort), many candidates per strand (lane-grouped verification with threshold break),
mate rescue with many windows, multi-mappers (reservoir sampling), chromosome edges."""
import os

import numpy as np

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
COMP = np.zeros(256, dtype=np.uint8)
for a, b in zip(b"ACGTN", b"TGCAN"):
    COMP[a] = b


def make_case(seed, genome=120_000, n_chr=3, elem_len=400, copies=700, div=0.01, pairs=3000, L=50, tandem=True):
    rng = np.random.default_rng(seed)
    lens = [genome // n_chr] * n_chr
    chroms = [ACGT[rng.integers(0, 4, size=l)].copy() for l in lens]
    elem = ACGT[rng.integers(0, 4, size=elem_len)]
    for _ in range(copies):
        c = chroms[rng.integers(0, n_chr)]
        p = rng.integers(0, len(c) - elem_len)
        cp = elem.copy()
        m = rng.random(elem_len) < div
        cp[m] = ACGT[rng.integers(0, 4, size=int(m.sum()))]
        if rng.random() < 0.5:
            cp = COMP[cp[::-1]]
        c[p:p + elem_len] = cp
    if tandem:
        c = chroms[0]
        unit = ACGT[rng.integers(0, 4, size=23)]
        c[1000:1000 + 23 * 120] = np.tile(unit, 120)
        c[5000:5300] = ord("A")
        c[7000:7040] = ord("N")
    r1, r2 = [], []
    for i in range(pairs):
        ci = int(rng.integers(0, n_chr))
        c = chroms[ci]
        fl = int(rng.integers(35, 500))
        u = rng.random()
        if u < 0.05:
            st = 0 if rng.random() < 0.5 else len(c) - fl
            st = int(min(max(0, st + int(rng.integers(-10, 10))), len(c) - fl))
        else:
            st = int(rng.integers(0, len(c) - fl))
        frag = c[st:st + fl]
        a = frag[:L].copy()
        b = COMP[frag[::-1]][:L].copy()
        for x in (a, b):
            m = rng.random(len(x)) < 0.015
            x[m] = ACGT[rng.integers(0, 4, size=int(m.sum()))]
        if rng.random() < 0.03:
            a[rng.integers(0, len(a))] = ord("N")
        if rng.random() < 0.5:
            a, b = b, a
        l1 = int(rng.integers(28, L + 1)) if rng.random() < 0.1 else len(a)
        r1.append(a[:l1].tobytes())
        r2.append(b.tobytes())
    return chroms, r1, r2


def write_case(d, seed, **kw):
    os.makedirs(d, exist_ok=True)
    chroms, r1, r2 = make_case(seed, **kw)
    fa = os.path.join(d, "f.fa")
    with open(fa, "wb") as f:
        for i, c in enumerate(chroms):
            f.write(b">c%d\n" % i + c.tobytes() + b"\n")

    def pack(rs):
        off = np.zeros(len(rs) + 1, np.uint32)
        off[1:] = np.cumsum([len(r) for r in rs])
        return np.frombuffer(b"".join(rs), np.uint8).copy(), off
    b1, o1 = pack(r1)
    b2, o2 = pack(r2)
    return fa, b1, o1, b2, o2


CONFIGS = [
    # (seed, preset, param overrides, generator overrides)
    (1, "atac", {"mapq_threshold": 0}, {}),
    (2, "chip", {"mapq_threshold": 0}, {"copies": 1200, "elem_len": 200}),
    (3, "atac", {"mapq_threshold": 0, "max_seed_freq0": 40, "max_seed_freq1": 90}, {"copies": 300}),
    (4, "hic", {"mapq_threshold": 0}, {"L": 100, "copies": 200}),
    (5, None, {"mapq_threshold": 0, "error_threshold": 5}, {"L": 70}),
    (6, "chip", {"mapq_threshold": 0, "min_num_seeds": 3, "max_insert_size": 300}, {"div": 0.0, "copies": 150}),
]


def write_wrap_case(d):
    """a repeat element that also stands at the very start of every chromosome, and reads that carry 15 foreign bases in front of the
    element's first 35: the + hit at position 0 has its diagonal before the sequence start (candidate position wraps below zero)"""
    os.makedirs(d, exist_ok=True)
    rng = np.random.default_rng(77)
    elem = ACGT[rng.integers(0, 4, 300)]
    chroms = [ACGT[rng.integers(0, 4, 40000)].copy() for _ in range(3)]
    for c in chroms:
        c[:300] = elem
        for _ in range(40):
            p = int(rng.integers(400, len(c) - 400))
            c[p:p + 300] = elem
    fa = os.path.join(d, "w.fa")
    with open(fa, "wb") as f:
        for i, c in enumerate(chroms):
            f.write(b">c%d\n" % i + c.tobytes() + b"\n")
    r1, r2 = [], []
    for i in range(300):
        a = np.concatenate([ACGT[rng.integers(0, 4, 15)], elem[:35]])
        b = COMP[chroms[i % 3][0:260][::-1]][:50]
        r1.append(a.tobytes())
        r2.append(b.tobytes())

    def pack(rs):
        off = np.zeros(len(rs) + 1, np.uint32)
        off[1:] = np.cumsum([len(r) for r in rs])
        return np.frombuffer(b"".join(rs), np.uint8).copy(), off
    b1, o1 = pack(r1)
    b2, o2 = pack(r2)
    return fa, b1, o1, b2, o2
