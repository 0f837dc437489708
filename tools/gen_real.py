#!/usr/bin/env python3
"""Instrument-shaped synthetic data at scale (inputs only -- no reference code involved), vectorised so that a million pairs take
seconds: what tools/gen_synth.py's per-read loop covers at 10^4-10^5 pairs, with the properties real FASTQ files have and the
constant-quality synthetic batches lack:

  reference   random bases with planted repeat families; 10 % of it SOFT-MASKED (lower case, as RepeatMasker leaves it), IUPAC
              ambiguity codes (R Y K M S W B D H V) singly and in short runs, N runs of 20-3000 bases, lines of 60 columns
  reads       lengths MIXED per read, 36-151 bases (what adapter / quality trimming upstream leaves), per-base qualities from a
              position-dependent model (high plateau, decaying tail, occasional low-quality windows), substitution errors drawn
              FROM the qualities (p = 10^(-Q/10)), 0.5 % of the bases N with quality '#', adapter read-through for short fragments,
              PCR duplicates, R1 / R2 swapped at random; FASTQ with '+' lines, optionally gzip / BGZF
"""
import argparse
import gzip
import os
import sys

import numpy as np

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
COMP = np.arange(256, dtype=np.uint8)
for a_, b_ in zip(b"ACGTNacgtnRYKMSWBDHVrykmswbdhv", b"TGCANtgcanYRMKSWVHDByrmkswvhdb"):
    COMP[a_] = b_
UPPER = np.arange(256, dtype=np.uint8)
for ch in range(ord("a"), ord("z") + 1):
    UPPER[ch] = ch - 32
ADAPTER1 = np.frombuffer(b"CTGTCTCTTATACACATCTCCGAGCCCACGAGACTAAGGCGAATCTCGTATGCCGTCTTCTGCTTG" * 4, dtype=np.uint8)
ADAPTER2 = np.frombuffer(b"CTGTCTCTTATACACATCTGACGCTGCCGACGAGTGTAGATCTCGGTGGTCGCCGTATCATTAAAA" * 4, dtype=np.uint8)
IUPAC = np.frombuffer(b"RYKMSWBDHV", dtype=np.uint8)


def make_genome(rng, total, n_chr):
    w = np.linspace(3.0, 1.0, n_chr)
    lens = np.maximum((w / w.sum() * total).astype(np.int64), 5000)
    chroms = []
    for ln in lens:
        c = ACGT[rng.integers(0, 4, size=int(ln))].copy()
        chroms.append(c)
    # repeat families: 3-kb elements at 2 % divergence, 300-base elements at 8 %
    for elen, div, copies in ((3000, 0.02, max(8, total // 40000)), (300, 0.08, max(20, total // 8000))):
        elem = ACGT[rng.integers(0, 4, size=elen)]
        for _ in range(int(copies)):
            c = chroms[int(rng.integers(0, n_chr))]
            if len(c) < 3 * elen:
                continue
            p = int(rng.integers(100, len(c) - elen - 100))
            cp = elem.copy()
            m = rng.random(elen) < div
            cp[m] = ACGT[rng.integers(0, 4, size=int(m.sum()))]
            c[p:p + elen] = COMP[cp[::-1]] if rng.random() < 0.5 else cp
    for c in chroms:
        n = len(c)
        # soft-masked segments: ~10 % of the bases, pieces of 200-5000
        masked = 0
        while masked < n // 10:
            ln = int(rng.integers(200, 5000))
            p = int(rng.integers(0, max(1, n - ln)))
            c[p:p + ln] |= 0x20  # ASCII lower case
            masked += ln
        # IUPAC codes: single ones (1 in 20 000 bases) and a few short runs; either case
        k = max(1, n // 20000)
        pos = rng.integers(0, n, size=k)
        c[pos] = IUPAC[rng.integers(0, len(IUPAC), size=k)] | np.where(rng.random(k) < 0.3, 0x20, 0).astype(np.uint8)
        for _ in range(max(1, n // 2_000_000)):
            ln = int(rng.integers(2, 12))
            p = int(rng.integers(0, n - ln))
            c[p:p + ln] = IUPAC[rng.integers(0, len(IUPAC), size=ln)]
        # N runs
        for _ in range(max(1, n // 3_000_000)):
            ln = int(rng.integers(20, 3000))
            p = int(rng.integers(0, max(1, n - ln)))
            c[p:p + ln] = ord("N") if rng.random() < 0.8 else ord("n")
    return chroms


def qualities(rng, n, lmax):
    """Phred scores, n x lmax: a plateau near 37, a tail that decays from a random onset, low-quality windows in a tenth of the reads"""
    pos = np.arange(lmax)[None, :]
    onset = rng.integers(lmax // 3, lmax + 40, size=(n, 1))
    slope = rng.uniform(0.05, 0.45, size=(n, 1))
    q = 37.0 - np.maximum(0, pos - onset) * slope + rng.normal(0, 1.5, size=(n, lmax))
    q[:, :4] -= rng.uniform(0, 6, size=(n, 4))  # the first cycles are a little worse
    bad = rng.random(n) < 0.1
    wst = rng.integers(0, lmax, size=n)
    wln = rng.integers(3, 25, size=n)
    win = bad[:, None] & (pos >= wst[:, None]) & (pos < (wst + wln)[:, None])
    q[win] = rng.uniform(2, 15, size=int(win.sum()))
    return np.clip(np.rint(q), 2, 41).astype(np.uint8)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--genome", type=int, default=50_000_000)
    ap.add_argument("--chroms", type=int, default=8)
    ap.add_argument("--pairs", type=int, default=1_000_000)
    ap.add_argument("--len-min", type=int, default=36)
    ap.add_argument("--len-max", type=int, default=151)
    ap.add_argument("--frag-min", type=int, default=40)
    ap.add_argument("--frag-max", type=int, default=700)
    ap.add_argument("--n-rate", type=float, default=0.005)
    ap.add_argument("--dup-frac", type=float, default=0.03)
    ap.add_argument("--seed", type=int, default=2026)
    ap.add_argument("--gz", action="store_true")
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    chroms = make_genome(rng, a.genome, a.chroms)
    with open(a.out + ".fa", "wb") as f:
        for i, c in enumerate(chroms):
            f.write(b">chr%d instrument-shaped synthetic len=%d\n" % (i + 1, len(c)))
            b = c.tobytes()
            f.write(b"\n".join(b[p:p + 60] for p in range(0, len(b), 60)) + b"\n")
    offs = np.concatenate([[0], np.cumsum([len(c) for c in chroms])]).astype(np.int64)
    G = np.concatenate(chroms + [np.full(4096, ord("N"), np.uint8)])  # (reads of the last fragment may look beyond it; masked below)
    lmax = a.len_max
    pos = np.arange(lmax, dtype=np.int64)[None, :]
    op = gzip.open if a.gz else open
    sfx = ".gz" if a.gz else ""
    prev = None  # the last fragment of the chunk before (a PCR duplicate of it may open the next chunk)
    with op(a.out + "_1.fq" + sfx, "wb") as f1, op(a.out + "_2.fq" + sfx, "wb") as f2:
        for base in range(0, a.pairs, 100_000):  # (chunks: the n x lmax arrays of a million pairs are gigabytes)
            n = min(100_000, a.pairs - base)
            ci = rng.choice(len(chroms), size=n, p=np.diff(offs) / offs[-1])
            clen = np.diff(offs)[ci]
            fl = rng.integers(a.frag_min, a.frag_max, size=n)
            st = (rng.random(n) * np.maximum(1, clen - fl)).astype(np.int64)
            isdup = rng.random(n) < a.dup_frac
            if prev is None:
                isdup[0] = False
            for i in np.nonzero(isdup)[0]:  # PCR duplicates: the fragment before, fresh errors
                ci[i], fl[i], st[i] = (ci[i - 1], fl[i - 1], st[i - 1]) if i else prev
            prev = (ci[-1], fl[-1], st[-1])
            clen = np.diff(offs)[ci]
            fl = np.minimum(fl, clen)
            g0 = offs[ci] + st
            out = []
            for mate, adapter in ((0, ADAPTER1), (1, ADAPTER2)):
                if mate == 0:
                    seq = G[np.minimum(g0[:, None] + pos, len(G) - 1)]
                else:
                    seq = COMP[G[np.maximum((g0 + fl - 1)[:, None] - pos, 0)]]
                seq = UPPER[seq]  # a sequencer emits upper case; ambiguity codes of the reference come through as they are
                past = pos >= fl[:, None]  # adapter read-through
                seq = np.where(past, adapter[np.minimum(np.maximum(pos - fl[:, None], 0), len(adapter) - 1)], seq)
                q = qualities(rng, n, lmax)
                err = rng.random((n, lmax)) < np.power(10.0, -q.astype(np.float64) / 10.0)
                seq = np.where(err, ACGT[rng.integers(0, 4, size=(n, lmax))], seq)
                isn = rng.random((n, lmax)) < a.n_rate
                seq = np.where(isn, np.uint8(ord("N")), seq).astype(np.uint8)
                q = np.where(isn, np.uint8(2), q)
                out.append((seq, (q + 33).astype(np.uint8)))
            len1 = rng.integers(a.len_min, a.len_max + 1, size=n)
            len2 = rng.integers(a.len_min, a.len_max + 1, size=n)
            same = rng.random(n) < 0.6  # most pairs were trimmed little: full length on both mates
            len1[same] = a.len_max
            len2[same & (rng.random(n) < 0.8)] = a.len_max
            swap = rng.random(n) < 0.5
            (s1, q1), (s2, q2) = out
            for i in range(n):
                a1 = (s1[i, :len1[i]].tobytes(), q1[i, :len1[i]].tobytes())
                a2 = (s2[i, :len2[i]].tobytes(), q2[i, :len2[i]].tobytes())
                if swap[i]:
                    a1, a2 = a2, a1
                k = base + i
                f1.write(b"@M0:%d:FC:1:%d:%d:%d 1:N:0:ACGT\n%s\n+\n%s\n" % (a.seed, k // 1000, k % 1000, k, a1[0], a1[1]))
                f2.write(b"@M0:%d:FC:1:%d:%d:%d 2:N:0:ACGT\n%s\n+\n%s\n" % (a.seed, k // 1000, k % 1000, k, a2[0], a2[1]))
    n = a.pairs
    print("wrote %s.fa (%d bases, %d sequences) and %d pairs" % (a.out, int(offs[-1]), len(chroms), n), file=sys.stderr)


if __name__ == "__main__":
    main()
