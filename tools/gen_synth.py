#!/usr/bin/env python3
"""Synthetic genome + paired-end read generator (inputs only -- no reference code involved).

Produces a FASTA reference and FASTQ read pairs whose shape follows SURVEY.md 8(d):
uniform-random genome with planted repeat families, low-complexity stretches, N runs
and a soft-masked (lower-case) region; read pairs drawn from fragments with
substitutions / indels / Ns, adapter read-through for short fragments, PCR duplicates,
chromosome-edge fragments and unmappable junk pairs.  Everything is seeded.
"""
import argparse
import gzip
import numpy as np

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
COMP = np.zeros(256, dtype=np.uint8)
for a, b in zip(b"ACGTNacgtn", b"TGCANtgcan"):
    COMP[a] = b
ADAPTER1 = np.frombuffer(b"CTGTCTCTTATACACATCTCCGAGCCCACGAGACTAAGGCGAATCTCGTATGCCGTCTTCTGCTTG" * 4, dtype=np.uint8)
ADAPTER2 = np.frombuffer(b"CTGTCTCTTATACACATCTGACGCTGCCGACGAGTGTAGATCTCGGTGGTCGCCGTATCATTAAAA" * 4, dtype=np.uint8)


def revcomp(a):
    return COMP[a[::-1]]


def make_genome(rng, total, n_chr, repeats=True):
    # chromosome lengths with a spread (largest ~3x smallest)
    w = np.linspace(3.0, 1.0, n_chr)
    lens = np.maximum((w / w.sum() * total).astype(np.int64), 2000)
    chroms = [ACGT[rng.integers(0, 4, size=int(l))].copy() for l in lens]
    if repeats:
        # repeat family: element of 3 kb, copies with 2% divergence
        n_copies = max(4, int(total // 30000))
        elem = ACGT[rng.integers(0, 4, size=3000)]
        for _ in range(n_copies):
            c = chroms[rng.integers(0, n_chr)]
            if len(c) < 8000:
                continue
            p = rng.integers(100, len(c) - 3100)
            cp = elem.copy()
            m = rng.random(3000) < 0.02
            cp[m] = ACGT[rng.integers(0, 4, size=int(m.sum()))]
            if rng.random() < 0.5:
                cp = revcomp(cp)
            c[p:p + 3000] = cp
        # an exact-copy short family (300 bp x many) -> multi-mappers with equal scores
        elem2 = ACGT[rng.integers(0, 4, size=300)]
        for _ in range(max(3, n_copies // 2)):
            c = chroms[rng.integers(0, n_chr)]
            p = rng.integers(100, len(c) - 400)
            c[p:p + 300] = elem2
        # low complexity: poly-A, (AC)n, (AAG)n stretches
        for _ in range(max(2, int(total // 200000))):
            c = chroms[rng.integers(0, n_chr)]
            p = rng.integers(100, len(c) - 700)
            kind = rng.integers(0, 3)
            unit = [b"A", b"AC", b"AAG"][kind]
            n = int(rng.integers(100, 600))
            s = np.frombuffer((unit * (n // len(unit) + 1))[:n], dtype=np.uint8)
            c[p:p + n] = s
        # N runs and a soft-masked region
        for ci in range(n_chr):
            c = chroms[ci]
            if len(c) > 20000:
                p = rng.integers(1000, len(c) - 3000)
                c[p:p + int(rng.integers(20, 1500))] = ord("N")
                p = rng.integers(1000, len(c) - 6000)
                seg = c[p:p + 5000]
                low = seg.copy()
                low[seg == ord("A")] = ord("a")
                low[seg == ord("C")] = ord("c")
                low[seg == ord("G")] = ord("g")
                low[seg == ord("T")] = ord("t")
                c[p:p + 5000] = low
    return chroms


def mutate(rng, seq, sub, indel):
    out = seq.copy()
    m = rng.random(len(out)) < sub
    if m.any():
        out[m] = ACGT[rng.integers(0, 4, size=int(m.sum()))]
    if indel > 0 and rng.random() < indel * len(out):
        p = int(rng.integers(1, max(2, len(out) - 1)))
        if rng.random() < 0.5:
            out = np.concatenate([out[:p], ACGT[rng.integers(0, 4, size=1)], out[p:]])
        else:
            out = np.concatenate([out[:p], out[p + 1:]])
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True, help="output prefix")
    ap.add_argument("--genome", type=int, default=2_000_000)
    ap.add_argument("--chroms", type=int, default=4)
    ap.add_argument("--pairs", type=int, default=10000)
    ap.add_argument("--readlen", type=int, default=50)
    ap.add_argument("--frag-min", type=int, default=0, help="default 2L; <L gives adapter read-through")
    ap.add_argument("--frag-max", type=int, default=600)
    ap.add_argument("--sub", type=float, default=0.01)
    ap.add_argument("--indel", type=float, default=0.001, help="per-base prob of one 1-bp indel per read")
    ap.add_argument("--seed", type=int, default=12345)
    ap.add_argument("--no-repeats", action="store_true")
    ap.add_argument("--dup-frac", type=float, default=0.05)
    ap.add_argument("--junk-frac", type=float, default=0.02)
    ap.add_argument("--n-frac", type=float, default=0.01, help="fraction of reads that get an N")
    ap.add_argument("--varlen", action="store_true", help="vary read lengths (some < 30, some longer)")
    ap.add_argument("--barcodes", type=int, default=0, help="if >0: whitelist size; writes .bc.fq and .whitelist.txt")
    ap.add_argument("--reads-only", action="store_true", help="reuse existing <out>.fa")
    ap.add_argument("--gz", action="store_true")
    ap.add_argument("--hic", action="store_true",
                    help="Hi-C like pairs: mates from two independent loci, some reads chimeric across the ligation junction")
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    L = a.readlen
    if a.reads_only:
        chroms = []
        cur = []
        with open(a.out + ".fa", "rb") as f:
            for line in f:
                if line.startswith(b">"):
                    if cur:
                        chroms.append(np.frombuffer(b"".join(cur), dtype=np.uint8).copy())
                    cur = []
                else:
                    cur.append(line.strip())
        if cur:
            chroms.append(np.frombuffer(b"".join(cur), dtype=np.uint8).copy())
        rng = np.random.default_rng(a.seed + 1)
    else:
        chroms = make_genome(rng, a.genome, a.chroms, repeats=not a.no_repeats)
        with open(a.out + ".fa", "wb") as f:
            for i, c in enumerate(chroms):
                f.write(b">chr%d synthetic len=%d\n" % (i + 1, len(c)))
                b = c.tobytes()
                for p in range(0, len(b), 70):
                    f.write(b[p:p + 70] + b"\n")
    clens = np.array([len(c) for c in chroms], dtype=np.float64)
    cprob = clens / clens.sum()
    fmin = a.frag_min if a.frag_min > 0 else 2 * L
    op = gzip.open if a.gz else open
    sfx = ".gz" if a.gz else ""
    f1 = op(a.out + "_1.fq" + sfx, "wb")
    f2 = op(a.out + "_2.fq" + sfx, "wb")
    fb = None
    wl = None
    if a.barcodes > 0:
        wl = ACGT[rng.integers(0, 4, size=(a.barcodes, 16))]
        with open(a.out + ".whitelist.txt", "wb") as f:
            for b in wl:
                f.write(b.tobytes() + b"\n")
        fb = op(a.out + "_bc.fq" + sfx, "wb")
        # skewed cell abundance
        wl_p = rng.random(a.barcodes) ** 3
        wl_p /= wl_p.sum()
    prev = None
    for i in range(a.pairs):
        u = rng.random()
        if prev is not None and u < a.dup_frac:
            ci, st, fl = prev  # PCR duplicate: same fragment again (fresh errors)
        else:
            ci = int(rng.choice(len(chroms), p=cprob))
            fl = int(rng.integers(fmin, a.frag_max))
            c = chroms[ci]
            if rng.random() < 0.01:  # chromosome-edge fragments
                st = 0 if rng.random() < 0.5 else max(0, len(c) - fl)
                if rng.random() < 0.5:
                    st = min(max(0, len(c) - fl), max(0, st + int(rng.integers(-12, 12))))
            else:
                st = int(rng.integers(0, max(1, len(c) - fl)))
            prev = (ci, st, fl)
        c = chroms[ci]
        frag = c[st:st + fl]
        l1 = l2 = L
        if a.varlen:
            r = rng.random()
            if r < 0.05:
                l1 = int(rng.integers(20, 30))
            elif r < 0.3:
                l1 = int(rng.integers(30, L + 30))
            r = rng.random()
            if r < 0.05:
                l2 = int(rng.integers(20, 30))
            elif r < 0.3:
                l2 = int(rng.integers(30, L + 30))
        fw = frag
        rv = revcomp(frag)
        if a.hic:
            # second locus for the mate; ligation junction inside a read with p=0.35
            cj = int(rng.choice(len(chroms), p=cprob))
            cc = chroms[cj]
            sj = int(rng.integers(0, max(1, len(cc) - fl)))
            other = cc[sj:sj + fl]
            if rng.random() < 0.5:
                other = revcomp(other)
            rv = other
            if rng.random() < 0.35:
                cut = int(rng.integers(25, max(26, L - 25)))
                if rng.random() < 0.5:
                    fw = np.concatenate([fw[:cut], revcomp(other)[:max(0, fl - cut)]])
                else:
                    rv = np.concatenate([rv[:cut], revcomp(frag)[:max(0, fl - cut)]])
        r1 = np.concatenate([fw, ADAPTER1])[:l1] if len(fw) < l1 else fw[:l1]
        r2 = np.concatenate([rv, ADAPTER2])[:l2] if len(rv) < l2 else rv[:l2]
        r1 = mutate(rng, r1, a.sub, a.indel)
        r2 = mutate(rng, r2, a.sub, a.indel)
        if rng.random() < a.junk_frac:
            r1 = ACGT[rng.integers(0, 4, size=len(r1))]
            if rng.random() < 0.5:
                r2 = ACGT[rng.integers(0, 4, size=len(r2))]
        if rng.random() < a.n_frac:
            r1 = r1.copy()
            r1[rng.integers(0, len(r1))] = ord("N")
        if rng.random() < a.n_frac:
            r2 = r2.copy()
            r2[rng.integers(0, len(r2))] = ord("N")
        if rng.random() < 0.5:
            r1, r2 = r2, r1
        # reads are upper-cased like a sequencer would emit (reference keeps its case)
        s1 = r1.tobytes().upper()
        s2 = r2.tobytes().upper()
        q1 = b"I" * len(s1)
        q2 = b"I" * len(s2)
        f1.write(b"@r%d/1\n%s\n+\n%s\n" % (i, s1, q1))
        f2.write(b"@r%d/2\n%s\n+\n%s\n" % (i, s2, q2))
        if fb is not None:
            b = wl[int(rng.choice(a.barcodes, p=wl_p))].copy()
            qual = np.full(16, ord("I"), dtype=np.uint8)
            r = rng.random()
            if r < 0.10:
                p = int(rng.integers(0, 16))
                b[p] = ACGT[rng.integers(0, 4)]
                qual[p] = ord("#") if rng.random() < 0.5 else ord("5")
            elif r < 0.12:
                b[int(rng.integers(0, 16))] = ord("N")
            elif r < 0.14:
                b = ACGT[rng.integers(0, 4, size=16)]
            fb.write(b"@r%d\n%s\n+\n%s\n" % (i, b.tobytes(), qual.tobytes()))
    f1.close()
    f2.close()
    if fb is not None:
        fb.close()


if __name__ == "__main__":
    main()
