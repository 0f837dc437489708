"""Adapter trimming (K0) in isolation: the device stage function (2-bit fast path for upper-case
ACGT reads, byte-wise definition otherwise) against the oracle's restatement of
Chromap::TrimAdapterForPairedEndRead (chromap.cc:176-289) on adversarial pairs: overlaps of every
length around the 30-base minimum, mismatches inside and outside both seeds, N and lower-case
bases, unequal mate lengths (swap branch), reads longer than 64 / 160 / 256 bases."""
import ctypes as C

import numpy as np
import pytest

import datasets
import hostemu_lib as he
import oracle_lib as ol

COMP = bytes.maketrans(b"ACGTNacgtn", b"TGCANtgcan")
AD1 = b"CTGTCTCTTATACACATCTCCGAGCCCACGAGACTAAGGCGAATCTCGTATGCCGTCTTCTGCTTG" * 6
AD2 = b"CTGTCTCTTATACACATCTGACGCTGCCGACGAGTGTAGATCTCGGTGGTCGCCGTATCATTAAAA" * 6


def make_pairs(seed, n):
    rng = np.random.default_rng(seed)
    r1s, r2s = [], []
    for i in range(n):
        big = rng.random() < 0.15
        l1 = int(rng.integers(30, 330 if big else 80))
        l2 = l1 if rng.random() < 0.5 else int(rng.integers(30, 330 if big else 80))
        fl = int(rng.integers(10, max(l1, l2) + 40))
        frag = bytes(rng.choice(list(b"ACGT"), fl).astype(np.uint8))
        a = bytearray((frag + AD1)[:l1])
        b = bytearray((frag.translate(COMP)[::-1] + AD2)[:l2])
        u = rng.random()
        nmut = 0 if u < 0.4 else 1 if u < 0.7 else 2 if u < 0.9 else 3
        for _ in range(nmut):
            t = a if rng.random() < 0.5 else b
            p = int(rng.integers(0, min(len(t), fl + 3)))
            t[p] = ord("ACGT"[(b"ACGT".index(bytes([t[p]]).upper()) + 1 + int(rng.integers(0, 3))) % 4]) if bytes([t[p]]).upper() in b"ACGT" else t[p]
        v = rng.random()
        if v < 0.08:
            t = a if rng.random() < 0.5 else b
            t[int(rng.integers(0, len(t)))] = ord("N")
        elif v < 0.14:
            t = a if rng.random() < 0.5 else b
            p = int(rng.integers(0, len(t)))
            t[p] = ord(chr(t[p]).lower())
        elif v < 0.17:
            a = bytearray(bytes(a).lower())
        # low-complexity fragments give several seed occurrences
        if rng.random() < 0.05:
            a = bytearray((b"AC" * 200)[:l1])
            b = bytearray((b"GT" * 200)[:l2])
        r1s.append(bytes(a))
        r2s.append(bytes(b))
    return r1s, r2s


def pack(ss):
    off = np.zeros(len(ss) + 1, np.uint32)
    off[1:] = np.cumsum([len(s) for s in ss])
    return np.frombuffer(b"".join(ss), np.uint8).copy(), off


@pytest.mark.parametrize("seed,minlen", [(1, 30), (2, 30), (3, 31), (4, 40), (5, 64)])
def test_trim_lengths_match_oracle(seed, minlen):
    from chromap_amd import _capi
    r1s, r2s = make_pairs(seed, 4000)
    b1, o1 = pack(r1s)
    b2, o2 = pack(r2s)
    fa, _, _ = datasets.case_inputs("toy_atac")
    o = ol.Oracle(datasets.case_index("toy_atac"), fa, ol.params("atac", min_read_length=minlen))
    _, _, _, tr = o.map_pairs(b1, o1, b2, o2, trace=True)
    L = he.lib()
    p = _capi.default_params("atac", min_read_length=minlen)
    n = len(r1s)
    bt = _capi.Batch(n, 0, b1.ctypes.data, o1.ctypes.data, b2.ctypes.data, o2.ctypes.data)
    rlen = np.zeros(2 * n, np.uint32)
    L.hostemu_trim.argtypes = [C.POINTER(_capi.Params), C.POINTER(_capi.Batch), C.c_void_p]
    assert L.hostemu_trim(C.byref(p), C.byref(bt), rlen.ctypes.data) == 0
    trimmed = 0
    for i in range(n):
        if len(r1s[i]) < minlen or len(r2s[i]) < minlen:
            assert (rlen[2 * i], rlen[2 * i + 1]) == (0, 0)
            continue
        assert (rlen[2 * i], rlen[2 * i + 1]) == (tr[i].len1, tr[i].len2), (i, r1s[i], r2s[i])
        trimmed += rlen[2 * i] != len(r1s[i]) or rlen[2 * i + 1] != len(r2s[i])
    assert trimmed > 200
    o.close()


def test_swar_hamming_equals_byte_loop():
    """cm_hamming_diag (8 bytes at a time, reversed + complemented for the - strand) against the loop
    it replaces: every alignment of both pointers, lengths 1..150, N / lower case / other bytes"""
    L = he.lib()
    L.hostemu_hamming.restype = C.c_int
    L.hostemu_hamming.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
    rng = np.random.default_rng(5)
    alphabet = np.frombuffer(b"ACGTACGTACGTACGTNacgtn*", np.uint8)
    comp = bytes.maketrans(b"ACGTNacgtn", b"TGCANTGCAN")
    for trial in range(3000):
        Lfull = int(rng.integers(1, 151))
        toff = int(rng.integers(0, Lfull)) if rng.random() < 0.3 else 0
        Ln = int(rng.integers(1, Lfull - toff + 1))
        neg = int(rng.integers(0, 2))
        read = alphabet[rng.integers(0, len(alphabet), Lfull)]
        text = bytes(read) if not neg else bytes(read).translate(comp)[::-1]
        text = text[toff:toff + Ln]
        pat = bytearray(text if rng.random() < 0.7 else bytes(alphabet[rng.integers(0, len(alphabet), Ln)]))
        for p in rng.integers(0, Ln, int(rng.integers(0, 4))):
            pat[p] = int(alphabet[rng.integers(0, len(alphabet))])
        # buffers with padding and arbitrary alignment of both pointers
        oa, ob = int(rng.integers(0, 8)), int(rng.integers(0, 8))
        bufp = np.zeros(oa + Ln + 32, np.uint8)
        bufr = np.zeros(ob + Lfull + 32, np.uint8)
        bufp[oa:oa + Ln] = np.frombuffer(bytes(pat), np.uint8)
        bufr[ob:ob + Lfull] = read
        base_p = (bufp.ctypes.data + 7) & ~7
        base_r = (bufr.ctypes.data + 7) & ~7
        # re-place the data at the chosen offsets from an 8-aligned base inside the arrays
        pp = np.zeros(Ln + 48, np.uint8); rr = np.zeros(Lfull + 48, np.uint8)
        sp = (-pp.ctypes.data) % 8 + oa; sr = (-rr.ctypes.data) % 8 + 8 + ob
        pp[sp:sp + Ln] = np.frombuffer(bytes(pat), np.uint8); rr[sr:sr + Lfull] = read
        naive = C.c_int(0)
        got = L.hostemu_hamming(pp.ctypes.data + sp, rr.ctypes.data + sr, Lfull, neg, toff, Ln, C.byref(naive))
        assert got == naive.value, (trial, Lfull, toff, Ln, neg)
