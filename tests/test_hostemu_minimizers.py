"""The branch-light closed form of the w = 7 minimizer pass (cm_minimizers_w7_oddk, what k_prep_mm and the
two-pass kernels run) against the oracle's transcription of MinimizerGenerator::GenerateMinimizers
(minimizer_generator.cc:7-139) on reads built to hit its corner cases: equal hashes inside a window
(homopolymers, short tandem repeats), the first-window rule, reads shorter than one window, N runs,
lower case, even k (state machine)."""
import ctypes as C

import numpy as np
import pytest

import hostemu_lib as he
import oracle_lib as ol


def _reads(rng, n):
    out = []
    for i in range(n):
        kind = i % 8
        ln = int(rng.integers(1, 130)) if kind != 7 else int(rng.integers(17, 31))
        if kind in (0, 7):
            s = rng.choice(list(b"ACGT"), ln)
        elif kind == 1:  # two-letter alphabet: many equal canonical k-mers
            s = rng.choice(list(b"AT"), ln)
        elif kind == 2:  # tandem repeat of period 1..6 with a few substitutions
            unit = rng.choice(list(b"ACGT"), int(rng.integers(1, 7)))
            s = np.resize(unit, ln).copy()
            for _ in range(int(rng.integers(0, 4))):
                s[int(rng.integers(0, ln))] = rng.choice(list(b"ACGT"))
        elif kind == 3:  # homopolymer stretches
            s = np.concatenate([np.full(int(rng.integers(1, 40)), rng.choice(list(b"ACGT"))) for _ in range(6)])[:ln]
        elif kind == 4:  # random with N runs
            s = rng.choice(list(b"ACGT"), ln)
            for _ in range(int(rng.integers(1, 4))):
                a = int(rng.integers(0, ln))
                s[a:a + int(rng.integers(1, 5))] = ord("N")
        elif kind == 5:  # lower case mixed in (CharToUint8 maps it like upper case)
            s = rng.choice(list(b"ACGTacgt"), ln)
        else:  # repeat + N
            unit = rng.choice(list(b"ACGT"), int(rng.integers(1, 5)))
            s = np.resize(unit, ln).copy()
            s[int(rng.integers(0, ln))] = ord("n")
        out.append(np.asarray(s, dtype=np.uint8))
    return out


@pytest.mark.parametrize("k", [17, 19, 23, 15, 27, 16, 18])
def test_w7_minimizers_equal_the_reference_state_machine(k):
    L = he.lib()
    O = ol.lib()
    f = L.hostemu_minimizers_w7
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_uint32]
    rng = np.random.default_rng(1000 + k)
    oh, ot = np.zeros(256, np.uint64), np.zeros(256, np.uint64)
    gh, gp = np.zeros(256, np.uint64), np.zeros(256, np.uint32)
    n_emitted = 0
    for s in _reads(rng, 24000):
        buf = np.concatenate([s, np.zeros(8, np.uint8)])
        c = O.ora_minimizers(buf.ctypes.data_as(C.c_char_p), len(s), 0, k, 7, oh.ctypes.data, ot.ctypes.data)
        want = [(int(oh[i]), int(ot[i]) & 0x1FFFFFFFF) for i in range(c)]
        for variant in (1, 0):
            g = f(buf.ctypes.data, len(s), k, variant, gh.ctypes.data, gp.ctypes.data, 256)
            got = [(int(gh[i]), int(gp[i])) for i in range(g)]
            assert got == want, (variant, bytes(s), got, want)
        n_emitted += c
    assert n_emitted > 50000


@pytest.mark.parametrize("k,w", [(17, 5), (21, 11), (15, 3), (16, 10), (27, 7), (19, 1), (23, 32)])
def test_runtime_window_minimizers_equal_the_reference_state_machine(k, w):
    """cm_minimizers_core with a runtime window size (indexes built with -w other than 7) against the oracle"""
    L = he.lib()
    O = ol.lib()
    f = L.hostemu_minimizers_ring
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_uint32]
    rng = np.random.default_rng(3000 + 100 * k + w)
    oh, ot = np.zeros(512, np.uint64), np.zeros(512, np.uint64)
    gh, gp = np.zeros(512, np.uint64), np.zeros(512, np.uint32)
    n_emitted = 0
    for s in _reads(rng, 8000):
        buf = np.concatenate([s, np.zeros(8, np.uint8)])
        c = O.ora_minimizers(buf.ctypes.data_as(C.c_char_p), len(s), 0, k, w, oh.ctypes.data, ot.ctypes.data)
        g = f(buf.ctypes.data, len(s), k, w, gh.ctypes.data, gp.ctypes.data, 512)
        assert g == c, (bytes(s), g, c)
        assert [(int(gh[i]), int(gp[i])) for i in range(g)] == [(int(oh[i]), int(ot[i]) & 0x1FFFFFFFF) for i in range(c)], bytes(s)
        n_emitted += c
    assert n_emitted > 5000


@pytest.mark.parametrize("k", [17, 19, 23, 15, 25])
def test_position_parallel_minimizers_equal_the_reference_state_machine(k):
    """the per-lane pieces of k_prep_flat (2-bit packing, k-mer extraction, strand / hash, sliding-extrema selection,
    and the cases it hands to the sequential code) against the oracle's state machine"""
    L = he.lib()
    O = ol.lib()
    f = L.hostemu_minimizers_flat
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_int)]
    rng = np.random.default_rng(7000 + k)
    oh, ot = np.zeros(256, np.uint64), np.zeros(256, np.uint64)
    gh, gp = np.zeros(256, np.uint64), np.zeros(256, np.uint32)
    n_flat = n_fallback = 0
    for s in _reads(rng, 24000):
        buf = np.concatenate([s, np.zeros(8, np.uint8)])
        c = O.ora_minimizers(buf.ctypes.data_as(C.c_char_p), len(s), 0, k, 7, oh.ctypes.data, ot.ctypes.data)
        want = [(int(oh[i]), int(ot[i]) & 0x1FFFFFFFF) for i in range(c)]
        path = C.c_int(0)
        g = f(buf.ctypes.data, len(s), k, gh.ctypes.data, gp.ctypes.data, 256, C.byref(path))
        got = [(int(gh[i]), int(gp[i])) for i in range(g)]
        assert got == want, (path.value, bytes(s), got, want)
        if path.value == 0:
            n_flat += 1
        else:
            n_fallback += 1
    assert n_flat > 8000 and n_fallback > 1000  # both routes exercised


@pytest.mark.parametrize("k", [17, 21, 27])
def test_block_form_at_every_read_length(k):
    """the block-of-seven selection (cm_minimizers_w7_oddk) at every read length from one k-mer to five blocks and a bit -- the
    first-window rule alone in its block, last blocks of one to seven k-mers, the empty closing block -- on sequences that tie
    (homopolymer, period 2 and 3), random ones, and random ones with the first window's tie forced"""
    L = he.lib()
    O = ol.lib()
    f = L.hostemu_minimizers_w7
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_uint32]
    rng = np.random.default_rng(9000 + k)
    oh, ot = np.zeros(256, np.uint64), np.zeros(256, np.uint64)
    gh, gp = np.zeros(256, np.uint64), np.zeros(256, np.uint32)
    n = 0
    for ln in range(k, k + 40):
        seqs = [np.full(ln, ord("A"), np.uint8), np.resize(np.frombuffer(b"AC", np.uint8), ln).copy(), np.resize(np.frombuffer(b"ACG", np.uint8), ln).copy()]
        for _ in range(12):
            seqs.append(rng.choice(list(b"ACGT"), ln).astype(np.uint8))
        for _ in range(6):  # the seventh k-mer a copy of one of the first six: period p in the first k + 6 bases
            p = int(rng.integers(1, 7))
            s = rng.choice(list(b"ACGT"), ln).astype(np.uint8)
            head = min(ln, k + 6)
            s[:head] = np.resize(s[:p], head)
            seqs.append(s)
        for s in seqs:
            buf = np.concatenate([s, np.zeros(8, np.uint8)])
            c = O.ora_minimizers(buf.ctypes.data_as(C.c_char_p), len(s), 0, k, 7, oh.ctypes.data, ot.ctypes.data)
            want = [(int(oh[i]), int(ot[i]) & 0x1FFFFFFFF) for i in range(c)]
            g = f(buf.ctypes.data, len(s), k, 1, gh.ctypes.data, gp.ctypes.data, 256)
            assert [(int(gh[i]), int(gp[i])) for i in range(g)] == want, (ln, bytes(s))
            n += c
    assert n > 3000
