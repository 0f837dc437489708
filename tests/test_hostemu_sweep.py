"""The cluster sweep cut at its state-free breaks (cm_sweep_cluster, used by the cooperative kernel for long hit
lists, k_s3b_heavy) against the sequential sweep (cm_sweep_strided = CandidateProcessor::GenerateCandidatesOnOneStrand,
candidate_processor.cc:283-342) on adversarial sorted hit lists: dense diagonals, repeated hits, runs that exceed the
minimizer count (the state-dependent third break), several reference sequences, positions near the u32 wrap."""
import ctypes as C

import numpy as np

import hostemu_lib as he


def test_sweep_by_local_clusters_equals_sequential_sweep():
    L = he.lib()
    L.hostemu_sweep_clusters.restype = C.c_int
    L.hostemu_sweep_clusters.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_uint32]
    rng = np.random.default_rng(20260927)
    for it in range(4000):
        n = int(rng.integers(1, 400))
        e = int(rng.choice([1, 3, 8, 15]))
        mode = it % 5
        if mode == 0:    # clusters around a few loci, many exact repeats of a hit
            loci = rng.integers(0, 5000, size=int(rng.integers(1, 8)))
            pos = rng.choice(loci, n) + rng.integers(0, e + 2, n) * rng.integers(0, 2, n)
        elif mode == 1:  # one long dense diagonal: every gap <= e, longer than any minimizer count
            pos = np.cumsum(rng.integers(0, e + 1, n))
        elif mode == 2:  # sparse
            pos = rng.integers(0, 1 << 20, n)
        elif mode == 3:  # around the u32 wrap of pos + e
            pos = (1 << 32) - 1 - rng.integers(0, 3 * e + 4, n)
        else:            # gaps of exactly e and e + 1
            pos = np.cumsum(rng.choice([0, 1, e, e + 1], n))
        rid = rng.integers(0, 3, n) if mode != 1 else np.zeros(n, np.int64)
        h = np.sort((rid.astype(np.uint64) << np.uint64(32)) | (pos.astype(np.uint64) & np.uint64(0xffffffff)))
        for req, nm in ((1, 3), (2, 8), (2, 34), (1, 1), (3, 5)):
            rc = L.hostemu_sweep_clusters(h.ctypes.data, n, e, req, nm)
            assert rc == 0, (it, mode, n, e, req, nm, rc)
