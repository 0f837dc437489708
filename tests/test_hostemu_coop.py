"""The cooperative stage functions (chromap_amd/csrc/cm_coop.h: a group of lanes per read with a long hit / candidate
list) run on the CPU with one OS thread per lane (tests/hostemu/emu_group.h) and compared with the oracle on the
repeat-rich fuzz data: same records, same counters.  Group sizes 16 / 64 / 256, and table sizes small enough that the
'decline' path (more runs than the tables hold -> the one-lane definition) is taken too."""
import ctypes as C

import pytest

import fuzz_data
import hostemu_lib as he
from test_hostemu_fuzz import run_case


def _set(L, G, thr, P, MM, RB, reverse=0):
    L.hostemu_set_coop.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
    L.hostemu_set_coop(G, thr, P, MM, RB)
    L.hostemu_set_coop_order(int(reverse))


def _items(L):
    a = (C.c_ulonglong * 8)()
    L.hostemu_coop_items(a)
    return list(a)


# (group size, list length above which a read goes to the group, hits the work area holds, minimizer table, run table)
# + whether the emulated lanes take their turns in descending order
GEOMETRIES = [(64, 8, 8192, 64, 130, 0), (256, 16, 8192, 64, 130, 1), (16, 8, 2048, 64, 130, 1), (64, 8, 700, 5, 11, 1)]


@pytest.mark.parametrize("geo", GEOMETRIES, ids=["G%d_P%d_MM%d" % (g[0], g[2], g[3]) for g in GEOMETRIES])
@pytest.mark.parametrize("cfg", fuzz_data.CONFIGS[:4], ids=[str(c[0]) for c in fuzz_data.CONFIGS[:4]])
def test_cooperative_stages_equal_oracle(cfg, geo, tmp_path):
    L = he.lib()

    def factory(idx, fa, preset, gkw, b1, o1, b2, o2):
        h = he.HostEmu(idx, fa, he.params(preset, **gkw))
        _set(L, *geo)
        try:
            rec, k, st, _ = h.map_pairs(b1, o1, b2, o2)
            items = _items(L)
        finally:
            _set(L, 0, 0, 0, 0, 0, 0)
        factory.items = items
        return rec, k, st.as_dict()
    run_case(factory, cfg, tmp_path)
    done, declined = factory.items[0], factory.items[1]
    assert done > 0, "no read went through the cooperative hit-list stage"
    if cfg[1] != "hic":  # split alignment never supplements from the mate
        assert factory.items[2] > 0, "no read went through the cooperative rescue stage"
    if geo[3] < 10:
        assert declined > 0, "the decline path was not taken"


def test_cooperative_sort_sweep_merge_on_adversarial_lists():
    """cm_coop_rescue_dir against cm_sort_u64 + cm_sweep + cm_merge: dense chains (every gap <= e: one long greedy chain
    across all lanes' chunks), gaps of exactly e and e + 1, positions shared by both lists, several sequences, unsorted
    hits in few or many runs, lists longer than the work area (the one-lane path), empty sides."""
    import numpy as np
    L = he.lib()
    f = L.hostemu_rescue_dir_check
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_int, C.c_uint32, C.c_int, C.c_uint32, C.c_uint32, C.c_int]
    rng = np.random.default_rng(77)
    for it in range(600):
        e = int(rng.choice([0, 1, 4, 8]))
        mode = it % 6
        n1 = int(rng.integers(0, 300))
        cnt = int(rng.integers(1, 900))
        span = {0: 400, 1: 3000, 2: 1 << 20, 3: 2000, 4: 100000, 5: 800}[mode]
        if mode == 1:    # dense: gaps 0..e in the union
            base = np.cumsum(rng.integers(0, e + 2, n1 + cnt))
        elif mode == 3:  # gaps of exactly e / e + 1
            base = np.cumsum(rng.choice([e, e + 1, 1], n1 + cnt))
        else:
            base = rng.integers(0, span, n1 + cnt)
        rid = rng.integers(0, 2 if mode != 4 else 5, n1 + cnt).astype(np.uint64)
        keys = (rid << np.uint64(32)) | base.astype(np.uint64)
        c0 = np.unique(keys[:n1])
        if mode == 5 and len(c0):  # hits that coincide with the read's own candidates
            k = min(len(c0) // 2, cnt)
            keys[n1:n1 + k] = c0[:k]
        hits = keys[n1:].copy()
        nruns = int(rng.choice([1, 2, 7, 40]))
        # the hits as the fill pass leaves them: a few ascending runs
        parts = np.array_split(rng.permutation(len(hits)), nruns)
        hits = np.concatenate([np.sort(hits[p]) for p in parts]) if len(hits) else hits
        c0c = rng.integers(1, 9, len(c0)).astype(np.uint8)
        G = int(rng.choice([16, 64, 256]))
        P = int(rng.choice([64, 1024])) if it % 9 == 0 else 1024
        RB = int(rng.choice([3, 90]))
        c0 = np.ascontiguousarray(c0, np.uint64)
        hits = np.ascontiguousarray(hits, np.uint64)
        rc = f(c0.ctypes.data, c0c.ctypes.data, len(c0), hits.ctypes.data, len(hits), e, int(rng.integers(1, 12)), G, P, RB, it & 1)
        assert rc == 0, (it, mode, e, len(c0), len(hits), G, P, RB, rc)
