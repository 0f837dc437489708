"""The cooperative stage functions (chromap_amd/csrc/cm_coop.h: a group of lanes per read with a long hit / candidate
list) run on the CPU with one OS thread per lane (tests/hostemu/emu_group.h) and compared with the oracle on the
repeat-rich fuzz data: same records, same counters.  Group sizes 16 / 64 / 256, and table sizes small enough that the
'decline' path (more runs than the tables hold -> the one-lane definition) is taken too."""
import ctypes as C

import pytest

import fuzz_data
import hostemu_lib as he
from test_hostemu_fuzz import run_case


def _set(L, G, thr, P, MM, RB, reverse=0, slab=0):
    L.hostemu_set_coop.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
    L.hostemu_set_coop(G, thr, P, MM, RB)
    L.hostemu_set_coop_order(int(reverse))
    L.hostemu_set_coop_slab.argtypes = [C.c_uint32]
    L.hostemu_set_coop_slab(int(slab))


def _items(L):
    a = (C.c_ulonglong * 12)()
    L.hostemu_coop_items(a)
    return list(a)


# (group size, list length above which a read goes to the group, hits the work area holds, minimizer table, run table)
# + whether the emulated lanes take their turns in descending order
# + the entries of the global-memory slab that lists longer than the work area use (0: such lists are declined)
GEOMETRIES = [(64, 8, 8192, 64, 130, 0, 0), (256, 16, 8192, 64, 130, 1, 0), (16, 8, 2048, 64, 130, 1, 0), (64, 8, 700, 5, 11, 1, 0),
              (1024, 64, 8192, 64, 130, 0, 0), (64, 8, 40, 64, 130, 1, 20000)]


# every fuzz configuration through wave-sized groups; the other group sizes, the decline path and the global-memory slab on
# one of them each.  Half the pairs of the plain fuzz test: an emulated group costs milliseconds per read.
def _small(cfg):
    seed, preset, kw, gen = cfg
    return (seed, preset, kw, dict(gen, pairs=1500))


CASES = [(_small(c), GEOMETRIES[0]) for c in fuzz_data.CONFIGS[:4]] + [(_small(fuzz_data.CONFIGS[0]), GEOMETRIES[1]),
                                                                        (_small(fuzz_data.CONFIGS[1]), GEOMETRIES[2]),
                                                                        (_small(fuzz_data.CONFIGS[0]), GEOMETRIES[3]),
                                                                        (_small(fuzz_data.CONFIGS[0]), GEOMETRIES[5])]


@pytest.mark.parametrize("cfg,geo", CASES, ids=["%d-G%d_P%d_MM%d%s" % (c[0], g[0], g[2], g[3], "_slab" if g[6] else "") for c, g in CASES])
def test_cooperative_stages_equal_oracle(cfg, geo, tmp_path):
    L = he.lib()

    def factory(idx, fa, preset, gkw, b1, o1, b2, o2):
        h = he.HostEmu(idx, fa, he.params(preset, **gkw))
        _set(L, *geo)
        try:
            rec, k, st, _ = h.map_pairs(b1, o1, b2, o2)
            items = _items(L)
        finally:
            _set(L, 0, 0, 0, 0, 0, 0, 0)
        factory.items = items
        return rec, k, st.as_dict()
    run_case(factory, cfg, tmp_path)
    done, declined = factory.items[0], factory.items[1]
    if cfg[0] == 3 and geo[1] >= 64:
        return  # this configuration's seed-frequency caps leave no list that long
    assert done > 0, "no read went through the cooperative hit-list stage"
    if cfg[1] != "hic":  # split alignment never supplements from the mate
        assert factory.items[2] > 0, "no read went through the cooperative rescue stage"
        assert factory.items[3] > 0, "no pair went through the cooperative pair filter"
        assert factory.items[4] > 0, "no read went through the cooperative acceptance stage"
        assert factory.items[5] > 0, "no pair went through the cooperative pairing stage"
        assert factory.items[7] > 0, "no read had its rescue searches run by a group"
        assert factory.items[8] > 0, "no search's hits were kept in the pool for the fill pass"
        if geo[1] <= 16:  # (thresholds of 64 and more leave no multi-mapped pair with lists that long in these cases)
            assert factory.items[6] > 0, "no multi-mapped pair went through the cooperative location of its sampled pairings"
    if geo[3] < 10:
        assert declined > 0, "the decline path was not taken"


def test_hit_list_with_a_diagonal_before_the_sequence_start(tmp_path):
    """reads whose repeat element also stands at the very start of a chromosome, behind 15 bases that belong to nothing: the + hit there
    has its diagonal before position 0 (the only way an occurrence run is not ascending in candidate positions) -- the group declines the
    read (cm_coop_s3b_expand) and the one-lane definition takes it; results equal the oracle's"""
    import oracle_lib as ol
    from test_hostemu_fuzz import _tuples_g, _tuples_o
    fa, b1, o1, b2, o2 = fuzz_data.write_wrap_case(str(tmp_path))
    kw = {"mapq_threshold": 0}
    o = ol.Oracle(None, fa, ol.params("atac", **kw))
    idx = str(tmp_path / "w.idx")
    assert o.L.ora_index_save(idx.encode(), C.byref(o.idx)) == 0
    orec, ok, ost, _ = o.map_pairs(b1, o1, b2, o2)
    L = he.lib()
    h = he.HostEmu(idx, fa, he.params("atac", **kw))
    _set(L, *GEOMETRIES[0])
    try:
        rec, k, st, _ = h.map_pairs(b1, o1, b2, o2)
        items = _items(L)
    finally:
        _set(L, 0, 0, 0, 0, 0, 0, 0)
    assert k == ok and _tuples_g(rec, k, False) == _tuples_o(orec, ok, False)
    assert items[1] > 0, "no read was declined: the wrapped diagonal was not met"
    o.close()


def test_hit_lists_on_64_bit_keys(tmp_path):
    """the cooperative hit-list stage with the reference's 64-bit keys (what a reference beyond 32-bit global coordinates gets;
    every other case here runs on 32-bit keys, as the device does when the reference fits)"""
    L = he.lib()
    cfg, geo = _small(fuzz_data.CONFIGS[1]), GEOMETRIES[0]

    def factory(idx, fa, preset, gkw, b1, o1, b2, o2):
        L.hostemu_set_coop_key32(0)
        try:
            h = he.HostEmu(idx, fa, he.params(preset, **gkw))
            _set(L, *geo)
            rec, k, st, _ = h.map_pairs(b1, o1, b2, o2)
            factory.items = _items(L)
        finally:
            L.hostemu_set_coop_key32(1)
            _set(L, 0, 0, 0, 0, 0, 0, 0)
        return rec, k, st.as_dict()
    run_case(factory, cfg, tmp_path)
    assert factory.items[0] > 0


def test_rescue_searches_with_small_and_full_tables(tmp_path):
    """the rescue searches' two table sizes (k_s4a/4b_rescue_wave<true / false>): small tables made tiny, so that searches take
    several rounds in them and others are handed to the full tables"""
    L = he.lib()
    cfg, geo = _small(fuzz_data.CONFIGS[0]), GEOMETRIES[0]

    def factory(idx, fa, preset, gkw, b1, o1, b2, o2):
        h = he.HostEmu(idx, fa, he.params(preset, **gkw))
        _set(L, *geo)
        L.hostemu_set_rescue_small.argtypes = [C.c_uint32, C.c_uint32]
        L.hostemu_set_rescue_small(6, 40)
        try:
            rec, k, st, _ = h.map_pairs(b1, o1, b2, o2)
            factory.items = _items(L)
        finally:
            L.hostemu_set_rescue_small(0, 0)
            _set(L, 0, 0, 0, 0, 0, 0, 0)
        return rec, k, st.as_dict()
    run_case(factory, cfg, tmp_path)
    assert factory.items[10] > 0 and factory.items[11] > 0, factory.items


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
        P = int(rng.choice([64, 1024])) if it % 4 == 0 else 1024
        RB = int(rng.choice([3, 90]))
        c0 = np.ascontiguousarray(c0, np.uint64)
        hits = np.ascontiguousarray(hits, np.uint64)
        L.hostemu_set_coop_slab.argtypes = [C.c_uint32]
        L.hostemu_set_coop_slab(2000 if it % 3 == 0 else 0)  # lists longer than P: on the slab, or (no slab) by one lane
        rc = f(c0.ctypes.data, c0c.ctypes.data, len(c0), hits.ctypes.data, len(hits), e, int(rng.integers(1, 12)), G, P, RB, it & 1)
        L.hostemu_set_coop_slab(0)
        assert rc == 0, (it, mode, e, len(c0), len(hits), G, P, RB, rc)


def test_cooperative_rescue_search_on_adversarial_runs():
    """cm_coop_rescue (a group per read: merged windows once, bounds of every (minimizer, window) pair side by side, the chain of
    binary searches replayed on indices) against cm_rescue: the same hits in the same order, count, repetitive-seed length and
    return value -- bail-outs included"""
    L = he.lib()
    f = L.hostemu_rescue_search_check
    f.restype = C.c_int
    f.argtypes = [C.c_uint64, C.c_uint32, C.c_int, C.c_int]
    L.hostemu_rescue_replays.restype = C.c_uint64
    r0 = L.hostemu_rescue_replays()
    for seed, G, rev in ((1, 64, 0), (2, 16, 1), (3, 256, 0), (4, 64, 1)):
        assert f(seed, 400, G, rev) == 0, (seed, G, rev)
    stats = (C.c_uint64 * 4)()
    L.hostemu_rescue_search_stats(stats)
    assert stats[0] > 50 * 4
    # the chain of searches goes through the transition tables alone (no table has been seen not to apply: r1 == r0 here); the
    # window-by-window replay that backs them up is taken by decree for every second minimizer, with the same results
    r1 = L.hostemu_rescue_replays()
    L.hostemu_force_rescue_replay(1)
    try:
        for seed, G, rev in ((5, 64, 0), (6, 16, 1), (7, 256, 1)):
            assert f(seed, 300, G, rev) == 0, (seed, G, rev)
    finally:
        L.hostemu_force_rescue_replay(0)
    assert L.hostemu_rescue_replays() - r1 > 1000, (r0, r1, L.hostemu_rescue_replays())


def test_cooperative_pair_filter_on_adversarial_lists():
    """cm_coop_reduce_dir against cm_reduce_dir: lists that pair densely, sparsely or not at all, more than five unpaired
    entries with qualifying counts (the first-five rule), counts around the running maxima, several sequences, one list
    running out first (entries never looked at), empty lists."""
    import numpy as np
    L = he.lib()
    f = L.hostemu_reduce_dir_check
    f.restype = C.c_int
    f.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int]
    rng = np.random.default_rng(99)
    for it in range(1500):
        dist = int(rng.choice([0, 5, 300, 1000]))
        mode = it % 5
        n1 = int(rng.integers(0, 400))
        n2 = int(rng.integers(0, 400))
        span = {0: 3000, 1: 200000, 2: 1 << 22, 3: 40000, 4: 3000}[mode]
        nrid = 1 if mode in (0, 3) else 4

        def mk(n, shift):
            pos = rng.integers(0, span, n) + shift
            rid = rng.integers(0, nrid, n).astype(np.uint64)
            k = np.unique((rid << np.uint64(32)) | pos.astype(np.uint64))
            c = rng.integers(1, 12 if mode != 3 else 9, len(k)).astype(np.uint8)
            return np.ascontiguousarray(k), c
        p1, c1 = mk(n1, 0)
        p2, c2 = mk(n2, 0 if mode != 4 else span + 2 * dist + 5)  # mode 4: list 2 entirely beyond list 1
        if it % 7 == 0:
            p1, c1, p2, c2 = p2, c2, p1, c1
        G = int(rng.choice([16, 64, 256]))
        rc = f(dist, p1.ctypes.data, c1.ctypes.data, len(p1), p2.ctypes.data, c2.ctypes.data, len(p2), G, it & 1)
        assert rc == 0, (it, mode, dist, len(p1), len(p2), G, rc)


def test_cooperative_acceptance_loop_on_adversarial_lists():
    """cm_coop_draft_strand against cm_draft_strand: candidate lists sorted by count, rejected candidates at every place
    of a verification group, invalid positions between them, count ties around the threshold, lists shorter than a group,
    no grouping at all (lanes = 0)."""
    import numpy as np
    L = he.lib()
    f = L.hostemu_draft_strand_check
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_uint32,
                  C.c_int, C.c_int]
    rng = np.random.default_rng(5)
    ref_len = np.array([100000, 5000], np.uint32)
    for it in range(2500):
        e = int(rng.choice([3, 8]))
        lanes = int(rng.choice([0, 4, 8]))
        nc = int(rng.integers(1, 300)) if it % 10 else int(rng.integers(1, 7))
        rl = 50
        cc = np.sort(rng.integers(1, int(rng.choice([3, 9])), nc).astype(np.uint8))[::-1].copy()
        rid = rng.integers(0, 2, nc).astype(np.uint64)
        # positions: mostly valid, some at the sequence edges (invalid)
        pos = rng.integers(0, 6000, nc).astype(np.uint64)
        strand = int(it & 1)
        if strand:
            pos += rl
        cp = (rid << np.uint64(32)) | pos
        p_rej = float(rng.choice([0.0, 0.1, 0.5, 0.95]))
        ne = np.where(rng.random(nc) < p_rej, e + 1, rng.integers(0, e + 1, nc)).astype(np.int16)
        end = rng.integers(40, 60, nc).astype(np.int16)
        G = int(rng.choice([16, 64, 256]))
        rc = f(cp.ctypes.data, cc.ctypes.data, nc, ne.ctypes.data, end.ctypes.data, rl, strand, e, lanes, ref_len.ctypes.data, 2, G, (it >> 1) & 1)
        assert rc == 0, (it, e, lanes, nc, strand, G, rc)


def test_cooperative_pairing_on_adversarial_lists():
    """cm_coop_pair_dir against cm_pair_dir: sorted draft-mapping lists with equal positions, dense and sparse partner
    ranges, ties of the minimal error sum across directions (the first one in sweep order counts), empty lists; and
    cm_coop_pair_find (the want-th minimal-sum pairing in sweep order, what k_s6c_coop reports for a multi-mapped pair) against
    the counting sweep for every / a spread of the indices."""
    import numpy as np
    L = he.lib()
    f = L.hostemu_pairing_check
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p,
                  C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    rng = np.random.default_rng(6)
    for it in range(1500):
        e = int(rng.choice([3, 8]))
        span = int(rng.choice([2000, 50000, 1 << 21]))

        def mk(n):
            pos = np.sort(((rng.integers(0, 3, n).astype(np.uint64)) << np.uint64(32)) | rng.integers(1000, 1000 + span, n).astype(np.uint64))
            err = rng.integers(0, int(rng.choice([2, e + 1])), n).astype(np.int16)
            return np.ascontiguousarray(pos), err
        lists = [mk(int(rng.integers(0, 250))) for _ in range(4)]
        G = int(rng.choice([16, 64, 256]))
        rc = f(lists[0][0].ctypes.data, lists[0][1].ctypes.data, len(lists[0][0]), lists[1][0].ctypes.data, lists[1][1].ctypes.data, len(lists[1][0]),
               lists[2][0].ctypes.data, lists[2][1].ctypes.data, len(lists[2][0]), lists[3][0].ctypes.data, lists[3][1].ctypes.data, len(lists[3][0]),
               int(rng.integers(30, 51)), int(rng.integers(30, 51)), e, int(rng.choice([300, 1000, 2000])), 30, G, it & 1)
        assert rc == 0, (it, e, span, [len(x[0]) for x in lists], G, rc)


def test_cooperative_candidate_sort():
    """cm_coop_sort_cand (counting sort by count of a position-ordered list) against cm_sort_cand: few and many distinct
    counts, counts at and above the bin capacity (one-lane path), lists that are not in position order (one-lane path)"""
    import numpy as np
    L = he.lib()
    f = L.hostemu_sort_cand_check
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_int]
    rng = np.random.default_rng(12)
    for it in range(400):  # (800 until the round's end: two minutes of the CPU suite on a slow day of the build container)
        n = int(rng.integers(0, 1500))
        pos = np.unique((rng.integers(0, 3, n).astype(np.uint64) << np.uint64(32)) | rng.integers(0, 1 << 22, n).astype(np.uint64))
        if it % 11 == 0 and len(pos) > 3:
            pos = pos[rng.permutation(len(pos))]  # --chr-order re-ranking leaves lists out of position order
        cmax = int(rng.choice([2, 9, 40, 70, 130, 200, 255]))
        cnt = rng.integers(1, cmax + 1, len(pos)).astype(np.uint8)
        pos = np.ascontiguousarray(pos)
        rc = f(pos.ctypes.data, cnt.ctypes.data, len(pos), 64, int(rng.choice([16, 64, 256])), it & 1)
        assert rc == 0, (it, len(pos), cmax)
