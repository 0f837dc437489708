"""Stage-level parity of the gfx950 build (not of a host compile of the stage functions): after a mapping call the
per-pair trace of the device pipeline -- read lengths after trimming, minimizer counts, candidates entering
verification, repetitive-seed length, draft mappings with the best / second-best error bookkeeping, pairing sums --
is downloaded (cmgpu_debug_trace) and compared with the oracle's trace (ora_set_trace) pair by pair, and every
read's minimizer list (hash, position, strand) with the oracle's transcription of GenerateMinimizers.  A kernel
change that breaks parity fails here at the stage that broke, not as "BED differs"."""
import ctypes as C

import numpy as np
import pytest

import datasets
import fuzz_data
import oracle_lib as ol

pytestmark = pytest.mark.gpu
FIELDS = [n for n, _ in ol.OraTrace._fields_]


def _check(g, o, b1, o1, b2, o2, k, label, minimizers=True):
    from chromap_amd import _capi
    n = len(o1) - 1
    g.upload(b1, o1, b2, o2)
    kk = g.map_resident()
    orec, ok, ost, tr = o.map_pairs(b1, o1, b2, o2, trace=True)
    gt = (ol.OraTrace * n)()  # cmgpu_trace has ora_trace's layout
    assert g.L.cmgpu_debug_trace(g.ctx, C.cast(gt, C.c_void_p), n) == 0, g.L.cmgpu_last_error(g.ctx)
    stage_of = {"len": "K0 trimming", "n_mm": "K1 minimizers", "n_cand": "K2/K3 probe, candidates, rescue, pair filter", "rep": "K3 repetitive seeds",
                "n_draft": "K4 verification", "min_err": "K4 verification", "nbest": "K4/K5 best mappings", "second": "K4 verification",
                "nsecond": "K4 verification", "min_sum": "K5 pairing", "second_sum": "K5 pairing", "force_mapq": "K3c supplement"}
    for i in range(n):
        a, b = gt[i], tr[i]
        for f in FIELDS:
            if getattr(a, f) != getattr(b, f):
                st = next(v for p, v in stage_of.items() if f.startswith(p))
                raise AssertionError("%s pair %d: %s = %d on the device, %d in the oracle (stage: %s)" % (label, i, f, getattr(a, f), getattr(b, f), st))
    assert kk == ok
    if not minimizers:
        return
    cnt = np.zeros(2 * n, np.uint32)
    off = np.zeros(2 * n, np.uint32)
    tot = C.c_uint64(0)
    cap = int(sum(t.n_mm1 + t.n_mm2 for t in tr)) + 16
    gh = np.zeros(cap, np.uint64)
    gp = np.zeros(cap, np.uint32)
    assert g.L.cmgpu_debug_minimizers_all(g.ctx, cnt.ctypes.data, off.ctypes.data, gh.ctypes.data, gp.ctypes.data, cap, C.byref(tot)) == 0
    O = ol.lib()
    oh, ot = np.zeros(1024, np.uint64), np.zeros(1024, np.uint64)
    for i in range(n):
        for mate, (b, of, ln) in enumerate(((b1, o1, tr[i].len1), (b2, o2, tr[i].len2))):
            if ln == 0:
                continue
            s = np.concatenate([b[of[i]:of[i] + ln], np.zeros(8, np.uint8)])
            c = O.ora_minimizers(s.ctypes.data_as(C.c_char_p), int(ln), 0, k, 7, oh.ctypes.data, ot.ctypes.data)
            r = 2 * i + mate
            assert cnt[r] == c, (label, i, mate)
            a0 = int(off[r])
            assert gh[a0:a0 + c].tolist() == oh[:c].tolist(), (label, i, mate, "hash")
            assert gp[a0:a0 + c].tolist() == [int(x) & 0x1FFFFFFFF for x in ot[:c]], (label, i, mate, "position / strand")


@pytest.mark.parametrize("case", ["s1_atac", "s4_atac_q0", "s3_chip", "h1_hic"])
def test_stage_trace_golden_cases(case):
    from chromap_amd import ChromapGPU
    meta = datasets.case_meta(case)
    fa, r1, r2 = datasets.case_inputs(case)
    idx = datasets.case_index(case)
    preset, kw = datasets.flags_to_params(meta["chromap_flags"])
    g = ChromapGPU(idx, fa, preset=preset, **kw)
    o = ol.Oracle(idx, fa, ol.params(preset, **kw))
    b1, o1 = ol.read_fastx(r1)
    b2, o2 = ol.read_fastx(r2)
    _check(g, o, b1, o1, b2, o2, 17, case)
    g.close()
    o.close()


@pytest.mark.parametrize("prep_kernel", [1, 0])
@pytest.mark.parametrize("cfg", fuzz_data.CONFIGS, ids=[str(c[0]) for c in fuzz_data.CONFIGS])
def test_stage_trace_fuzz(cfg, prep_kernel, tmp_path):
    """repeat-rich adversarial data (N runs, homopolymers, tandem repeats, short reads): both minimizer kernels"""
    from chromap_amd import ChromapGPU
    from test_hostemu_fuzz import CAPI_NAMES
    seed, preset, kw, gen = cfg
    fa, b1, o1, b2, o2 = fuzz_data.write_case(str(tmp_path), seed, **gen)
    o = ol.Oracle(None, fa, ol.params(preset, **kw))
    idx = str(tmp_path / "f.idx")
    assert o.L.ora_index_save(idx.encode(), C.byref(o.idx)) == 0
    g = ChromapGPU(idx, fa, preset=preset, **{CAPI_NAMES.get(k, k): v for k, v in kw.items()})
    g.set_option("prep_kernel", prep_kernel)
    _check(g, o, b1, o1, b2, o2, 17, "fuzz %s" % seed)
    g.close()
    o.close()


@pytest.mark.parametrize("cfg", fuzz_data.CONFIGS[:2], ids=[str(c[0]) for c in fuzz_data.CONFIGS[:2]])
def test_stage_trace_fuzz_64_bit_hit_keys(cfg, tmp_path, monkeypatch):
    """the cooperative hit-list stage on the reference's 64-bit keys (a reference too long for 32-bit global coordinates gets these;
    CM_NO_KEY32 asks for them on any reference)"""
    from chromap_amd import ChromapGPU
    from test_hostemu_fuzz import CAPI_NAMES
    monkeypatch.setenv("CM_NO_KEY32", "1")
    seed, preset, kw, gen = cfg
    fa, b1, o1, b2, o2 = fuzz_data.write_case(str(tmp_path), seed, **gen)
    o = ol.Oracle(None, fa, ol.params(preset, **kw))
    idx = str(tmp_path / "f.idx")
    assert o.L.ora_index_save(idx.encode(), C.byref(o.idx)) == 0
    g = ChromapGPU(idx, fa, preset=preset, **{CAPI_NAMES.get(k, k): v for k, v in kw.items()})
    _check(g, o, b1, o1, b2, o2, 17, "fuzz %s, 64-bit keys" % seed)
    g.close()
    o.close()


def test_hit_list_with_a_diagonal_before_the_sequence_start(tmp_path):
    """a + hit whose candidate position wraps below zero (fuzz_data.write_wrap_case): the cooperative kernel declines the read, the
    bitonic kernel takes it; every stage count equals the oracle's"""
    from chromap_amd import ChromapGPU
    fa, b1, o1, b2, o2 = fuzz_data.write_wrap_case(str(tmp_path))
    o = ol.Oracle(None, fa, ol.params("atac", mapq_threshold=0))
    idx = str(tmp_path / "w.idx")
    assert o.L.ora_index_save(idx.encode(), C.byref(o.idx)) == 0
    g = ChromapGPU(idx, fa, preset="atac", mapq_threshold=0)
    _check(g, o, b1, o1, b2, o2, 17, "wrapped diagonal")
    g.close()
    o.close()


@pytest.mark.parametrize("tile", [8, 32, 128])
def test_position_parallel_minimizers_tile_sizes(tile):
    """k_prep_flat with other tile geometries (reads per tile) on reads of mixed lengths incl. adapter-trimmed ones"""
    from chromap_amd import ChromapGPU
    case = "s2_atac_q0"
    meta = datasets.case_meta(case)
    fa, r1, r2 = datasets.case_inputs(case)
    idx = datasets.case_index(case)
    preset, kw = datasets.flags_to_params(meta["chromap_flags"])
    g = ChromapGPU(idx, fa, preset=preset, **kw)
    g.set_option("prep_tile_reads", tile)
    o = ol.Oracle(idx, fa, ol.params(preset, **kw))
    b1, o1 = ol.read_fastx(r1)
    b2, o2 = ol.read_fastx(r2)
    _check(g, o, b1, o1, b2, o2, 17, "%s tile %d" % (case, tile))
    g.close()
    o.close()
