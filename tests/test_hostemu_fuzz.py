"""Oracle vs the product's stage functions (host emulation) on adversarial repeat-rich data."""
import ctypes as C

import numpy as np
import pytest

import fuzz_data
import hostemu_lib as he
import oracle_lib as ol

CAPI_NAMES = {"max_seed_freq0": "max_seed_frequency0", "max_seed_freq1": "max_seed_frequency1"}


def _tuples_o(rec, k, hic):
    if hic:
        pr = C.cast(rec, C.POINTER(ol.OraPairsRecord))
        return sorted((pr[i].read_id, pr[i].rid1, pr[i].rid2, pr[i].pos1, pr[i].pos2, pr[i].strand1, pr[i].strand2, pr[i].mapq,
                       pr[i].is_unique) for i in range(k))
    return sorted((rec[i].read_id, rec[i].rid, rec[i].fragment_start, rec[i].fragment_length, rec[i].mapq, rec[i].direction,
                   rec[i].is_unique, rec[i].pos_aln_len, rec[i].neg_aln_len) for i in range(k))


def _tuples_g(rec, k, hic):
    from chromap_amd import _capi
    if hic:
        pr = C.cast(rec, C.POINTER(_capi.PairsRecord))
        return sorted((pr[i].read_id, pr[i].rid1, pr[i].rid2, pr[i].pos1, pr[i].pos2, pr[i].strand1, pr[i].strand2, pr[i].mapq,
                       pr[i].is_unique) for i in range(k))
    return sorted((rec[i].read_id, rec[i].rid, rec[i].fragment_start, rec[i].fragment_length, rec[i].mapq, rec[i].direction,
                   rec[i].is_unique, rec[i].positive_alignment_length, rec[i].negative_alignment_length) for i in range(k))


def run_case(mapper_factory, cfg, tmp_path):
    seed, preset, kw, gen = cfg
    fa, b1, o1, b2, o2 = fuzz_data.write_case(str(tmp_path), seed, **gen)
    okw = dict(kw)
    if preset == "hic":
        okw.setdefault("error_threshold", 4)
    o = ol.Oracle(None, fa, ol.params(preset, **kw))
    idx = str(tmp_path / "f.idx")
    assert o.L.ora_index_save(idx.encode(), C.byref(o.idx)) == 0
    orec, ok, ost, _ = o.map_pairs(b1, o1, b2, o2)
    gkw = {CAPI_NAMES.get(k, k): v for k, v in kw.items()}
    grec, gk, gst = mapper_factory(idx, fa, preset, gkw, b1, o1, b2, o2)
    hic = preset == "hic"
    assert gk == ok
    assert _tuples_g(grec, gk, hic) == _tuples_o(orec, ok, hic)
    od = ost.as_dict()
    for key in ("num_candidates", "num_mappings", "num_mapped_reads", "num_uniquely_mapped_reads"):
        assert gst[key] == od[key], key
    # the data really is adversarial: multi-mappers and rescue are exercised
    return od, gst


@pytest.mark.parametrize("cfg", fuzz_data.CONFIGS, ids=[str(c[0]) for c in fuzz_data.CONFIGS])
def test_fuzz_stage_functions(cfg, tmp_path):
    def factory(idx, fa, preset, gkw, b1, o1, b2, o2):
        h = he.HostEmu(idx, fa, he.params(preset, **gkw))
        rec, k, st, _ = h.map_pairs(b1, o1, b2, o2)
        return rec, k, st.as_dict()
    od, gst = run_case(factory, cfg, tmp_path)
    assert od["num_mapped_reads"] > 0


@pytest.mark.parametrize("bc_err", [1, 2])
def test_barcode_correction_dense_whitelist(bc_err, tmp_path):
    """short barcodes + nearly complete whitelist: > 132 candidates per uncorrectable barcode at
    threshold 2 (selection path), N handling, quality clamps"""
    import bc_fuzz
    fa, idx, b1, o1, b2, o2, bc, bcq, bco, wl = bc_fuzz.build(str(tmp_path))
    want, n_in, n_corr, keys = bc_fuzz.oracle_result(fa, idx, b1, o1, b2, o2, bc, bcq, bco, wl, bc_err)
    h = he.HostEmu(idx, fa, he.params("atac", mapq_threshold=0, bc_error_threshold=bc_err))
    rec, k, st = h.map_pairs_bc(b1, o1, b2, o2, bc, bcq, bco, keys)
    got = sorted((rec[i].r.read_id, rec[i].r.rid, rec[i].r.fragment_start, rec[i].r.fragment_length, rec[i].r.mapq,
                  rec[i].r.direction, rec[i].barcode) for i in range(k))
    s = st.as_dict()
    assert (s["num_barcode_in_whitelist"], s["num_corrected_barcode"]) == (n_in, n_corr)
    assert n_corr > 10
    assert got == want


def test_alignment_on_bit_planes_equals_byte_form():
    """cm_banded_align_planes (what k_s5b_verify runs) against cm_banded_align on random windows: both cases of the letters,
    bytes outside ACGT on both sides, indels, every window offset modulo 32, reads of 1..150 (one to five plane words) and
    error thresholds 1..15, both strands through the product's packer (cm_pack_read_planes)"""
    import hostemu_lib as hl
    L = hl.lib()
    f = L.hostemu_align_planes_check
    f.restype = C.c_int
    f.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_int]
    assert f(1, 40000, 70, 8) == 0
    assert f(2, 20000, 150, 15) == 0
    assert f(3, 20000, 33, 4) == 0


def test_dropoff_alignment_on_bit_planes_equals_byte_form():
    """cm_banded_align_dropoff_planes (split-alignment verification, --preset hic) against the byte form in the four shapes
    cm_draft_strand_split calls it in, on reads with chimeric halves, indels and bytes outside ACGT"""
    import hostemu_lib as hl
    L = hl.lib()
    f = L.hostemu_dropoff_planes_check
    f.restype = C.c_int
    f.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_int]
    assert f(1, 40000, 160, 8) == 0
    assert f(2, 20000, 70, 4) == 0
    assert f(3, 20000, 300, 15) == 0
