"""The product's per-item stage functions (chromap_amd/csrc/cm_stages.h), compiled for the host
and driven by loops in the order of cmgpu_map_resident, must reproduce the reference's BED
and counters.  This checks the device LOGIC without a GPU; the HIP build of the same
functions is checked on the GPU box by tests/test_gpu_parity.py."""
import hashlib

import pytest

import datasets
import hostemu_lib as he
import oracle_lib as ol


@pytest.mark.parametrize("case", [c for c in datasets.BC_CASES if "bulk_level" not in c])
def test_barcode_stage_matches_reference(case, tmp_path):
    """K6 (bc-error-threshold 1) + barcoded records through the stage functions"""
    meta = datasets.case_meta(case)
    fa, r1, r2 = datasets.case_inputs(case)
    idx = datasets.case_index(case)
    bcf, wlf = datasets.case_barcode_inputs(case)
    preset, kw = datasets.flags_to_params(meta["chromap_flags"])
    h = he.HostEmu(idx, fa, he.params(preset, **kw))
    b1, o1 = ol.read_fastx(r1)
    b2, o2 = ol.read_fastx(r2)
    bc, bcq, bco = ol.read_fastq_qual(bcf)
    wl = ol.Whitelist(wlf, int(bco[1] - bco[0]))
    keys, _ = wl.export()
    rec, k, st = h.map_pairs_bc(b1, o1, b2, o2, bc, bcq, bco, keys)
    out = str(tmp_path / "e.bed")
    h.write_bed_bc(rec, k, wl.barcode_length, out)
    assert hashlib.md5(open(out, "rb").read()).hexdigest() == meta["bed_md5"]
    ref = meta["reference_stderr_counters"]
    s = st.as_dict()
    for key in ("num_candidates", "num_mappings", "num_mapped_reads", "num_uniquely_mapped_reads",
                "num_barcode_in_whitelist", "num_corrected_barcode"):
        assert s[key] == ref[key], key


@pytest.mark.parametrize("case", datasets.SE_CASES)
def test_single_end_stage_matches_reference(case, tmp_path):
    meta = datasets.case_meta(case)
    fa, r1, r2 = datasets.case_inputs(case)
    idx = datasets.case_index(case)
    preset, kw = datasets.flags_to_params(meta["chromap_flags"])
    h = he.HostEmu(idx, fa, he.params(preset, **kw))
    b, off = ol.read_fastx(r1 if datasets.single_end_mate(case) == 1 else r2)
    rec, k, st = h.map_single(b, off)
    out = str(tmp_path / "e.bed")
    h.write_bed_se(rec, k, out)
    assert hashlib.md5(open(out, "rb").read()).hexdigest() == meta["bed_md5"]
    ref = meta["reference_stderr_counters"]
    s = st.as_dict()
    for key in ("num_candidates", "num_mappings", "num_mapped_reads", "num_uniquely_mapped_reads"):
        assert s[key] == ref[key], key


@pytest.mark.parametrize("case", datasets.BED_CASES + datasets.HIC_CASES)
def test_stage_functions_match_reference(case, tmp_path):
    meta = datasets.case_meta(case)
    fa, r1, r2 = datasets.case_inputs(case)
    idx = datasets.case_index(case)
    preset, kw = datasets.flags_to_params(meta["chromap_flags"])
    p = he.params(preset, **kw)
    h = he.HostEmu(idx, fa, p)
    b1, o1 = ol.read_fastx(r1)
    b2, o2 = ol.read_fastx(r2)
    rec, k, st, dbg = h.map_pairs(b1, o1, b2, o2)
    out = str(tmp_path / "e.bed")
    if datasets.is_hic(case):
        h.write_pairs(rec, k, ol.read_names(r1), out)
    else:
        h.write_bed(rec, k, out)
    got = open(out, "rb").read()
    assert hashlib.md5(got).hexdigest() == meta["bed_md5"]
    ref = meta["reference_stderr_counters"]
    s = st.as_dict()
    for key in ("num_candidates", "num_mappings", "num_mapped_reads", "num_uniquely_mapped_reads"):
        assert s[key] == ref[key], key
    # stage-level agreement with the oracle's per-pair trace
    o = ol.Oracle(idx, fa, ol.params(preset, **kw))
    _, _, _, tr = o.map_pairs(b1, o1, b2, o2, trace=True)
    n = len(o1) - 1
    for i in range(n):
        t = tr[i]
        if t.len1 == 0 and t.len2 == 0:
            continue  # dropped by the length filter before anything was traced
        assert (dbg["mm_cnt"][2 * i], dbg["mm_cnt"][2 * i + 1]) == (t.n_mm1, t.n_mm2), i
        if t.n_cand1 > 0 and t.n_cand2 > 0:
            assert (dbg["ncand"][2 * i], dbg["ncand"][2 * i + 1]) == (t.n_cand1, t.n_cand2), i
            assert (dbg["ndraft"][2 * i], dbg["ndraft"][2 * i + 1]) == (t.n_draft1, t.n_draft2), i
            if t.n_draft1 > 0 and t.n_draft2 > 0:
                assert dbg["nbest"][i] == t.nbest, i
    o.close()


@pytest.mark.parametrize("case", datasets.SAM_CASES + datasets.HIC_SAM_CASES)
def test_sam_stage_functions_match_reference(case, tmp_path):
    """--SAM through the stage functions (register-window ksw, NM/MD, SAM records) and the host SAM writer"""
    meta = datasets.case_meta(case)
    fa, r1, r2 = datasets.case_inputs(case)
    preset, kw = datasets.flags_to_params(meta["chromap_flags"])
    h = he.HostEmu(datasets.case_index(case), fa, he.params(preset, **kw))
    mate = datasets.single_end_mate(case)
    out = str(tmp_path / "e.sam")
    if mate:
        f = r1 if mate == 1 else r2
        b, q, off = ol.read_fastq_qual(f)
        so, st = he.map_sam(h, b, off)
        lines = he.write_sam(h.L, h.ref, h.p, so, False, ol.read_names(f), None, b, q, off, None, None, None, out)
    else:
        b1, q1, o1 = ol.read_fastq_qual(r1)
        b2, q2, o2 = ol.read_fastq_qual(r2)
        so, st = he.map_sam(h, b1, o1, b2, o2)
        lines = he.write_sam(h.L, h.ref, h.p, so, True, ol.read_names(r1), ol.read_names(r2), b1, q1, o1, b2, q2, o2, out)
    got = open(out, "rb").read()
    want = datasets.case_golden_bed(case)
    if got != want:
        g, w = got.split(b"\n"), want.split(b"\n")
        for i in range(min(len(g), len(w))):
            assert g[i] == w[i], (i, g[i], w[i])
    assert hashlib.md5(got).hexdigest() == meta["bed_md5"]
    assert lines == meta["reference_stderr_counters"]["num_output"]


@pytest.mark.parametrize("case", datasets.SAM_BC_CASES)
def test_sam_with_barcodes_matches_reference(case, tmp_path):
    """single-cell --SAM: barcode correction + SAM records through the stage functions; the host writer sorts on the
    barcode and prints CB:Z"""
    meta = datasets.case_meta(case)
    fa, r1, r2 = datasets.case_inputs(case)
    bcf, wlf = datasets.case_barcode_inputs(case)
    preset, kw = datasets.flags_to_params(meta["chromap_flags"])
    h = he.HostEmu(datasets.case_index(case), fa, he.params(preset, **kw))
    b1, q1, o1 = ol.read_fastq_qual(r1)
    b2, q2, o2 = ol.read_fastq_qual(r2)
    bc, bcq, bco = ol.read_fastq_qual(bcf)
    wl = ol.Whitelist(wlf, int(bco[1] - bco[0]))
    keys, _ = wl.export()
    so, bck, st = he.map_bc_sam(h, b1, o1, b2, o2, bc, bcq, bco, keys)
    out = str(tmp_path / "e.sam")
    lines = he.write_sam_bc(h.L, h.ref, h.p, so, ol.read_names(r1), ol.read_names(r2), b1, q1, o1, b2, q2, o2, bck, wl.barcode_length, out)
    got = open(out, "rb").read()
    want = datasets.case_golden_bed(case)
    if got != want:
        g, w = got.split(b"\n"), want.split(b"\n")
        for i in range(min(len(g), len(w))):
            assert g[i] == w[i], (i, g[i], w[i])
    assert hashlib.md5(got).hexdigest() == meta["bed_md5"]
    ref = meta["reference_stderr_counters"]
    assert lines == ref["num_output"]
    s = st.as_dict()
    for key in ("num_candidates", "num_mappings", "num_mapped_reads", "num_uniquely_mapped_reads", "num_barcode_in_whitelist",
                "num_corrected_barcode"):
        assert s[key] == ref[key], key


@pytest.mark.parametrize("case", datasets.SE_BC_CASES)
def test_single_end_barcode_stage_matches_reference(case, tmp_path):
    """single-end reads with cell barcodes through the stage functions: records equal the oracle's, and the
    oracle's writer (the reference's merge loop, pinned by the golden files) renders them to the golden text"""
    meta = datasets.case_meta(case)
    fa, r1, r2 = datasets.case_inputs(case)
    idx = datasets.case_index(case)
    bcf, wlf = datasets.case_barcode_inputs(case)
    preset, kw = datasets.flags_to_params(meta["chromap_flags"])
    h = he.HostEmu(idx, fa, he.params(preset, **kw))
    b, off = ol.read_fastx(r1 if datasets.single_end_mate(case) == 1 else r2)
    bc, bcq, bco = ol.read_fastq_qual(bcf)
    wl = ol.Whitelist(wlf, int(bco[1] - bco[0]))
    assert wl.abundance(bc, bco) > 0
    keys, _ = wl.export()
    rec, k, st = h.map_single_bc(b, off, bc.copy(), bcq, bco, keys)
    o = ol.Oracle(idx, fa, ol.params(preset, **kw))
    orec, ok_, _, _, _ = ol.map_single_bc(o, b, off, bc.copy(), bcq, bco, wl)

    def tup(r):
        return (r.r.read_id, r.r.rid, r.r.fragment_start, r.r.fragment_length, r.r.mapq, r.r.direction, r.r.is_unique, r.barcode)
    assert k == ok_
    assert sorted(tup(rec[i]) for i in range(k)) == sorted(tup(orec[i]) for i in range(ok_))
    out = str(tmp_path / "e.bed")
    lines = ol.write_se_bc(o, rec, k, wl.barcode_length, wl, datasets.is_tagalign(case), out)
    assert hashlib.md5(open(out, "rb").read()).hexdigest() == meta["bed_md5"]
    ref = meta["reference_stderr_counters"]
    assert lines == ref["num_output"]
    s = st.as_dict()
    for key in ("num_candidates", "num_mappings", "num_mapped_reads", "num_uniquely_mapped_reads",
                "num_barcode_in_whitelist", "num_corrected_barcode"):
        assert s[key] == ref[key], key
    o.close()


@pytest.mark.parametrize("case", datasets.TAGALIGN_CASES)
def test_tagalign_records_render_to_reference_text(case, tmp_path):
    """--TagAlign: records from the stage functions, text by the oracle's writer (pinned by the same golden files)"""
    meta = datasets.case_meta(case)
    fa, r1, r2 = datasets.case_inputs(case)
    idx = datasets.case_index(case)
    preset, kw = datasets.flags_to_params(meta["chromap_flags"])
    h = he.HostEmu(idx, fa, he.params(preset, **kw))
    o = ol.Oracle(idx, fa, ol.params(preset, **kw))
    o.p.output_format = 2
    out = str(tmp_path / "e.tagalign")
    mate = datasets.single_end_mate(case)
    if mate:
        b, off = ol.read_fastx(r1 if mate == 1 else r2)
        rec, k, st = h.map_single(b, off)
        lines = ol.write_bed_se(o, rec, k, out)
    elif datasets.has_barcodes(case):
        bcf, wlf = datasets.case_barcode_inputs(case)
        b1, o1 = ol.read_fastx(r1)
        b2, o2 = ol.read_fastx(r2)
        bc, bcq, bco = ol.read_fastq_qual(bcf)
        wl = ol.Whitelist(wlf, int(bco[1] - bco[0]))
        keys, _ = wl.export()
        rec, k, st = h.map_pairs_bc(b1, o1, b2, o2, bc, bcq, bco, keys)
        lines = ol.write_bed_bc(o, rec, k, wl.barcode_length, out)
    else:
        b1, o1 = ol.read_fastx(r1)
        b2, o2 = ol.read_fastx(r2)
        rec, k, st, _ = h.map_pairs(b1, o1, b2, o2)
        lines = o.write_bed(rec, k, out)
    assert hashlib.md5(open(out, "rb").read()).hexdigest() == meta["bed_md5"]
    ref = meta["reference_stderr_counters"]
    assert lines == ref["num_output"]
    s = st.as_dict()
    for key in ("num_candidates", "num_mappings", "num_mapped_reads", "num_uniquely_mapped_reads"):
        assert s[key] == ref[key], key
    o.close()
