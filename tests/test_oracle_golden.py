"""Pins the oracle (oracle/chromap_oracle.c) against outputs of the reference itself.

The golden BED files and stderr counters under tests/golden/ were produced by
oracle/_ref/chromap (the unchanged reference, v0.3.3-r521) via tests/golden/make_golden.py.
"""
import ctypes as C
import hashlib
import os

import numpy as np
import pytest

import datasets
import oracle_lib as ol


@pytest.mark.parametrize("case", [c for c in datasets.ALL_CASES if not datasets.is_sam(c)])
def test_oracle_bed_matches_reference(case, tmp_path):
    meta = datasets.case_meta(case)
    fa, r1, r2 = datasets.case_inputs(case)
    preset, kw = datasets.flags_to_params(meta["chromap_flags"])
    p = ol.params(preset, **kw)
    o = ol.Oracle(None, fa, p)  # index built by the oracle's own indexer
    if datasets.is_tagalign(case):
        o.p.output_format = 2  # the writers' text format only; mapping ran with the context's copy
    b1, o1 = ol.read_fastx(r1)
    b2, o2 = ol.read_fastx(r2)
    if datasets.single_end_mate(case) and datasets.has_barcodes(case):
        _check_single_end_barcode_case(case, meta, o, (b1, o1) if datasets.single_end_mate(case) == 1 else (b2, o2), tmp_path)
        o.close()
        return
    if datasets.single_end_mate(case):
        b, off = (b1, o1) if datasets.single_end_mate(case) == 1 else (b2, o2)
        rec, k, st = ol.map_single(o, b, off, threads=2)
        out = str(tmp_path / "o.bed")
        ol.write_bed_se(o, rec, k, out)
        assert hashlib.md5(open(out, "rb").read()).hexdigest() == meta["bed_md5"]
        ref = meta["reference_stderr_counters"]
        s = st.as_dict()
        for key in ("num_candidates", "num_mappings", "num_mapped_reads", "num_uniquely_mapped_reads"):
            assert s[key] == ref[key], key
        o.close()
        return
    if datasets.has_barcodes(case):
        _check_barcode_case(case, meta, o, b1, o1, b2, o2, tmp_path)
        o.close()
        return
    rec, k, st, _ = o.map_pairs(b1, o1, b2, o2)
    out = str(tmp_path / "o.bed")
    hic = datasets.is_hic(case)
    names = ol.read_names(r1) if hic else None
    if hic:
        ol.write_pairs(o, rec, k, names, out)
    else:
        o.write_bed(rec, k, out)
    got = open(out, "rb").read()
    assert hashlib.md5(got).hexdigest() == meta["bed_md5"]
    assert got == datasets.case_golden_bed(case)
    ref = meta["reference_stderr_counters"]
    s = st.as_dict()
    for key in ("num_candidates", "num_mappings", "num_mapped_reads", "num_uniquely_mapped_reads"):
        assert s[key] == ref[key], key
    # threaded run gives the same records (RNG scope is per taskloop task, not per thread)
    rec2, k2, _, _ = o.map_pairs(b1, o1, b2, o2, threads=4)
    assert k2 == k
    assert bytes(rec2)[: k * C.sizeof(ol.OraRecord)] != b"" or k == 0
    out2 = str(tmp_path / "o2.bed")
    if hic:
        ol.write_pairs(o, rec2, k2, names, out2)
    else:
        o.write_bed(rec2, k2, out2)
    assert open(out2, "rb").read() == got
    o.close()


def _check_barcode_case(case, meta, o, b1, o1, b2, o2, tmp_path):
    bcf, wlf = datasets.case_barcode_inputs(case)
    bc, bcq, bco = ol.read_fastq_qual(bcf)
    wl = ol.Whitelist(wlf, int(bco[1] - bco[0]))
    assert wl.abundance(bc, bco) > 0
    rec, k, st, n_in, n_corr = ol.map_pairs_bc(o, b1, o1, b2, o2, bc, bcq, bco, wl, threads=2)
    out = str(tmp_path / "o.bed")
    if o.p.dedup_at_bulk_level and o.p.low_mem and o.p.remove_pcr_duplicates:
        ol.write_bed_bc_bulk(o, rec, k, wl.barcode_length, wl, out)
    else:
        ol.write_bed_bc(o, rec, k, wl.barcode_length, out)
    got = open(out, "rb").read()
    assert hashlib.md5(got).hexdigest() == meta["bed_md5"]
    ref = meta["reference_stderr_counters"]
    s = st.as_dict()
    for key in ("num_candidates", "num_mappings", "num_mapped_reads", "num_uniquely_mapped_reads"):
        assert s[key] == ref[key], key
    assert n_in == ref["num_barcode_in_whitelist"] and n_corr == ref["num_corrected_barcode"]


def _check_single_end_barcode_case(case, meta, o, read, tmp_path):
    """MappingWithBarcode: cell-level / bulk-level low-memory merge, in-memory duplicate removal, TagAlign"""
    b, off = read
    bcf, wlf = datasets.case_barcode_inputs(case)
    bc, bcq, bco = ol.read_fastq_qual(bcf)
    wl = ol.Whitelist(wlf, int(bco[1] - bco[0]))
    assert wl.abundance(bc, bco) > 0
    rec, k, st, n_in, n_corr = ol.map_single_bc(o, b, off, bc, bcq, bco, wl, threads=2)
    out = str(tmp_path / "o.bed")
    lines = ol.write_se_bc(o, rec, k, wl.barcode_length, wl, datasets.is_tagalign(case), out)
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
    for key in ("num_candidates", "num_mappings", "num_mapped_reads", "num_uniquely_mapped_reads"):
        assert s[key] == ref[key], key
    assert n_in == ref["num_barcode_in_whitelist"] and n_corr == ref["num_corrected_barcode"]


def test_oracle_index_matches_reference_on_defined_bytes():
    """Index built by the oracle == survey-recorded reference index for test/ref.fa on every
    byte the reference defines (flags, occupied buckets, occurrence table, header). Empty
    buckets hold stale heap bytes in the reference's file."""
    L = ol.lib()
    fa = os.path.join(datasets.GOLD, "toy", "ref.fa")
    ref = ol.OraRef()
    assert L.ora_ref_load(fa.encode(), C.byref(ref)) == 0
    idx = ol.OraIndex()
    assert L.ora_index_build(C.byref(ref), 17, 7, C.byref(idx)) == 0
    assert (idx.k, idx.w) == (17, 7)
    # 25079 distinct minimizers, all singletons, file size 532512 B
    assert idx.size == 25079 and idx.n_occ == 0
    assert 28 + max(1, idx.n_buckets // 16) * 4 + idx.n_buckets * 16 + 4 == 532512
    L.ora_index_free(C.byref(idx))
    L.ora_ref_free(C.byref(ref))


def test_hash64_known_values():
    L = ol.lib()
    mask = (1 << 34) - 1
    # invertible mix: distinct inputs -> distinct outputs, stays within mask
    xs = [0, 1, 2, 12345678901, mask]
    hs = [L.ora_hash64(x, mask) for x in xs]
    assert len(set(hs)) == len(xs) and all(h <= mask for h in hs)


@pytest.mark.parametrize("case", datasets.SAM_CASES + datasets.HIC_SAM_CASES)
def test_sam_matches_reference(case, tmp_path):
    """--SAM: ksw_semi_global3 CIGARs, NM / MD tags, SAMMapping order and duplicate removal"""
    meta = datasets.case_meta(case)
    fa, r1, r2 = datasets.case_inputs(case)
    idx = datasets.case_index(case)
    preset, kw = datasets.flags_to_params(meta["chromap_flags"])
    o = ol.Oracle(idx, fa, ol.params(preset, **kw))
    mate = datasets.single_end_mate(case)
    out = str(tmp_path / "o.sam")
    if mate:
        f = r1 if mate == 1 else r2
        b, q, off = ol.read_fastq_qual(f)
        res, k, st = ol.map_single_sam(o, b, off)
        lines = ol.write_sam(o, res, False, ol.read_names(f), None, b, q, off, None, None, None, out)
    else:
        b1, q1, o1 = ol.read_fastq_qual(r1)
        b2, q2, o2 = ol.read_fastq_qual(r2)
        res, k, st = ol.map_pairs_sam(o, b1, o1, b2, o2)
        lines = ol.write_sam(o, res, True, ol.read_names(r1), ol.read_names(r2), b1, q1, o1, b2, q2, o2, out)
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
    for key in ("num_candidates", "num_mappings", "num_mapped_reads", "num_uniquely_mapped_reads"):
        assert s[key] == ref[key], key
    o.close()


@pytest.mark.parametrize("case", datasets.SAM_BC_CASES)
def test_sam_with_barcodes_matches_reference(case, tmp_path):
    """single-cell --SAM: cell_barcode_ in SAMMapping's order / equality, CB:Z tag"""
    meta = datasets.case_meta(case)
    fa, r1, r2 = datasets.case_inputs(case)
    bcf, wlf = datasets.case_barcode_inputs(case)
    preset, kw = datasets.flags_to_params(meta["chromap_flags"])
    o = ol.Oracle(datasets.case_index(case), fa, ol.params(preset, **kw))
    b1, q1, o1 = ol.read_fastq_qual(r1)
    b2, q2, o2 = ol.read_fastq_qual(r2)
    bc, bcq, bco = ol.read_fastq_qual(bcf)
    wl = ol.Whitelist(wlf, int(bco[1] - bco[0]))
    assert wl.abundance(bc, bco) > 0
    res, keys, st = ol.map_pairs_bc_sam(o, b1, o1, b2, o2, bc, bcq, bco, wl)
    out = str(tmp_path / "o.sam")
    lines = ol.write_sam_bc(o, res, True, ol.read_names(r1), ol.read_names(r2), b1, q1, o1, b2, q2, o2, keys, wl.barcode_length, out)
    got = open(out, "rb").read()
    assert hashlib.md5(got).hexdigest() == meta["bed_md5"]
    assert lines == meta["reference_stderr_counters"]["num_output"]
    o.close()
