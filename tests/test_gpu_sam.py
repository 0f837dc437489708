"""--SAM on the device: banded affine-gap alignment of every reported read
(register-window ksw_semi_global3 with the move bits in HBM), NM / MD, SAM records; text by the
host writer.  Byte-identical to the reference's SAM files, record-identical to the oracle."""
import hashlib

import pytest

import datasets
import oracle_lib as ol

pytestmark = pytest.mark.gpu


def _tuples(sam):
    from chromap_amd import _capi
    rec, cigar, md, md_cap, n = sam
    out = []
    for i in range(n):
        r = rec[i]
        if not r.valid:
            continue
        cg = tuple(int(x) for x in cigar[i * _capi.SAM_CIGAR_CAP:i * _capi.SAM_CIGAR_CAP + r.n_cigar])
        out.append((i, r.read_id, r.rid, r.pos, r.mpos, r.mrid, r.tlen, r.nm, r.flag, r.mapq, r.strand, r.is_unique, cg,
                    md[i * md_cap:i * md_cap + r.md_len].tobytes(), r.length_after_trim))
    return out


@pytest.mark.parametrize("case", datasets.SAM_CASES + datasets.HIC_SAM_CASES)
def test_sam_matches_reference_and_oracle(case, tmp_path):
    from chromap_amd import ChromapGPU
    meta = datasets.case_meta(case)
    fa, r1, r2 = datasets.case_inputs(case)
    idx = datasets.case_index(case)
    preset, kw = datasets.flags_to_params(meta["chromap_flags"])
    g = ChromapGPU(idx, fa, preset=preset, **kw)
    o = ol.Oracle(idx, fa, ol.params(preset, **kw))
    mate = datasets.single_end_mate(case)
    out = str(tmp_path / "g.sam")
    if mate:
        f = r1 if mate == 1 else r2
        b, q, off = ol.read_fastq_qual(f)
        g.map_single(b, off)
        sam = g.download_sam()
        lines = g.write_sam(sam, False, ol.read_names(f), None, b, q, off, None, None, None, out)
        ores, _, _ = ol.map_single_sam(o, b, off)
    else:
        b1, q1, o1 = ol.read_fastq_qual(r1)
        b2, q2, o2 = ol.read_fastq_qual(r2)
        g.map_pairs(b1, o1, b2, o2)
        sam = g.download_sam()
        lines = g.write_sam(sam, True, ol.read_names(r1), ol.read_names(r2), b1, q1, o1, b2, q2, o2, out)
        ores, _, _ = ol.map_pairs_sam(o, b1, o1, b2, o2)
    assert _tuples(sam) == ores.tuples()
    got = open(out, "rb").read()
    assert hashlib.md5(got).hexdigest() == meta["bed_md5"]
    assert got == datasets.case_golden_bed(case)
    ref = meta["reference_stderr_counters"]
    assert lines == ref["num_output"]
    s = g.stats.as_dict()
    for key in ("num_candidates", "num_mappings", "num_mapped_reads", "num_uniquely_mapped_reads"):
        assert s[key] == ref[key], key
    g.close()
    o.close()


@pytest.mark.parametrize("case", datasets.SAM_BC_CASES)
def test_sam_with_barcodes_matches_reference(case, tmp_path):
    """single-cell --SAM: barcode correction on the device, keys downloaded for the writer's sort and the CB:Z tag"""
    from chromap_amd import ChromapGPU
    meta = datasets.case_meta(case)
    fa, r1, r2 = datasets.case_inputs(case)
    bcf, wlf = datasets.case_barcode_inputs(case)
    preset, kw = datasets.flags_to_params(meta["chromap_flags"])
    g = ChromapGPU(datasets.case_index(case), fa, preset=preset, **kw)
    b1, q1, o1 = ol.read_fastq_qual(r1)
    b2, q2, o2 = ol.read_fastq_qual(r2)
    bc, bcq, bco = ol.read_fastq_qual(bcf)
    g.set_whitelist_file(wlf, int(bco[1] - bco[0]))
    g.compute_barcode_abundance(bc, bco)
    g.map_pairs_barcoded(b1, o1, b2, o2, bc, bcq, bco)
    sam = g.download_sam()
    keys = g.download_barcode_keys(len(o1) - 1)
    out = str(tmp_path / "g.sam")
    lines = g.write_sam(sam, True, ol.read_names(r1), ol.read_names(r2), b1, q1, o1, b2, q2, o2, out, barcode_keys=keys,
                        barcode_length=int(bco[1] - bco[0]))
    got = open(out, "rb").read()
    assert hashlib.md5(got).hexdigest() == meta["bed_md5"]
    ref = meta["reference_stderr_counters"]
    assert lines == ref["num_output"]
    s = g.stats.as_dict()
    for key in ("num_candidates", "num_mappings", "num_mapped_reads", "num_uniquely_mapped_reads", "num_barcode_in_whitelist",
                "num_corrected_barcode"):
        assert s[key] == ref[key], key
    g.close()
