"""Golden cases of the output modes added last (single-end reads with cell barcodes in their three duplicate-removal
flavours, TagAlign text): mapping and text both on the device, byte-identical to the reference's files."""
import hashlib

import pytest

import datasets
import oracle_lib as ol

pytestmark = pytest.mark.gpu


def _gpu(case):
    from chromap_amd import ChromapGPU
    meta = datasets.case_meta(case)
    fa, r1, r2 = datasets.case_inputs(case)
    preset, kw = datasets.flags_to_params(meta["chromap_flags"])
    return ChromapGPU(datasets.case_index(case), fa, preset=preset, **kw), meta, r1, r2


@pytest.mark.parametrize("case", datasets.SE_BC_CASES)
def test_single_end_barcoded_text_equals_reference(case):
    from chromap_amd import _capi
    g, meta, r1, r2 = _gpu(case)
    bcf, wlf = datasets.case_barcode_inputs(case)
    b, off = ol.read_fastx(r1 if datasets.single_end_mate(case) == 1 else r2)
    bc, bcq, bco = ol.read_fastq_qual(bcf)
    g.set_whitelist_file(wlf, int(bco[1] - bco[0]))
    g.compute_barcode_abundance(bc, bco)
    _, k = g.map_single_barcoded(b, off, bc, bcq, bco)
    assert g.store_append_resident() == k
    kind = _capi.TEXT_TAGALIGN_SE_BC if datasets.is_tagalign(case) else _capi.TEXT_BED_SE_BC
    lines, _ = g.store_format(kind, barcode_length=g.barcode_length)
    ref = meta["reference_stderr_counters"]
    assert hashlib.md5(g.store_text()).hexdigest() == meta["bed_md5"]
    assert lines == ref["num_output"]
    s = g.stats.as_dict()
    for key in ("num_candidates", "num_mappings", "num_mapped_reads", "num_uniquely_mapped_reads", "num_barcode_in_whitelist",
                "num_corrected_barcode"):
        assert s[key] == ref[key], key
    g.close()


@pytest.mark.parametrize("case", datasets.TAGALIGN_CASES)
def test_tagalign_text_equals_reference(case):
    from chromap_amd import _capi
    g, meta, r1, r2 = _gpu(case)
    mate = datasets.single_end_mate(case)
    if mate:
        b, off = ol.read_fastx(r1 if mate == 1 else r2)
        rec, k = g.map_single(b, off)
        g.store_append(rec, k)
        lines, _ = g.store_format(_capi.TEXT_BED_SE)  # single-end bulk TagAlign is the BED line (mapping_writer.cc:44-67)
    elif datasets.has_barcodes(case):
        bcf, wlf = datasets.case_barcode_inputs(case)
        b1, o1 = ol.read_fastx(r1)
        b2, o2 = ol.read_fastx(r2)
        bc, bcq, bco = ol.read_fastq_qual(bcf)
        g.set_whitelist_file(wlf, int(bco[1] - bco[0]))
        g.compute_barcode_abundance(bc, bco)
        _, k = g.map_pairs_barcoded(b1, o1, b2, o2, bc, bcq, bco)
        assert g.store_append_resident() == k
        lines, _ = g.store_format(_capi.TEXT_TAGALIGN_PE_BC, barcode_length=g.barcode_length)
    else:
        b1, o1 = ol.read_fastx(r1)
        b2, o2 = ol.read_fastx(r2)
        _, k = g.map_pairs(b1, o1, b2, o2)
        assert g.store_append_resident() == k
        lines, _ = g.store_format(_capi.TEXT_TAGALIGN_PE)
    assert hashlib.md5(g.store_text()).hexdigest() == meta["bed_md5"]
    assert lines in (meta["reference_stderr_counters"]["num_output"], 2 * meta["reference_stderr_counters"]["num_output"])
    g.close()
