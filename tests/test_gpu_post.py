"""Device-side post-processing (cm_post.hip): records stay in HBM, are radix
sorted on the record's operator< key, de-duplicated, filtered, Tn5-shifted and rendered as BED
text on the device.  The text must equal the reference's output file byte for byte (golden
cases, both the low-memory and the in-memory flavour of duplicate removal), and the host
writers on adversarial random record sets (long duplicate runs, > 65536 sequences)."""
import ctypes as C
import hashlib

import numpy as np
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


@pytest.mark.parametrize("case", datasets.BED_CASES)
def test_device_bed_pe_equals_reference(case):
    from chromap_amd import _capi
    g, meta, r1, r2 = _gpu(case)
    b1, o1 = ol.read_fastx(r1)
    b2, o2 = ol.read_fastx(r2)
    _, k = g.map_pairs(b1, o1, b2, o2)
    assert g.store_append_resident() == k
    lines, nbytes = g.store_format(_capi.TEXT_BED_PE)
    text = g.store_text()
    assert len(text) == nbytes and text.count(b"\n") == lines
    assert lines == meta["reference_stderr_counters"]["num_output"]
    assert hashlib.md5(text).hexdigest() == meta["bed_md5"]
    assert text == datasets.case_golden_bed(case)
    g.close()


@pytest.mark.parametrize("case", datasets.SE_CASES)
def test_device_bed_se_equals_reference(case):
    from chromap_amd import _capi
    g, meta, r1, r2 = _gpu(case)
    b, off = ol.read_fastx(r1 if datasets.single_end_mate(case) == 1 else r2)
    rec, k = g.map_single(b, off)
    # host records appended in two pieces: the store grows and keeps earlier content
    g.store_append(rec, k // 3)
    g.store_append(C.addressof(rec) + (k // 3) * 24, k - k // 3)
    lines, _ = g.store_format(_capi.TEXT_BED_SE)
    assert hashlib.md5(g.store_text()).hexdigest() == meta["bed_md5"]
    assert lines == meta["reference_stderr_counters"]["num_output"]
    g.close()


@pytest.mark.parametrize("case", datasets.BC_CASES)
def test_device_bed_barcoded_equals_reference(case, tmp_path):
    from chromap_amd import _capi
    g, meta, r1, r2 = _gpu(case)
    bcf, wlf = datasets.case_barcode_inputs(case)
    b1, o1 = ol.read_fastx(r1)
    b2, o2 = ol.read_fastx(r2)
    bc, bcq, bco = ol.read_fastq_qual(bcf)
    g.set_whitelist_file(wlf, int(bco[1] - bco[0]))
    g.compute_barcode_abundance(bc, bco)
    rec, k = g.map_pairs_barcoded(b1, o1, b2, o2, bc, bcq, bco)
    assert g.store_append_resident() == k
    lines, _ = g.store_format(_capi.TEXT_BED_PE_BC, barcode_length=g.barcode_length)
    out = str(tmp_path / "d.bed")
    g.store_write_text(out)
    assert datasets.md5(out) == meta["bed_md5"]
    assert lines == meta["reference_stderr_counters"]["num_output"]
    if g.params.dedup_at_bulk_level:
        g.close()
        return  # bulk-level duplicate removal uses the whitelist abundances on the device: no host twin
    # same through the host-array entry (cmgpu_record_bc, 32 bytes)
    g.store_clear()
    g.store_append(rec, k, barcoded=True)
    g.store_format(_capi.TEXT_BED_PE_BC, barcode_length=g.barcode_length)
    assert hashlib.md5(g.store_text()).hexdigest() == meta["bed_md5"]
    g.close()


def _random_records(rng, n, n_seq, span, bc=False):
    from chromap_amd.distributed import REC_DTYPE
    r = np.zeros(n, REC_DTYPE)
    r["read_id"] = rng.permutation(n).astype(np.uint32)
    r["rid"] = rng.integers(0, n_seq, n)
    r["fragment_start"] = rng.integers(10, 10 + span, n)
    r["fragment_length"] = rng.integers(30, 34, n)
    r["mapq"] = rng.choice([0, 1, 3, 30, 60], n)
    r["direction"] = rng.integers(0, 2, n)
    r["is_unique"] = rng.integers(0, 2, n)
    r["num_dups"] = 1
    r["positive_alignment_length"] = rng.integers(20, 60, n)
    r["negative_alignment_length"] = rng.integers(20, 60, n)
    if not bc:
        return r
    rb = np.zeros(n, np.dtype([("r", REC_DTYPE), ("barcode", "<u8")]))
    rb["r"] = r
    rb["barcode"] = rng.integers(0, 5, n).astype(np.uint64) * np.uint64(0x1234567) % np.uint64(1 << 32)
    return rb


@pytest.mark.parametrize("kind,n_seq,span,dedup,lowmem,tn5,q", [
    (0, 3, 40, 1, 1, 1, 30), (0, 3, 40, 1, 0, 1, 0), (0, 70000, 3, 1, 1, 0, 1), (0, 5, 100000, 0, 1, 0, 0),
    (0, 1, 1, 1, 1, 0, 0), (0, 1, 1, 1, 0, 1, 0),  # pile-ups: runs of ~15 000 duplicates (k_pp_select_long)
    (1, 1, 1, 1, 1, 0, 0), (1, 1, 1, 1, 0, 0, 30), (2, 1, 1, 1, 1, 1, 0), (2, 1, 1, 1, 0, 0, 0), (0, 2, 3, 1, 1, 1, 3), (2, 2, 2, 1, 0, 1, 1),
    (1, 4, 60, 1, 1, 1, 0), (1, 4, 60, 1, 0, 1, 30), (1, 4, 60, 0, 0, 1, 0),
    (2, 4, 30, 1, 1, 1, 0), (2, 4, 30, 1, 0, 1, 30), (2, 4, 30, 0, 1, 0, 0)])
def test_device_text_equals_host_writer_on_random_records(kind, n_seq, span, dedup, lowmem, tn5, q, tmp_path):
    from chromap_amd import ChromapGPU, _capi
    fa, _, _ = datasets.case_inputs("toy_chip")
    g = ChromapGPU(datasets.case_index("toy_chip"), fa, preset="chip")
    g.names = [b"seq%d" % i for i in range(n_seq)]
    p = _capi.default_params(None, remove_pcr_duplicates=dedup, low_memory_mode=lowmem, tn5_shift=tn5, mapq_threshold=q)
    rng = np.random.default_rng(1000 * kind + n_seq + span)
    n = 60000
    rec = _random_records(rng, n, n_seq, span, bc=(kind == 2))
    host = rec.copy()
    out = str(tmp_path / "h.bed")
    if kind == 0:
        g.write_bed(host.ctypes.data, n, out, params=p)
    elif kind == 1:
        g.write_bed_se(host.ctypes.data, n, out, params=p)
    else:
        g.barcode_length = 16
        g.write_bed_bc(host.ctypes.data, n, out, params=p)
    want = open(out, "rb").read()
    g.store_clear()
    half = n // 2
    g.store_append(rec.ctypes.data, half, barcoded=(kind == 2))
    g.store_append(rec.ctypes.data + half * rec.dtype.itemsize, n - half, barcoded=(kind == 2))
    lines, nbytes = g.store_format(kind, params=p, barcode_length=16 if kind == 2 else 0)
    got = g.store_text()
    assert nbytes == len(want) and lines == want.count(b"\n")
    assert got == want
    # formatting again gives the same text (the store is not consumed)
    g.store_format(kind, params=p, barcode_length=16 if kind == 2 else 0)
    assert g.store_text() == want
    g.close()


@pytest.mark.parametrize("span,q", [(1, 0), (1, 30), (3, 1), (400, 30)])
def test_bulk_level_dedup_of_single_cell_records_long_runs(span, q, tmp_path):
    """duplicate removal at bulk level for single-cell data (mapping_writer.h:126-163, 202-345) on random records with pile-ups of
    thousands of duplicates (span 1: four runs of ~15 000 records, taken by a wave each, k_pp_select_long) and with short runs
    (one thread each), against the rule written out in numpy: runs of equal (rid, start, length), barcode groups inside a run
    weigh 2 (two records or more) or 1, the survivor is the group of largest (weight, barcode abundance), the first on ties,
    represented by its last record; MAPQ filter on that record, on the run's maximum for the very last run"""
    from chromap_amd import ChromapGPU, _capi
    fa, _, _ = datasets.case_inputs("toy_chip")
    g = ChromapGPU(datasets.case_index("toy_chip"), fa, preset="chip")
    n_seq = 2
    g.names = [b"seq%d" % i for i in range(n_seq)]
    p = _capi.default_params(None, remove_pcr_duplicates=1, low_memory_mode=1, tn5_shift=0, mapq_threshold=q, dedup_at_bulk_level=1)
    rng = np.random.default_rng(77 + span + q)
    n = 60000
    rb = _random_records(rng, n, n_seq, span, bc=True)
    # seven barcodes, 16 bases each; their abundances come from a sample of barcode reads (two of them equally frequent: the tie)
    bl = 16
    keys = np.array([rng.integers(0, 1 << 32) for _ in range(7)], np.uint64)
    rb["barcode"] = keys[rng.integers(0, 7, n)]
    abundance = {int(keys[i]): a for i, a in enumerate([5, 9, 9, 2, 30, 1, 14])}
    seq = lambda k: bytes(b"ACGT"[(int(k) >> (2 * (bl - 1 - j))) & 3] for j in range(bl))
    sample = b"".join(seq(k) * abundance[int(k)] for k in keys)
    bco = np.arange(0, len(sample) + 1, bl, dtype=np.uint32)
    kk = np.ascontiguousarray(keys)
    assert g.L.cmgpu_set_whitelist(g.ctx, kk.ctypes.data, len(kk), bl) == 0
    g.barcode_length = bl
    g.compute_barcode_abundance(np.frombuffer(sample, np.uint8), bco)
    g.store_clear()
    g.store_append(rb.ctypes.data, n, barcoded=True)
    lines, nbytes = g.store_format(_capi.TEXT_BED_PE_BC, params=p, barcode_length=bl)
    got = g.store_text()
    g.close()
    # the rule in numpy
    r = rb["r"]
    order = np.lexsort((r["read_id"], r["is_unique"], r["direction"], r["mapq"], rb["barcode"], r["fragment_length"], r["fragment_start"], r["rid"]))
    R, B = r[order], rb["barcode"][order]
    runkey = np.stack([R["rid"].astype(np.int64), R["fragment_start"].astype(np.int64), R["fragment_length"].astype(np.int64)], 1)
    starts = np.flatnonzero(np.r_[True, (np.diff(runkey, axis=0) != 0).any(1)])
    ends = np.r_[starts[1:], n]
    want = []
    for a, e in zip(starts, ends):
        gb = B[a:e]
        gs = np.flatnonzero(np.r_[True, gb[1:] != gb[:-1]]) + a
        ge = np.r_[gs[1:], e]
        best = None
        for x, y in zip(gs, ge):
            cand = (2 if y - x >= 2 else 1, abundance[int(B[x])])
            if best is None or cand > best[0]:
                best = (cand, y - 1)
        rep = R[best[1]]
        fm = int(R["mapq"][a:e].max()) if e == n else int(rep["mapq"])
        if fm < q:
            continue
        want.append(b"seq%d\t%d\t%d\t%s\t%d\n" % (rep["rid"], rep["fragment_start"], rep["fragment_start"] + rep["fragment_length"],
                                                   seq(B[best[1]]), min(255, e - a)))
    want = b"".join(want)
    assert lines == want.count(b"\n")
    assert got == want


def test_empty_store_formats_to_nothing():
    from chromap_amd import ChromapGPU, _capi
    fa, _, _ = datasets.case_inputs("toy_chip")
    g = ChromapGPU(datasets.case_index("toy_chip"), fa, preset="chip")
    assert g.store_format(_capi.TEXT_BED_PE) == (0, 0)
    assert g.store_text() == b""
    g.close()


def test_records_partition_by_owner_matches_host_twin():
    """send side of the multi-GPU exchange: records grouped by chromosome owner on the device"""
    import torch
    from chromap_amd import ChromapGPU
    from chromap_amd.distributed import REC_DTYPE, partition_by_owner
    case = "s3_chip"  # 5 chromosomes
    fa, r1, r2 = datasets.case_inputs(case)
    g = ChromapGPU(datasets.case_index(case), fa, preset="chip")
    b1, o1 = ol.read_fastx(r1)
    b2, o2 = ol.read_fastx(r2)
    rec, k = g.map_pairs(b1, o1, b2, o2)
    host = np.frombuffer(bytes(rec)[:k * 24], dtype=REC_DTYPE)
    n = len(o1) - 1
    for world in (1, 2, 3, 8):
        send = torch.zeros(n * 24, dtype=torch.uint8, device="cuda")
        counts = (C.c_uint64 * world)()
        assert g.L.cmgpu_records_partition(g.ctx, world, C.c_void_p(send.data_ptr()), n, counts) == 0
        want, wc = partition_by_owner(host, list(g.reference_lengths()), world)
        assert g.exchange_owner_table(world) == [int(x) for x in __import__('chromap_amd.distributed', fromlist=['owner_table']).owner_table(list(g.reference_lengths()), world)]
        assert list(counts) == wc.tolist() and sum(counts) == k
        got = np.frombuffer(send[:k * 24].cpu().numpy().tobytes(), dtype=REC_DTYPE)
        # same multiset per destination rank (order inside a destination is irrelevant: a sort follows)
        lo = 0
        for r in range(world):
            a = np.sort(got[lo:lo + wc[r]], order=["read_id"])
            b = np.sort(want[lo:lo + wc[r]], order=["read_id"])
            assert np.array_equal(a, b)
            lo += wc[r]
    g.close()
