"""Parity of the HIP path (through the C ABI) with the reference's golden outputs and with
the oracle, on the GPU box."""
import ctypes as C
import hashlib

import numpy as np
import pytest

import datasets
import oracle_lib as ol

pytestmark = pytest.mark.gpu


def _rec_tuple(r):
    return (r.read_id, r.rid, r.fragment_start, r.fragment_length, r.mapq, r.direction, r.is_unique, r.num_dups,
            getattr(r, "positive_alignment_length", None) if hasattr(r, "positive_alignment_length") else r.pos_aln_len,
            getattr(r, "negative_alignment_length", None) if hasattr(r, "negative_alignment_length") else r.neg_aln_len)


@pytest.mark.parametrize("case", datasets.BED_CASES)
def test_bed_and_counters_match_reference(case, tmp_path):
    from chromap_amd import ChromapGPU
    meta = datasets.case_meta(case)
    fa, r1, r2 = datasets.case_inputs(case)
    idx = datasets.case_index(case)
    preset, kw = datasets.flags_to_params(meta["chromap_flags"])
    g = ChromapGPU(idx, fa, preset=preset, **kw)
    b1, o1 = ol.read_fastx(r1)
    b2, o2 = ol.read_fastx(r2)
    rec, k = g.map_pairs(b1, o1, b2, o2)
    a = sorted(_rec_tuple(rec[i]) for i in range(k))  # before the writer: it sorts (and, in-memory flavour, shifts) in place
    out = str(tmp_path / "g.bed")
    g.write_bed(rec, k, out)
    got = open(out, "rb").read()
    assert hashlib.md5(got).hexdigest() == meta["bed_md5"]
    assert got == datasets.case_golden_bed(case)
    ref = meta["reference_stderr_counters"]
    s = g.stats.as_dict()
    for key in ("num_candidates", "num_mappings", "num_mapped_reads", "num_uniquely_mapped_reads"):
        assert s[key] == ref[key], key
    # record-level equality with the oracle (every field, including those BED does not print)
    o = ol.Oracle(idx, fa, ol.params(preset, **kw))
    orec, ok, ost, _ = o.map_pairs(b1, o1, b2, o2)
    assert ok == k
    b = sorted(_rec_tuple(orec[i]) for i in range(ok))
    assert a == b
    # the probe kernel visits exactly the buckets khash's probe sequence visits for the
    # first-round lookups (the oracle also counts re-probes of round 2 and rescue)
    assert s["num_minimizers"] == ost.num_minimizers
    g.close()
    o.close()


def test_empty_and_degenerate_batches():
    from chromap_amd import ChromapGPU
    fa, r1, r2 = datasets.case_inputs("toy_chip")
    idx = datasets.case_index("toy_chip")
    g = ChromapGPU(idx, fa, preset="chip")
    z8 = np.zeros(0, np.uint8)
    z32 = np.zeros(1, np.uint32)
    rec, k = g.map_pairs(z8, z32, z8, z32)
    assert k == 0
    # reads shorter than min_read_length, all-N reads, reads shorter than k
    seqs1 = [b"ACGT" * 5, b"N" * 60, b"ACGTACGTACGTAC", b"A" * 80]
    seqs2 = [b"ACGT" * 20, b"N" * 60, b"ACGTACGTACGTACGTACGTACGTACGTACGT", b"T" * 80]
    def pack(ss):
        off = np.zeros(len(ss) + 1, np.uint32)
        off[1:] = np.cumsum([len(s) for s in ss])
        return np.frombuffer(b"".join(ss), np.uint8).copy(), off
    b1, o1 = pack(seqs1)
    b2, o2 = pack(seqs2)
    rec, k = g.map_pairs(b1, o1, b2, o2)
    o = ol.Oracle(idx, fa, ol.params("chip"))
    orec, ok, _, _ = o.map_pairs(b1, o1, b2, o2)
    assert k == ok
    g.close()
    o.close()


@pytest.mark.parametrize("case", datasets.HIC_CASES)
def test_hic_pairs_match_reference(case, tmp_path):
    """--preset hic: split alignment (drop-off bit-vector verification), pairs output"""
    from chromap_amd import ChromapGPU
    meta = datasets.case_meta(case)
    fa, r1, r2 = datasets.case_inputs(case)
    idx = datasets.case_index(case)
    preset, kw = datasets.flags_to_params(meta["chromap_flags"])
    g = ChromapGPU(idx, fa, preset=preset, **kw)
    b1, o1 = ol.read_fastx(r1)
    b2, o2 = ol.read_fastx(r2)
    rec, k = g.map_pairs(b1, o1, b2, o2)
    out = str(tmp_path / "g.pairs")
    g.write_pairs(rec, k, ol.read_names(r1), out)
    got = open(out, "rb").read()
    assert hashlib.md5(got).hexdigest() == meta["bed_md5"]
    assert got == datasets.case_golden_bed(case)
    ref = meta["reference_stderr_counters"]
    s = g.stats.as_dict()
    for key in ("num_candidates", "num_mappings", "num_mapped_reads", "num_uniquely_mapped_reads"):
        assert s[key] == ref[key], key
    # the same text from the device: the records of the batch (still resident) join the store, sorted and rendered in HBM
    g.store_clear()
    assert g.store_append_resident() == k
    lines, nbytes = g.store_format_pairs(ol.read_names(r1))
    out2 = str(tmp_path / "d.pairs")
    g.write_pairs_header(out2)
    g.store_write_text(out2, append=True)
    got2 = open(out2, "rb").read()
    assert got2 == got and lines == got.count(b"\n") - sum(1 for ln in got.split(b"\n") if ln.startswith(b"#"))
    g.close()


@pytest.mark.parametrize("case", [c for c in datasets.BC_CASES if "bulk_level" not in c])
def test_barcoded_bed_matches_reference(case, tmp_path):
    """scATAC: whitelist + abundance + barcode correction (K6) on the device, barcoded BED"""
    from chromap_amd import ChromapGPU
    meta = datasets.case_meta(case)
    fa, r1, r2 = datasets.case_inputs(case)
    idx = datasets.case_index(case)
    bcf, wlf = datasets.case_barcode_inputs(case)
    preset, kw = datasets.flags_to_params(meta["chromap_flags"])
    g = ChromapGPU(idx, fa, preset=preset, **kw)
    b1, o1 = ol.read_fastx(r1)
    b2, o2 = ol.read_fastx(r2)
    bc, bcq, bco = ol.read_fastq_qual(bcf)
    g.set_whitelist_file(wlf, int(bco[1] - bco[0]))
    ns = g.compute_barcode_abundance(bc, bco)
    rec, k = g.map_pairs_barcoded(b1, o1, b2, o2, bc, bcq, bco)
    out = str(tmp_path / "g.bed")
    g.write_bed_bc(rec, k, out)
    got = open(out, "rb").read()
    assert hashlib.md5(got).hexdigest() == meta["bed_md5"]
    ref = meta["reference_stderr_counters"]
    s = g.stats.as_dict()
    assert ns == ref["num_barcode_in_whitelist"]  # abundance pre-pass counts the same barcodes
    for key in ("num_candidates", "num_mappings", "num_mapped_reads", "num_uniquely_mapped_reads",
                "num_barcode_in_whitelist", "num_corrected_barcode"):
        assert s[key] == ref[key], key
    g.close()


@pytest.mark.parametrize("case", datasets.SE_CASES)
def test_single_end_bed_matches_reference(case, tmp_path):
    from chromap_amd import ChromapGPU
    meta = datasets.case_meta(case)
    fa, r1, r2 = datasets.case_inputs(case)
    idx = datasets.case_index(case)
    preset, kw = datasets.flags_to_params(meta["chromap_flags"])
    g = ChromapGPU(idx, fa, preset=preset, **kw)
    b, off = ol.read_fastx(r1 if datasets.single_end_mate(case) == 1 else r2)
    rec, k = g.map_single(b, off)
    out = str(tmp_path / "g.bed")
    g.write_bed_se(rec, k, out)
    assert hashlib.md5(open(out, "rb").read()).hexdigest() == meta["bed_md5"]
    ref = meta["reference_stderr_counters"]
    s = g.stats.as_dict()
    for key in ("num_candidates", "num_mappings", "num_mapped_reads", "num_uniquely_mapped_reads"):
        assert s[key] == ref[key], key
    g.close()


import fuzz_data  # noqa: E402
import test_hostemu_fuzz as _fz  # noqa: E402


@pytest.mark.parametrize("cfg", fuzz_data.CONFIGS, ids=[str(c[0]) for c in fuzz_data.CONFIGS])
def test_fuzz_gpu_vs_oracle(cfg, tmp_path):
    """adversarial repeat-rich data: every record field and the counters, GPU vs oracle"""
    from chromap_amd import ChromapGPU

    def factory(idx, fa, preset, gkw, b1, o1, b2, o2):
        g = ChromapGPU(idx, fa, preset=preset, **gkw)
        rec, k = g.map_pairs(b1, o1, b2, o2)
        st = g.stats.as_dict()
        g.close()
        return rec, k, st
    _fz.run_case(factory, cfg, tmp_path)


HEAVY_SETTINGS = {
    "wave": {"heavy_last": 1},                                             # long hit lists: a wave each, heavy reads last
    "block": {"heavy_wave_max": 20, "heavy_last": 1},                      # ... a block each
    "block_big_lds": {"heavy_wave_max": 18, "heavy_block_max": 19},        # ... a block with the large LDS allocation
    "one_lane": {"heavy_wave_max": -1, "heavy_last": -1},                  # ... the sequential path in global memory
    "groups_of_16": {"heavy_mid_max": 256, "s3b_lane_cap": 4},             # ... four reads per wave (lists up to 256 hits here)
    "no_groups": {"heavy_mid_max": -1},                                    # ... without that class: a lane up to its LDS slots, waves beyond
    # the settings above sort hit lists and rescue hits as merges of their occurrence runs (cm_coop.h); the round-2 kernels:
    "wave_bitonic": {"heavy_last": 1, "coop": 0},
    "block_bitonic": {"heavy_wave_max": 20, "heavy_last": 1, "coop": 0},
    "block_big_lds_bitonic": {"heavy_wave_max": 18, "heavy_block_max": 19, "coop": 0},
    # run tables too small for most reads: the merge-sort kernels decline them (bitonic kernel over a device-side list; lane 0 for rescue hits)
    "declined": {"coop_run_table": 3},
    "declined_block": {"heavy_wave_max": 20, "coop_run_table": 3},
    # the index table re-hashed on the device into 4 times as many buckets: same lookups, fewer buckets visited
    "rehashed_table": {"probe_table_shift": 2},
    # the candidate arrays "left by the previous batch" far too small: the device raises the abort flag, every later kernel leaves,
    # the range is mapped again with exact sizes
    "capacity_guess_too_small": {"debug_candidate_capacity": 64},
    "no_speculative_sizes": {"speculative_sizes": 0},
    # the kernels of the long-list classes are launched for the classes the previous range used; -1: none -- the range finds items in
    # classes it did not launch for and is mapped again with all of them
    "launch_set_empty": {"speculative_sizes": -1},
    # the round-2 forms of the verification (alignments on the bytes instead of bit planes) and of the minimizer pass of reads
    # longer than 69 bases (count, scan, fill instead of the fused kernel with its global staging tile)
    "verify_on_bytes": {"verify_planes": 0},
    "long_reads_two_pass": {"long_read_fused": 0},
    "declined_hits_only": {"coop": 1, "coop_run_table": 3},
    "declined_rescue_only": {"coop": 2, "coop_run_table": 3},
}


@pytest.mark.parametrize("setting", sorted(HEAVY_SETTINGS))
@pytest.mark.parametrize("cfg", fuzz_data.CONFIGS[:4], ids=[str(c[0]) for c in fuzz_data.CONFIGS[:4]])
def test_fuzz_long_hit_lists_every_path(cfg, setting, tmp_path):
    """the cooperative kernel for long hit lists (k_s3b_heavy) in each of its size classes, and the heavy-last
    processing order of the later stages, on the repeat-rich fuzz data: every record field and the counters"""
    from chromap_amd import ChromapGPU

    def factory(idx, fa, preset, gkw, b1, o1, b2, o2):
        g = ChromapGPU(idx, fa, preset=preset, **gkw)
        for k_, v_ in HEAVY_SETTINGS[setting].items():
            g.set_option(k_, v_)
        rec, k = g.map_pairs(b1, o1, b2, o2)
        st = g.stats.as_dict()
        g.close()
        return rec, k, st
    _fz.run_case(factory, cfg, tmp_path)


@pytest.mark.parametrize("bc_err", [1, 2])
def test_barcode_correction_dense_whitelist_gpu(bc_err, tmp_path):
    """as tests/test_hostemu_fuzz.py::test_barcode_correction_dense_whitelist, on the device"""
    import bc_fuzz
    from chromap_amd import ChromapGPU
    fa, idx, b1, o1, b2, o2, bc, bcq, bco, wl = bc_fuzz.build(str(tmp_path))
    want, n_in, n_corr, _ = bc_fuzz.oracle_result(fa, idx, b1, o1, b2, o2, bc, bcq, bco, wl, bc_err)
    g = ChromapGPU(idx, fa, preset="atac", mapq_threshold=0, bc_error_threshold=bc_err)
    g.set_whitelist_file(wl, bc_fuzz.BC_LEN)
    g.compute_barcode_abundance(bc, bco)
    rec, k = g.map_pairs_barcoded(b1, o1, b2, o2, bc, bcq, bco)
    got = sorted((rec[i].r.read_id, rec[i].r.rid, rec[i].r.fragment_start, rec[i].r.fragment_length, rec[i].r.mapq,
                  rec[i].r.direction, rec[i].barcode) for i in range(k))
    s = g.stats.as_dict()
    assert (s["num_barcode_in_whitelist"], s["num_corrected_barcode"]) == (n_in, n_corr)
    assert got == want
    g.close()


def test_async_submit_wait_equals_sync():
    """one batch in flight on a worker thread of the library; the caller packs the next meanwhile"""
    from chromap_amd import ChromapGPU
    case = "s1_atac"
    fa, r1, r2 = datasets.case_inputs(case)
    g = ChromapGPU(datasets.case_index(case), fa, preset="atac")
    b1, o1 = ol.read_fastx(r1)
    b2, o2 = ol.read_fastx(r2)
    rec, k = g.map_pairs(b1, o1, b2, o2)
    want = sorted(_rec_tuple(rec[i]) for i in range(k))
    for _ in range(2):
        g.map_pairs_async(b1, o1, b2, o2)
        busy = sum(int(x) for x in o1[:1000])  # the caller is free to work here
        rec2, k2 = g.wait()
        assert k2 == k and busy >= 0
        assert sorted(_rec_tuple(rec2[i]) for i in range(k2)) == want
    with pytest.raises(Exception):
        g.wait()  # nothing in flight
    g.close()


def _long_read_batch(fa_path, rng, n, lmin, lmax):
    """pairs of long reads (lmin..lmax bases) cut from the reference with substitutions"""
    seqs = [s for s in open(fa_path, "rb").read().split(b">")[1:]]
    chroms = [b"".join(s.split(b"\n")[1:]) for s in seqs]
    comp = bytes.maketrans(b"ACGTNacgtn", b"TGCANtgcan")
    r1s, r2s = [], []
    for _ in range(n):
        c = chroms[rng.integers(0, len(chroms))]
        l1, l2 = int(rng.integers(lmin, lmax + 1)), int(rng.integers(lmin, lmax + 1))
        fl = max(l1, l2) + int(rng.integers(0, 400))
        st = int(rng.integers(0, len(c) - fl))
        frag = bytearray(c[st:st + fl].upper())
        for p in rng.integers(0, fl, min(5, fl // 60)):
            frag[p] = ord("ACGT"[rng.integers(0, 4)])
        frag = bytes(frag)
        r1s.append(frag[:l1])
        r2s.append(frag.translate(comp)[::-1][:l2])
    def pack(ss):
        off = np.zeros(len(ss) + 1, np.uint32)
        off[1:] = np.cumsum([len(s) for s in ss])
        return np.frombuffer(b"".join(ss), np.uint8).copy(), off
    return pack(r1s) + pack(r2s)


@pytest.mark.parametrize("lmin,lmax", [(240, 400), (900, 1000)])
def test_long_reads_match_oracle(lmin, lmax):
    """read lengths well above the usual 50-150: LDS staging geometry, trimming beyond the 2-bit fast
    path, long Myers bands"""
    from chromap_amd import ChromapGPU
    case = "s3_chip"
    fa, _, _ = datasets.case_inputs(case)
    idx = datasets.case_index(case)
    b1, o1, b2, o2 = _long_read_batch(fa, np.random.default_rng(lmin), 600, lmin, lmax)
    for preset in ("chip", "atac"):
        g = ChromapGPU(idx, fa, preset=preset, mapq_threshold=0)
        o = ol.Oracle(idx, fa, ol.params(preset, mapq_threshold=0))
        rec, k = g.map_pairs(b1, o1, b2, o2)
        orec, ok, ost, _ = o.map_pairs(b1, o1, b2, o2)
        assert k == ok and k > 300
        assert sorted(_rec_tuple(rec[i]) for i in range(k)) == sorted(_rec_tuple(orec[i]) for i in range(ok))
        g.close()
        o.close()


def test_reads_beyond_the_supported_length_are_rejected():
    from chromap_amd import ChromapError, ChromapGPU
    fa, _, _ = datasets.case_inputs("toy_chip")
    g = ChromapGPU(datasets.case_index("toy_chip"), fa, preset="chip")
    b = np.frombuffer(b"ACGT" * 1000, np.uint8).copy()
    off = np.array([0, len(b)], np.uint32)
    with pytest.raises(ChromapError, match="longer than"):
        g.map_pairs(b, off, b, off)
    g.close()


def test_shared_index_contexts_map_concurrently():
    """cmgpu_create_shared: two contexts over one resident index, one host thread each"""
    import threading
    from chromap_amd import ChromapGPU
    case = "s1_atac"
    fa, r1, r2 = datasets.case_inputs(case)
    g0 = ChromapGPU(datasets.case_index(case), fa, preset="atac")
    g1 = ChromapGPU(shared_from=g0)
    b1, o1 = ol.read_fastx(r1)
    b2, o2 = ol.read_fastx(r2)
    half = (len(o1) - 1) // 2
    rec, k = g0.map_pairs(b1, o1, b2, o2)
    want = sorted(_rec_tuple(rec[i]) for i in range(k))
    out = [None, None]

    def run(i, g, lo, hi):
        for _ in range(3):
            r, kk = g.map_pairs(b1, o1[lo:hi + 1], b2, o2[lo:hi + 1], first_read_id=lo)
            out[i] = [_rec_tuple(r[j]) for j in range(kk)]
    th = [threading.Thread(target=run, args=(0, g0, 0, half)), threading.Thread(target=run, args=(1, g1, half, len(o1) - 1))]
    [t.start() for t in th]
    [t.join() for t in th]
    # a split at an arbitrary pair changes which multi-mappers share an RNG chunk; compare the pairs with one best mapping
    got = sorted(out[0] + out[1])
    uniq = lambda v: [t for t in v if t[4] > 0]
    assert uniq(got) == uniq(want)
    # the child views the parent's re-hashed probe table: the parent may not rebuild or release it while the child lives
    from chromap_amd.mapper import ChromapError
    with pytest.raises(ChromapError, match="before cmgpu_create_shared"):
        g0.set_option("probe_table_shift", 0)
    g1.close()
    g0.set_option("probe_table_shift", 2)  # (no child left: allowed again)
    rec2, k2 = g0.map_pairs(b1, o1, b2, o2)
    assert sorted(_rec_tuple(rec2[i]) for i in range(k2)) == want
    g0.close()
