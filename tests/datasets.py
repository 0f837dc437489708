"""Regenerates the seeded synthetic inputs the golden outputs belong to (cached under
/tmp), checking input md5s against tests/golden/<case>.json."""
import gzip
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
CACHE = os.environ.get("CHROMAP_AMD_TEST_CACHE", "/tmp/chromap_amd_test_cache")

PRESET_KW = {"-q": "mapq_threshold"}


def md5(path):
    h = hashlib.md5()
    with open(path, "rb") as f:
        for b in iter(lambda: f.read(1 << 20), b""):
            h.update(b)
    return h.hexdigest()


def case_meta(name):
    with open(os.path.join(GOLD, name + ".json")) as f:
        return json.load(f)


def case_inputs(name):
    """returns (fa, r1, r2) paths for a golden case, generating them if needed"""
    meta = case_meta(name)
    if meta["generator_args"] is None:
        d = os.path.join(GOLD, "toy")
        paths = [os.path.join(d, f) for f in ("ref.fa", "read1.fq", "read2.fq")]
    else:
        key = hashlib.md5(" ".join(meta["generator_args"]).encode()).hexdigest()[:12]
        d = os.path.join(CACHE, key)
        paths = [os.path.join(d, f) for f in ("d.fa", "d_1.fq", "d_2.fq")]
        if not all(os.path.exists(p) for p in paths):
            os.makedirs(d, exist_ok=True)
            subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_synth.py"), "--out",
                                   os.path.join(d, "d")] + meta["generator_args"])
    got = {"fa": md5(paths[0]), "r1": md5(paths[1]), "r2": md5(paths[2])}
    if "bc" in meta["input_md5"]:
        got["bc"] = md5(os.path.join(d, "d_bc.fq"))
        got["whitelist"] = md5(os.path.join(d, "d.whitelist.txt"))
    if got != meta["input_md5"]:
        raise RuntimeError("regenerated inputs of golden case %s differ from the ones the golden output was "
                           "made from (numpy RNG stream changed?)" % name)
    return paths


def case_barcode_inputs(name):
    """(barcode fastq, whitelist) of a single-cell golden case"""
    fa, _, _ = case_inputs(name)
    d = os.path.dirname(fa)
    return os.path.join(d, "d_bc.fq"), os.path.join(d, "d.whitelist.txt")


def single_end_mate(name):
    return case_meta(name).get("single_end_mate", 0)


def has_barcodes(name):
    return "bc" in case_meta(name)["input_md5"]


def case_golden_bed(name):
    ext = ".sam.gz" if is_sam(name) else ".pairs.gz" if is_hic(name) else ".bed.gz"
    with gzip.open(os.path.join(GOLD, name + ext), "rb") as f:
        return f.read()


def is_sam(name):
    return "--SAM" in case_meta(name)["chromap_flags"]


def chr_order_ranks(order, names):
    """Chromap::GenerateCustomRidRanks (chromap.cc:867-913): rank per reference sequence"""
    pos = {n: i for i, n in enumerate(order)}
    ranks = [pos.get(n.decode() if isinstance(n, bytes) else n, -1) for n in names]
    k = len(pos)
    for i, r in enumerate(ranks):
        if r < 0:
            ranks[i] = k
            k += 1
    assert k == len(names), "unknown chromosome names in the order"
    return ranks


def is_tagalign(name):
    return "--TagAlign" in case_meta(name)["chromap_flags"]


def is_hic(name):
    return "hic" in case_meta(name)["chromap_flags"]


def flags_to_params(flags):
    """chromap CLI flags of a golden case -> (preset, overrides dict)"""
    preset = None
    kw = {}
    i = 0
    while i < len(flags):
        if flags[i] == "--preset":
            preset = flags[i + 1]
            i += 2
        elif flags[i] == "--bc-error-threshold":
            kw["bc_error_threshold"] = int(flags[i + 1])
            i += 2
        elif flags[i] == "--remove-pcr-duplicates-at-bulk-level":
            kw["dedup_at_bulk_level"] = 1
            i += 1
        elif flags[i] == "--chr-order":
            kw["chr_order"] = flags[i + 1].split(",")
            i += 2
        elif flags[i] == "--pairs-natural-chr-order":
            kw["pairs_order"] = flags[i + 1].split(",")
            i += 2
        elif flags[i] == "--SAM":
            kw["output_format"] = 1
            i += 1
        elif flags[i] == "--TagAlign":  # text format only (is_tagalign)
            i += 1
        elif flags[i] == "-l":
            kw["max_insert_size"] = int(flags[i + 1])
            i += 2
        elif flags[i] == "-e":
            kw["error_threshold"] = int(flags[i + 1])
            i += 2
        elif flags[i] == "-s":
            kw["min_num_seeds"] = int(flags[i + 1])
            i += 2
        elif flags[i] == "-f":
            f0, f1 = flags[i + 1].split(",")
            kw["max_seed_frequency0"], kw["max_seed_frequency1"] = int(f0), int(f1)
            i += 2
        elif flags[i] == "--min-read-length":
            kw["min_read_length"] = int(flags[i + 1])
            i += 2
        elif flags[i] == "-n":
            kw["max_num_best_mappings"] = int(flags[i + 1])
            i += 2
        elif flags[i] == "--drop-repetitive-reads":
            kw["drop_repetitive_reads"] = int(flags[i + 1])
            i += 2
        elif flags[i] in ("--remove-pcr-duplicates", "--Tn5-shift", "--trim-adapters", "--low-mem"):
            kw[{"--remove-pcr-duplicates": "remove_pcr_duplicates", "--Tn5-shift": "tn5_shift",
                "--trim-adapters": "trim_adapters", "--low-mem": "low_memory_mode"}[flags[i]]] = 1
            i += 1
        elif flags[i] == "-q":  # noqa: E501
            kw["mapq_threshold"] = int(flags[i + 1])
            i += 2
        else:
            raise ValueError(flags[i])
    return preset, kw


ALL_CASES = sorted(f[:-5] for f in os.listdir(GOLD) if f.endswith(".json"))
SAM_CASES = [c for c in ALL_CASES if is_sam(c) and not has_barcodes(c) and not is_hic(c)]
HIC_SAM_CASES = [c for c in ALL_CASES if is_sam(c) and is_hic(c)]
SAM_BC_CASES = [c for c in ALL_CASES if is_sam(c) and has_barcodes(c)]
# BED text through the product's writers (host and device); TagAlign cases of the same record types are listed apart
BED_CASES = [c for c in ALL_CASES if not is_hic(c) and not has_barcodes(c) and not single_end_mate(c) and not is_sam(c) and not is_tagalign(c)]
SE_CASES = [c for c in ALL_CASES if single_end_mate(c) and not is_sam(c) and not has_barcodes(c) and not is_tagalign(c)]
BC_CASES = [c for c in ALL_CASES if has_barcodes(c) and not is_sam(c) and not single_end_mate(c) and not is_tagalign(c)]
SE_BC_CASES = [c for c in ALL_CASES if has_barcodes(c) and not is_sam(c) and single_end_mate(c)]
TAGALIGN_CASES = [c for c in ALL_CASES if is_tagalign(c) and not (has_barcodes(c) and single_end_mate(c))]
HIC_CASES = [c for c in ALL_CASES if is_hic(c) and not is_sam(c)]


def case_index(name):
    """index file for a golden case, built with the oracle's indexer (the occupied buckets,
    flags and occurrence table equal the reference's file; tests/test_oracle_golden.py)"""
    import ctypes as C
    import oracle_lib as ol
    fa, _, _ = case_inputs(name)
    meta = case_meta(name)
    key = meta["input_md5"]["fa"][:12]
    os.makedirs(CACHE, exist_ok=True)
    path = os.path.join(CACHE, "idx_" + key + ".idx")
    if not os.path.exists(path):
        L = ol.lib()
        ref = ol.OraRef()
        assert L.ora_ref_load(fa.encode(), C.byref(ref)) == 0
        idx = ol.OraIndex()
        assert L.ora_index_build(C.byref(ref), 17, 7, C.byref(idx)) == 0
        assert L.ora_index_save((path + ".tmp").encode(), C.byref(idx)) == 0
        os.replace(path + ".tmp", path)
        L.ora_index_free(C.byref(idx))
        L.ora_ref_free(C.byref(ref))
    return path
