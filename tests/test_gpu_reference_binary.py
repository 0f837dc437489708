"""chromap-amd against the REFERENCE BINARY itself (oracle/_ref/chromap travels with the repo) on
the GPU box, at sizes well above the committed golden cases: adversarial synthetic data from
tools/gen_synth.py (repeats, N runs, indels, adapters, duplicates, chimeric Hi-C reads, barcodes
with errors), index built on the device and loaded by BOTH programs, outputs compared byte for
byte.  Skipped where the reference binary is not present."""
import os
import subprocess
import sys

import pytest

import datasets as ds

pytestmark = pytest.mark.gpu
CLI = os.path.join(ds.ROOT, "chromap_amd", "chromap-amd")
REF = os.path.join(ds.ROOT, "oracle", "_ref", "chromap")
GEN = os.path.join(ds.ROOT, "tools", "gen_synth.py")

DATA = {
    "short": ["--genome", "20000000", "--chroms", "6", "--pairs", "150000", "--readlen", "50", "--frag-min", "35", "--seed", "101",
              "--barcodes", "5000"],
    "long": ["--genome", "20000000", "--chroms", "6", "--pairs", "100000", "--readlen", "150", "--frag-min", "300", "--frag-max", "800",
             "--hic", "--seed", "102", "--indel", "0.002"],
    "long_bc": ["--genome", "20000000", "--chroms", "6", "--pairs", "60000", "--readlen", "150", "--frag-min", "300", "--frag-max", "800",
                "--hic", "--seed", "104", "--indel", "0.002", "--barcodes", "2000"],
    "mid": ["--genome", "20000000", "--chroms", "6", "--pairs", "100000", "--readlen", "100", "--seed", "103", "--indel", "0.003",
            "--sub", "0.02"],
}
# name -> (dataset, flags, single-end?, barcodes?)
RUNS = {
    "atac_pe": ("short", ["--preset", "atac"], False, False),
    "atac_pe_q0_inmem": ("short", ["-l", "2000", "--trim-adapters", "--remove-pcr-duplicates", "--Tn5-shift", "-q", "0"], False, False),
    "atac_barcodes": ("short", ["--preset", "atac"], False, True),
    "barcodes_bulk_level_dedup": ("short", ["--preset", "atac", "--remove-pcr-duplicates-at-bulk-level", "-q", "0"], False, True),
    "chip_se": ("short", ["--preset", "chip"], True, False),
    "atac_tagalign": ("short", ["--preset", "atac", "--TagAlign"], False, False),
    "atac_tagalign_barcodes": ("short", ["--preset", "atac", "--TagAlign"], False, True),
    "chip_tagalign_se_q0": ("short", ["--preset", "chip", "--TagAlign", "-q", "0"], True, False),
    "hic_pairs": ("long", ["--preset", "hic"], False, False),
    "chip_sam": ("mid", ["--preset", "chip", "--SAM"], False, False),
    "sam_se_q0": ("mid", ["--SAM", "-q", "0"], True, False),
    "sam_barcodes": ("short", ["--preset", "atac", "--SAM"], False, True),
    "sam_se_barcodes_q0": ("short", ["--SAM", "--remove-pcr-duplicates", "-q", "0"], True, True),
    # (single-end single-cell data in low-memory mode costs the reference ~30-60 s of fixed time in its merge; in-memory here)
    "tagalign_se_barcodes": ("short", ["--TagAlign", "--remove-pcr-duplicates", "--Tn5-shift", "-q", "0"], True, True),
    # off the presets: the 8-lane verification form (error threshold < 8 without split alignment, alignment.cc:503-654), wide
    # bands, min seeds, seed-frequency caps (second round, draft_mapping_generator.cc:159-357), insert limit, --min-read-length
    # (K0's seed length, chromap.cc:176-216), several best mappings
    "e5_pe_q0": ("mid", ["-e", "5", "-q", "0"], False, False),
    "e5_atac": ("short", ["--preset", "atac", "-e", "5"], False, False),
    "e3_se_q0": ("mid", ["--preset", "chip", "-e", "3", "-q", "0"], True, False),
    "e5_sam_q0": ("mid", ["-e", "5", "--SAM", "-q", "0"], False, False),
    "e12_chip_q0": ("mid", ["--preset", "chip", "-e", "12", "-q", "0"], False, False),
    "e6_hic_q0": ("long", ["--preset", "hic", "-e", "6", "-q", "0"], False, False),
    "s3_l300_q0": ("short", ["--preset", "chip", "-s", "3", "-l", "300", "-q", "0"], False, False),
    "s1_q0": ("short", ["--preset", "atac", "-s", "1", "-q", "0"], False, False),
    "f40_90_q0": ("short", ["--preset", "atac", "-f", "40,90", "-q", "0"], False, False),
    "f5_20_se_q0": ("short", ["--preset", "chip", "-f", "5,20", "-q", "0"], True, False),
    "minlen40_q0": ("short", ["--preset", "atac", "--min-read-length", "40", "-q", "0"], False, False),
    "minlen20_barcodes": ("short", ["--preset", "atac", "--min-read-length", "20"], False, True),
    "n5_q0": ("short", ["--preset", "atac", "-n", "5", "-q", "0"], False, False),
    "n5_e5_se_q0": ("short", ["--preset", "chip", "-n", "5", "-e", "5", "-q", "0"], True, False),
    # more best mappings than the 64 record slots per read rounds 1-5 had (the data has a family of ~300 exact copies)
    "n100_q0": ("short", ["--preset", "atac", "-n", "100", "-q", "0"], False, False),
    "n300_se_q0": ("short", ["--preset", "chip", "-n", "300", "-q", "0"], True, False),
    # --pairs on the ordinary (non-split) pairing: MapPairedEndReads<PairsMapping> (chromap_driver.cc:748-751)
    "pairs_nonsplit_q0": ("mid", ["--pairs", "-q", "0"], False, False),
    "pairs_nonsplit_atac": ("short", ["--preset", "atac", "--pairs"], False, False),
    # pairs with cell barcodes: the barcodes decide which pairs are mapped (correction against the whitelist) and are not printed
    "pairs_nonsplit_barcodes": ("short", ["--preset", "atac", "--pairs"], False, True),
    "hic_pairs_barcodes": ("long_bc", ["--preset", "hic"], False, True),
    "hic_pairs_dedup_q0": ("long", ["--preset", "hic", "--remove-pcr-duplicates", "-q", "0"], False, False),
    "pairs_nonsplit_n3_q0": ("short", ["--pairs", "-l", "1500", "-n", "3", "-q", "0", "--remove-pcr-duplicates"], False, False),
}


@pytest.fixture(scope="module")
def data(tmp_path_factory):
    if not os.path.exists(REF):
        pytest.skip("built reference binary not present")
    assert os.path.exists(CLI)
    made = {}

    def get(name):
        if name not in made:
            d = str(tmp_path_factory.mktemp(name))
            pre = os.path.join(d, "d")
            subprocess.check_call([sys.executable, GEN, "--out", pre] + DATA[name])
            idx = pre + ".idx"
            subprocess.run([CLI, "-i", "-r", pre + ".fa", "-o", idx], check=True, stderr=subprocess.PIPE)
            made[name] = (pre, idx)
        return made[name]
    return get


@pytest.mark.parametrize("run", sorted(RUNS))
def test_output_equals_reference_binary(run, data, tmp_path):
    dset, flags, single, barcoded = RUNS[run]
    pre, idx = data(dset)
    reads = ["-1", pre + "_1.fq"] + ([] if single else ["-2", pre + "_2.fq"])
    if barcoded:
        reads += ["-b", pre + "_bc.fq", "--barcode-whitelist", pre + ".whitelist.txt"]
    out_ref, out_gpu = str(tmp_path / "ref.out"), str(tmp_path / "gpu.out")
    common = flags + ["-x", idx, "-r", pre + ".fa"] + reads
    r = subprocess.run([REF] + common + ["-o", out_ref, "-t", "32"], stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    g = subprocess.run([CLI] + common + ["-o", out_gpu], stderr=subprocess.PIPE)
    assert g.returncode == 0, g.stderr.decode()[-2000:]
    assert os.path.getsize(out_ref) > 100000
    assert ds.md5(out_gpu) == ds.md5(out_ref)


@pytest.mark.parametrize("k,w", [(21, 11), (15, 5), (27, 7)])
def test_other_kmer_and_window_sizes(k, w, data, tmp_path):
    """non-default index geometry: ring-buffer minimizers (w != 7), two-pass kernels (k > 26), index
    built by either program and read by the other"""
    pre, _ = data("short")
    idx_ref, idx_gpu = str(tmp_path / "ref.idx"), str(tmp_path / "gpu.idx")
    subprocess.run([REF, "-i", "-k", str(k), "-w", str(w), "-r", pre + ".fa", "-o", idx_ref], check=True, stderr=subprocess.PIPE)
    subprocess.run([CLI, "-i", "-k", str(k), "-w", str(w), "-r", pre + ".fa", "-o", idx_gpu], check=True, stderr=subprocess.PIPE)
    reads = ["-1", pre + "_1.fq", "-2", pre + "_2.fq"]
    outs = {}
    for prog, tag in ((REF, "ref"), (CLI, "gpu")):
        for idx, itag in ((idx_ref, "refidx"), (idx_gpu, "gpuidx")):
            out = str(tmp_path / ("%s_%s.bed" % (tag, itag)))
            cmd = [prog, "--preset", "atac", "-q", "0", "-x", idx, "-r", pre + ".fa"] + reads + ["-o", out] + (["-t", "32"] if prog == REF else [])
            r = subprocess.run(cmd, stderr=subprocess.PIPE)
            assert r.returncode == 0, r.stderr.decode()[-2000:]
            outs[(tag, itag)] = ds.md5(out)
    assert len(set(outs.values())) == 1, outs


def test_multi_batch_q0_multimappers(tmp_path):
    """1.3 M pairs = three reference read batches (500 000 + 500 000 + 300 000) at -q 0 on a repeat-rich
    genome: ~10^5 multi-mapped reads whose reported position depends on the per-task reservoir RNG
    (DESIGN.md section 2).  One 1.3 M-pair call and explicit 500 000-pair calls must both equal the reference."""
    if not os.path.exists(REF):
        pytest.skip("built reference binary not present")
    pre = str(tmp_path / "d")
    subprocess.check_call([sys.executable, GEN, "--out", pre, "--genome", "30000000", "--chroms", "6", "--pairs", "1300000",
                           "--readlen", "50", "--frag-min", "35", "--seed", "77"])
    idx = pre + ".idx"
    subprocess.run([CLI, "-i", "-r", pre + ".fa", "-o", idx], check=True, stderr=subprocess.PIPE)
    common = ["--preset", "atac", "-q", "0", "-x", idx, "-r", pre + ".fa", "-1", pre + "_1.fq", "-2", pre + "_2.fq"]
    subprocess.run([REF] + common + ["-o", pre + ".ref.bed", "-t", "64"], check=True, stderr=subprocess.PIPE)
    subprocess.run([CLI] + common + ["-o", pre + ".gpu.bed"], check=True, stderr=subprocess.PIPE)
    subprocess.run([CLI] + common + ["--batch-pairs", "500000", "-o", pre + ".gpu2.bed"], check=True, stderr=subprocess.PIPE)
    want = ds.md5(pre + ".ref.bed")
    assert ds.md5(pre + ".gpu.bed") == want
    assert ds.md5(pre + ".gpu2.bed") == want
    mapq0 = sum(1 for ln in open(pre + ".ref.bed", "rb") if ln.split(b"\t")[4] == b"0")
    assert mapq0 > 10000


def test_read_format_ranges_and_strand(data, tmp_path):
    """--read-format: barcodes embedded reverse-complemented in a longer read, reads cut to ranges
    (SequenceEffectiveRange); device ingest and the host parser against the reference binary"""
    pre, idx = data("short")
    comp = bytes.maketrans(b"ACGTN", b"TGCAN")
    bc2 = str(tmp_path / "bc_padded.fq")
    with open(pre + "_bc.fq", "rb") as f, open(bc2, "wb") as g:
        lines = f.read().split(b"\n")
        for i in range(0, len(lines) - 3, 4):
            seq, qual = lines[i + 1], lines[i + 3]
            g.write(lines[i] + b"\n" + b"GGA" + seq.translate(comp)[::-1] + b"TC\n+\n" + b"#!#" + qual[::-1] + b"!!\n")
    fmt = "bc:3:18:-,r1:0:44,r2:2:-1"
    common = ["--preset", "atac", "--read-format", fmt, "-x", idx, "-r", pre + ".fa", "-1", pre + "_1.fq", "-2", pre + "_2.fq", "-b", bc2,
              "--barcode-whitelist", pre + ".whitelist.txt"]
    outs = []
    for prog, extra in ((REF, ["-t", "32"]), (CLI, []), (CLI, ["--host-ingest"])):
        out = str(tmp_path / ("o%d.bed" % len(outs)))
        r = subprocess.run([prog] + common + extra + ["-o", out], stderr=subprocess.PIPE)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        outs.append(ds.md5(out))
    assert os.path.getsize(out) > 100000
    assert outs[1] == outs[0] and outs[2] == outs[0], outs


def test_barcode_translate(data, tmp_path):
    """--barcode-translate: column 4 of the single-cell BED goes through the translation table"""
    pre, idx = data("short")
    table = str(tmp_path / "tr.tsv")
    with open(pre + ".whitelist.txt") as f, open(table, "w") as g:
        for i, ln in enumerate(f):
            g.write("CELL%05d\t%s\n" % (i, ln.strip()))
    common = ["--preset", "atac", "--barcode-translate", table, "-x", idx, "-r", pre + ".fa", "-1", pre + "_1.fq", "-2", pre + "_2.fq",
              "-b", pre + "_bc.fq", "--barcode-whitelist", pre + ".whitelist.txt"]
    out_ref, out_gpu = str(tmp_path / "r.bed"), str(tmp_path / "g.bed")
    subprocess.run([REF] + common + ["-o", out_ref, "-t", "32"], check=True, stderr=subprocess.PIPE)
    subprocess.run([CLI] + common + ["-o", out_gpu], check=True, stderr=subprocess.PIPE)
    assert b"CELL" in open(out_ref, "rb").read(4096)
    assert ds.md5(out_gpu) == ds.md5(out_ref)


def test_barcodes_without_whitelist(data, tmp_path):
    """-b without --barcode-whitelist: every barcode is kept as read (no correction), cell-level duplicate removal"""
    pre, idx = data("short")
    common = ["--preset", "atac", "-x", idx, "-r", pre + ".fa", "-1", pre + "_1.fq", "-2", pre + "_2.fq", "-b", pre + "_bc.fq"]
    outs = []
    for prog, extra in ((REF, ["-t", "32"]), (CLI, []), (CLI, ["--host-ingest"])):
        out = str(tmp_path / ("o%d.bed" % len(outs)))
        r = subprocess.run([prog] + common + extra + ["-o", out], stderr=subprocess.PIPE)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        outs.append(ds.md5(out))
    assert outs[1] == outs[0] and outs[2] == outs[0], outs


@pytest.mark.parametrize("flags", [["--preset", "atac"], ["--preset", "atac", "--remove-pcr-duplicates-at-bulk-level", "-q", "0"],
                                   ["--remove-pcr-duplicates", "--Tn5-shift", "-q", "0"]],
                         ids=["cell_level", "bulk_level_q0", "in_memory_q0"])
def test_single_end_with_barcodes(flags, data, tmp_path):
    """single-end single-cell data (MappingWithBarcode): device ingest and the host parser against the reference binary"""
    pre, idx = data("short")
    common = flags + ["-x", idx, "-r", pre + ".fa", "-1", pre + "_1.fq", "-b", pre + "_bc.fq", "--barcode-whitelist", pre + ".whitelist.txt"]
    outs = []
    for prog, extra in ((REF, ["-t", "32"]), (CLI, []), (CLI, ["--host-ingest"])):
        out = str(tmp_path / ("o%d.bed" % len(outs)))
        r = subprocess.run([prog] + common + extra + ["-o", out], stderr=subprocess.PIPE)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        outs.append(ds.md5(out))
    assert os.path.getsize(out) > 100000
    assert outs[1] == outs[0] and outs[2] == outs[0], outs


def test_multiple_input_files(data, tmp_path):
    """-1 a,b -2 c,d -b e,f: the reference keeps numbering reads across files (read ids feed the multi-mapper
    RNG seeds) and restarts its 500 000-read batches per file; one of the files is gzip-compressed"""
    import gzip
    pre, idx = data("short")
    cut = 60001 * 4  # lines: an uneven split
    parts = {}
    for tag in ("_1", "_2", "_bc"):
        lines = open(pre + tag + ".fq", "rb").read().split(b"\n")
        a, b = str(tmp_path / ("a%s.fq" % tag)), str(tmp_path / ("b%s.fq.gz" % tag))
        open(a, "wb").write(b"\n".join(lines[:cut]) + b"\n")
        with gzip.open(b, "wb", compresslevel=1) as g:
            g.write(b"\n".join(lines[cut:]))
        parts[tag] = a + "," + b
    common = ["--preset", "atac", "-q", "0", "-x", idx, "-r", pre + ".fa", "-1", parts["_1"], "-2", parts["_2"], "-b", parts["_bc"],
              "--barcode-whitelist", pre + ".whitelist.txt"]
    outs = []
    for prog, extra in ((REF, ["-t", "32"]), (CLI, []), (CLI, ["--host-ingest"])):
        out = str(tmp_path / ("o%d.bed" % len(outs)))
        r = subprocess.run([prog] + common + extra + ["-o", out], stderr=subprocess.PIPE)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        outs.append(ds.md5(out))
    assert os.path.getsize(out) > 100000
    assert outs[1] == outs[0] and outs[2] == outs[0], outs


def test_single_end_barcode_translate_and_skip_check(data, tmp_path):
    """--barcode-translate on single-end single-cell BED, --skip-barcode-check accepted"""
    pre, idx = data("short")
    table = str(tmp_path / "tr.tsv")
    with open(pre + ".whitelist.txt") as f, open(table, "w") as g:
        for i, ln in enumerate(f):
            g.write("C%d\t%s\n" % (i, ln.strip()))
    common = ["--remove-pcr-duplicates", "--barcode-translate", table, "--skip-barcode-check", "-x", idx, "-r", pre + ".fa", "-1", pre + "_2.fq",
              "-b", pre + "_bc.fq", "--barcode-whitelist", pre + ".whitelist.txt"]
    out_ref, out_gpu = str(tmp_path / "r.bed"), str(tmp_path / "g.bed")
    subprocess.run([REF] + common + ["-o", out_ref, "-t", "32"], check=True, stderr=subprocess.PIPE)
    subprocess.run([CLI] + common + ["-o", out_gpu], check=True, stderr=subprocess.PIPE)
    assert os.path.getsize(out_ref) > 100000
    assert ds.md5(out_gpu) == ds.md5(out_ref)


def test_instrument_shaped_million_pairs(tmp_path):
    """One million pairs shaped like an instrument's output (tools/gen_real.py): read lengths mixed per read, 36-151 bases,
    per-base qualities from a position-dependent model with the substitutions drawn from them, 0.5 % of the bases N, adapter
    read-through; a reference with 10 % soft-masked (lower-case) segments, IUPAC ambiguity codes and N runs.  --preset atac
    (trimming, duplicate removal, Tn5 shift), --preset chip -q 0 (every mapping, MAPQ 0 included) and --SAM on a slice, BGZF
    and plain text inputs: byte-identical to the reference binary."""
    if not os.path.exists(REF):
        pytest.skip("built reference binary not present")
    pre = str(tmp_path / "d")
    subprocess.check_call([sys.executable, os.path.join(ds.ROOT, "tools", "gen_real.py"), "--out", pre, "--genome", "60000000", "--chroms", "8",
                           "--pairs", "1000000", "--seed", "606"])
    idx = pre + ".idx"
    subprocess.run([CLI, "-i", "-r", pre + ".fa", "-o", idx], check=True, stderr=subprocess.PIPE)
    reads = ["-x", idx, "-r", pre + ".fa", "-1", pre + "_1.fq", "-2", pre + "_2.fq"]
    lens = set()
    with open(pre + "_1.fq", "rb") as f:
        for i, ln in enumerate(f):
            if i % 4 == 1:
                lens.add(len(ln) - 1)
            if i > 40000:
                break
    assert min(lens) <= 40 and max(lens) == 151 and len(lens) > 100  # mixed lengths, as generated
    for name, flags in (("atac", ["--preset", "atac"]), ("chip_q0", ["--preset", "chip", "-q", "0"])):
        out_ref, out_gpu = pre + "." + name + ".ref", pre + "." + name + ".gpu"
        r = subprocess.run([REF] + flags + reads + ["-o", out_ref, "-t", "64"], stderr=subprocess.PIPE)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        g = subprocess.run([CLI] + flags + reads + ["-o", out_gpu], stderr=subprocess.PIPE)
        assert g.returncode == 0, g.stderr.decode()[-2000:]
        assert os.path.getsize(out_ref) > 20_000_000, name
        assert ds.md5(out_gpu) == ds.md5(out_ref), name
    # --SAM (qualities and read names in the output) on the first 100 000 pairs
    for tag in ("_1", "_2"):
        with open(pre + tag + ".fq", "rb") as f, open(pre + tag + ".head.fq", "wb") as g:
            for i, ln in enumerate(f):
                if i >= 400000:
                    break
                g.write(ln)
    sreads = ["-x", idx, "-r", pre + ".fa", "-1", pre + "_1.head.fq", "-2", pre + "_2.head.fq"]
    r = subprocess.run([REF, "--preset", "chip", "--SAM"] + sreads + ["-o", pre + ".ref.sam", "-t", "64"], stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    g = subprocess.run([CLI, "--preset", "chip", "--SAM"] + sreads + ["-o", pre + ".gpu.sam"], stderr=subprocess.PIPE)
    assert g.returncode == 0, g.stderr.decode()[-2000:]
    assert ds.md5(pre + ".gpu.sam") == ds.md5(pre + ".ref.sam")


@pytest.mark.parametrize("single", [False, True], ids=["paired", "single_end"])
def test_sam_barcode_translate(single, data, tmp_path):
    """--SAM with --barcode-translate: the CB:Z: tag of every line goes through the translation table (a gzip-compressed one,
    which the reference reads with gzopen); a table that lacks a barcode ends both programs with the same message"""
    import gzip
    pre, idx = data("short")
    table = str(tmp_path / "tr.tsv.gz")
    with open(pre + ".whitelist.txt") as f, gzip.open(table, "wt") as g:
        for i, ln in enumerate(f):
            g.write("CELL%05d\t%s\n" % (i, ln.strip()))
    reads = ["-1", pre + "_1.fq"] + ([] if single else ["-2", pre + "_2.fq"])
    common = ["--preset", "atac", "--SAM", "--barcode-translate", table, "-x", idx, "-r", pre + ".fa"] + reads + \
             ["-b", pre + "_bc.fq", "--barcode-whitelist", pre + ".whitelist.txt"]
    out_ref, out_gpu = str(tmp_path / "r.sam"), str(tmp_path / "g.sam")
    subprocess.run([REF] + common + ["-o", out_ref, "-t", "32"], check=True, stderr=subprocess.PIPE)
    subprocess.run([CLI] + common + ["-o", out_gpu], check=True, stderr=subprocess.PIPE)
    assert b"CB:Z:CELL" in open(out_ref, "rb").read(1 << 16)
    assert os.path.getsize(out_ref) > 1000000
    assert ds.md5(out_gpu) == ds.md5(out_ref)
    if not single:
        short = str(tmp_path / "short.tsv")
        with open(pre + ".whitelist.txt") as f, open(short, "w") as g:
            for i, ln in enumerate(f):
                if i < 10:
                    g.write("C%d\t%s\n" % (i, ln.strip()))
        bad = [x if x != table else short for x in common]
        r = subprocess.run([REF] + bad + ["-o", out_ref, "-t", "32"], stderr=subprocess.PIPE)
        g = subprocess.run([CLI] + bad + ["-o", out_gpu], stderr=subprocess.PIPE)
        assert r.returncode != 0 and g.returncode != 0
        assert b"Barcode does not exist in the translation table." in r.stderr and b"Barcode does not exist in the translation table." in g.stderr
