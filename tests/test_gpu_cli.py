"""The command-line host (chromap_amd/chromap-amd, cm_cli.cpp) run exactly like the reference
was run for the golden outputs (tests/golden/make_golden.py): `-i` index build on the device,
then mapping with the golden case's flags; output bytes and stderr counters must equal the
reference's.  Where the built reference binary travelled to the box (oracle/_ref/chromap),
the device-built index file is also loaded by the *reference* to show it is a valid index."""
import hashlib
import os
import re
import subprocess

import pytest

import datasets as ds

pytestmark = pytest.mark.gpu
CLI = os.path.join(ds.ROOT, "chromap_amd", "chromap-amd")
REF = os.path.join(ds.ROOT, "oracle", "_ref", "chromap")
CASES = ["toy_atac", "s1_atac", "s2_atac_q0", "s3_chip", "h1_hic", "b1_atac_bc", "b2_atac_bc2_q0", "b3_bulk_level_bc_q0", "s1_se_chip", "s4_se_atac_q0",
         "s4_inmem_q0", "s4_se_inmem_q0", "b1_inmem_bc", "s1_inmem_nodedup", "s1_chip_sam", "s3_sam_q0", "s2_atac_sam", "s1_se_sam", "s3_chip_chrorder", "h1_hic_chrorder_q0", "h2_hic_natural_q0",
         "s4_atac_n3_q0", "s4_se_n2_q0", "h2_hic_n2_q0", "h2_hic_sam_q0", "h1_hic_sam"]


def _reads(name):
    fa, r1, r2 = ds.case_inputs(name)
    m = ds.single_end_mate(name)
    reads = ["-1", r1, "-2", r2] if not m else ["-1", r1 if m == 1 else r2]
    if ds.has_barcodes(name):
        bc, wl = ds.case_barcode_inputs(name)
        reads += ["-b", bc, "--barcode-whitelist", wl]
    return fa, reads


@pytest.fixture(scope="module")
def built(tmp_path_factory):
    assert os.path.exists(CLI), "chromap-amd not built (make -C chromap_amd/csrc)"
    cache = {}

    def get(name):
        fa, _ = _reads(name)
        if fa not in cache:
            idx = str(tmp_path_factory.mktemp("idx") / "d.idx")
            subprocess.run([CLI, "-i", "-r", fa, "-o", idx], check=True, stderr=subprocess.PIPE)
            cache[fa] = idx
        return cache[fa]
    return get


@pytest.mark.parametrize("name", CASES)
def test_cli_matches_reference_output(name, built, tmp_path):
    meta = ds.case_meta(name)
    fa, reads = _reads(name)
    out = str(tmp_path / "out.txt")
    flags = list(meta["chromap_flags"])
    for of in ("--chr-order", "--pairs-natural-chr-order"):  # the golden metadata keeps the order as a comma list; the programs read a file
        if of in flags:
            k = flags.index(of)
            order = str(tmp_path / (of.strip("-") + ".txt"))
            with open(order, "w") as f:
                f.write("\n".join(flags[k + 1].split(",")) + "\n")
            flags[k + 1] = order
    r = subprocess.run([CLI] + flags + ["-x", built(name), "-r", fa] + reads + ["-o", out, "-t", "1"],
                       stderr=subprocess.PIPE, check=True)
    with open(out, "rb") as f:
        got = f.read()
    assert hashlib.md5(got).hexdigest() == meta["bed_md5"]
    assert got == ds.case_golden_bed(name)
    log = r.stderr.decode()
    for key, pat in (("num_reads", r"Number of reads: (\d+)"), ("num_mapped_reads", r"Number of mapped reads: (\d+)"),
                     ("num_uniquely_mapped_reads", r"Number of uniquely mapped reads: (\d+)"),
                     ("num_candidates", r"Number of candidates: (\d+)"), ("num_mappings", r"Number of mappings: (\d+)"),
                     ("num_output", r"Number of output mappings \(passed filters\): (\d+)")):
        want = meta["reference_stderr_counters"].get(key)
        if want is not None:
            assert int(re.search(pat, log).group(1)) == want, key


@pytest.mark.parametrize("extra", [["--ingest-chunk-mb", "1"], ["--host-ingest"]])
def test_cli_ingest_variants(extra, built, tmp_path):
    """growing-chunk path of the device FASTQ parser, and the kseq-style host parser"""
    for name in ("s2_atac_q0", "b1_atac_bc"):
        meta = ds.case_meta(name)
        fa, reads = _reads(name)
        out = str(tmp_path / (name + ".txt"))
        subprocess.run([CLI] + meta["chromap_flags"] + extra + ["-x", built(name), "-r", fa] + reads + ["-o", out], check=True,
                       stderr=subprocess.PIPE)
        assert ds.md5(out) == meta["bed_md5"]


@pytest.mark.parametrize("name", ["s1_atac", "s3_chip", "b1_atac_bc", "s4_se_atac_q0", "s2_tagalign_q0"])
def test_cli_multi_gpu_path_with_one_gpu(name, built, tmp_path):
    """--gpus N code path (one context + host thread per GPU, RCCL record exchange to the chromosome owners issued by the
    library, every owner formats its section) with N = 1: same bytes and counters"""
    meta = ds.case_meta(name)
    fa, reads = _reads(name)
    out = str(tmp_path / "out.txt")
    r = subprocess.run([CLI] + list(meta["chromap_flags"]) + ["--force-exchange", "-x", built(name), "-r", fa] + reads + ["-o", out],
                       stderr=subprocess.PIPE, check=True)
    assert ds.md5(out) == meta["bed_md5"]
    log = r.stderr.decode()
    assert int(re.search(r"Number of mapped reads: (\d+)", log).group(1)) == meta["reference_stderr_counters"]["num_mapped_reads"]


def _n_gpus():
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:
        return 0


@pytest.mark.skipif(_n_gpus() < 2, reason="needs two GPUs (RCCL between real ranks; the 1-GPU box runs the same path with --force-exchange)")
@pytest.mark.parametrize("name", ["s1_atac", "s3_chip", "b1_atac_bc", "s5_atac_24chr_q0"])
def test_cli_two_gpus_equal_golden(name, built, tmp_path):
    """chromap-amd --gpus 2: two contexts, two host threads, the records exchanged by owner chromosome over RCCL between two
    real ranks -- the output must be the golden file's"""
    meta = ds.case_meta(name)
    fa, reads = _reads(name)
    out = str(tmp_path / "out.txt")
    subprocess.run([CLI] + list(meta["chromap_flags"]) + ["--gpus", "2", "-x", built(name), "-r", fa] + reads + ["-o", out],
                   stderr=subprocess.PIPE, check=True, timeout=600)
    assert ds.md5(out) == meta["bed_md5"]


def test_cli_refuses_what_it_cannot_write(built, tmp_path):
    """what is still outside this build ends with a message, not with garbage: --SAM with several best
    mappings, -n beyond the record slots a batch can address, pairs of a single-end run (the reference's own message), pairs
    records over several GPUs.  (--pairs on the ordinary pairing, pairs with cell barcodes and -n up to 8192 were refused in
    rounds 1-5: tests/test_gpu_cli_golden.py now holds the reference's output for them.)"""
    name = "s1_atac"
    fa, reads = _reads(name)
    out = str(tmp_path / "o")
    r = subprocess.run([CLI, "--SAM", "-n", "3", "-x", built(name), "-r", fa] + reads + ["-o", out], stderr=subprocess.PIPE)
    assert r.returncode != 0 and b"outside this build" in r.stderr and not os.path.exists(out)
    r = subprocess.run([CLI, "-n", "9000", "-x", built(name), "-r", fa] + reads + ["-o", out], stderr=subprocess.PIPE)
    assert r.returncode != 0 and b"outside this build" in r.stderr and not os.path.exists(out)
    r = subprocess.run([CLI, "--pairs", "-x", built(name), "-r", fa, "-1", reads[1], "-o", out], stderr=subprocess.PIPE)
    assert r.returncode != 0 and b"No support for single-end HiC yet!" in r.stderr
    r = subprocess.run([CLI, "--preset", "hic", "--gpus", "2", "-x", built(name), "-r", fa] + reads + ["-o", out], stderr=subprocess.PIPE)
    assert r.returncode != 0


def test_cli_reads_gzip(built, tmp_path):
    import gzip
    import shutil
    name = "s1_atac"
    meta = ds.case_meta(name)
    fa, r1, r2 = ds.case_inputs(name)
    gz = []
    for i, f in enumerate((r1, r2)):
        g = str(tmp_path / ("r%d.fq.gz" % i))
        with open(f, "rb") as src, gzip.open(g, "wb", compresslevel=1) as dst:
            shutil.copyfileobj(src, dst)
        gz.append(g)
    out = str(tmp_path / "o.bed")
    subprocess.run([CLI] + meta["chromap_flags"] + ["-x", built(name), "-r", fa, "-1", gz[0], "-2", gz[1], "-o", out], check=True,
                   stderr=subprocess.PIPE)
    assert ds.md5(out) == meta["bed_md5"]


def test_cli_reads_bgzf_inflated_on_the_device(built, tmp_path):
    """block-compressed input stays compressed on the host: the blocks are inflated on the device (one GPU); the records equal the
    plain-text run; a damaged block ends the run with the reference's corrupted-file message"""
    import sys
    sys.path.insert(0, os.path.join(ds.ROOT, "tools"))
    import bgzf
    name = "s1_atac"
    meta = ds.case_meta(name)
    fa, r1, r2 = ds.case_inputs(name)
    bg = []
    for i, f in enumerate((r1, r2)):
        g = str(tmp_path / ("r%d.fq.gz" % i))
        bgzf.compress_file(f, g, level=6 if i else 1)
        bg.append(g)
    out = str(tmp_path / "o.bed")
    r = subprocess.run([CLI] + meta["chromap_flags"] + ["-x", built(name), "-r", fa, "-1", bg[0], "-2", bg[1], "-o", out], stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr[-500:]
    assert ds.md5(out) == meta["bed_md5"]
    # half a record behind the last whole one: an error, as it is for text held on the host
    import bgzf as _b
    tail = str(tmp_path / "tail.fq.gz")
    with open(tail, "wb") as f:
        f.write(open(bg[1], "rb").read()[:-28])  # (without the end-of-file marker block)
        f.write(_b._block(b"@half\nACGT", 6))
        f.write(_b._block(b"", 6))
    r = subprocess.run([CLI] + meta["chromap_flags"] + ["-x", built(name), "-r", fa, "-1", bg[0], "-2", tail, "-o", out], stderr=subprocess.PIPE)
    assert r.returncode != 0 and b"corrupted" in r.stderr, r.stderr[-500:]
    raw = bytearray(open(bg[1], "rb").read())
    raw[len(raw) // 2] ^= 0x40
    bad = str(tmp_path / "bad.fq.gz")
    open(bad, "wb").write(raw)
    r = subprocess.run([CLI] + meta["chromap_flags"] + ["-x", built(name), "-r", fa, "-1", bg[0], "-2", bad, "-o", out], stderr=subprocess.PIPE)
    assert r.returncode != 0 and b"corrupted" in r.stderr, r.stderr[-500:]


@pytest.mark.parametrize("name", ["b1_atac_bc", "s4_se_atac_q0", "s2_atac_q0"])
def test_cli_bgzf_every_stream_kind(name, built, tmp_path):
    """reads, mates and barcodes each block-compressed (their own levels and block sizes, small pieces so that several calls per
    file hand blocks to the device): single-cell, single-end and variable-length inputs give the golden file"""
    import sys
    sys.path.insert(0, os.path.join(ds.ROOT, "tools"))
    import bgzf
    meta = ds.case_meta(name)
    fa, reads = _reads(name)
    args = []
    k = 0
    for a in reads:
        if os.path.isfile(a) and not a.endswith(".txt") and args and args[-1] in ("-1", "-2", "-b"):
            g = str(tmp_path / ("in%d.fq.gz" % k))
            text = open(a, "rb").read()
            block = (0xff00, 7001, 30000)[k % 3]
            with open(g, "wb") as f:
                for i in range(0, len(text), block):
                    f.write(bgzf._block(text[i:i + block], (1, 6, 9)[k % 3]))
                f.write(bgzf._block(b"", 1))
            a = g
            k += 1
        args.append(a)
    assert k >= 1
    out = str(tmp_path / "o.bed")
    r = subprocess.run([CLI] + list(meta["chromap_flags"]) + ["-x", built(name), "-r", fa] + args + ["-o", out, "--ingest-chunk-mb", "1"], stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr[-500:]
    assert ds.md5(out) == meta["bed_md5"]


@pytest.mark.skipif(not os.path.exists(REF), reason="built reference binary not present")
def test_device_built_index_loads_in_reference(built, tmp_path):
    name = "s1_atac"
    meta = ds.case_meta(name)
    fa, reads = _reads(name)
    out = str(tmp_path / "ref_out.bed")
    subprocess.run([REF] + meta["chromap_flags"] + ["-x", built(name), "-r", fa] + reads + ["-o", out, "-t", "2"],
                   check=True, stderr=subprocess.PIPE)
    assert ds.md5(out) == meta["bed_md5"]


def test_cli_rejects_unknown_option(tmp_path):
    r = subprocess.run([CLI, "--PAF", "-x", "x", "-r", os.path.join(ds.GOLD, "toy", "ref.fa"), "-1", "a", "-o",
                        str(tmp_path / "o")], stderr=subprocess.PIPE)
    assert r.returncode != 0 and b"unsupported option" in r.stderr


def test_cli_reads_large_gzip_with_several_inflating_threads(tmp_path):
    """ordinary gzip files of 16 MiB and more go through cm_pargz.h (speculative chunks, several threads): the BED must be the
    one the plain-text files give, and gzread's (CM_PARGZ=0)"""
    import sys
    pre = str(tmp_path / "d")
    subprocess.check_call([sys.executable, os.path.join(ds.ROOT, "tools", "gen_real.py"), "--out", pre, "--genome", "20000000", "--chroms", "4",
                           "--pairs", "200000", "--seed", "77"])
    for tag in ("_1", "_2"):
        subprocess.check_call(["gzip", "-1", "-k", pre + tag + ".fq"])
        assert os.path.getsize(pre + tag + ".fq.gz") > (16 << 20)
    idx = pre + ".idx"
    subprocess.run([CLI, "-i", "-r", pre + ".fa", "-o", idx], check=True, stderr=subprocess.PIPE)
    outs = {}
    for name, sfx, env in (("plain", ".fq", {}), ("pargz", ".fq.gz", {"CM_PARGZ_THREADS": "8"}), ("gzread", ".fq.gz", {"CM_PARGZ": "0"})):
        out = str(tmp_path / (name + ".bed"))
        r = subprocess.run([CLI, "--preset", "atac", "-x", idx, "-r", pre + ".fa", "-1", pre + "_1" + sfx, "-2", pre + "_2" + sfx, "-o", out],
                           stderr=subprocess.PIPE, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr.decode()[-1500:]
        outs[name] = ds.md5(out)
    assert os.path.getsize(out) > 1000000
    assert outs["pargz"] == outs["plain"] and outs["gzread"] == outs["plain"], outs


def test_cli_overlapped_batches_equal_the_serial_order(tmp_path):
    """round 6: the CLI reads, inflates and takes batch i + 1 while a worker maps batch i (staging buffers, cm_ingest.hip), warms the
    runtime up with a dry run, sizes the record store after the first piece and reads a half-size first piece.  Seven batches of
    500 000 pairs from plain text and from BGZF (blocks inflated on the device): the BED equals the serial order's
    (CM_CLI_NO_OVERLAP=1, no dry run, whole first piece) byte for byte, with and without -n 3 sampling"""
    import sys
    pre = str(tmp_path / "d")
    subprocess.check_call([sys.executable, os.path.join(ds.ROOT, "tools", "gen_real.py"), "--out", pre, "--genome", "30000000", "--chroms", "5",
                           "--pairs", "3300000", "--seed", "606", "--len-min", "50", "--len-max", "50"])
    sys.path.insert(0, os.path.join(ds.ROOT, "tools"))
    import bgzf
    for tag in ("_1", "_2"):
        bgzf.compress_file(pre + tag + ".fq", pre + tag + ".fq.bgz")
    idx = pre + ".idx"
    subprocess.run([CLI, "-i", "-r", pre + ".fa", "-o", idx], check=True, stderr=subprocess.PIPE)
    serial = {"CM_CLI_NO_OVERLAP": "1", "CM_NO_DRY_RUN": "1", "CM_FIRST_PIECE_DIV": "1", "CM_FQ_EARLY": "0"}
    for extra in ([], ["-n", "3", "-q", "0"]):
        outs = {}
        for name, sfx, env in (("serial_text", ".fq", serial), ("text", ".fq", {}), ("serial_bgzf", ".fq.bgz", serial), ("bgzf", ".fq.bgz", {})):
            out = str(tmp_path / (name + ".bed"))
            r = subprocess.run([CLI, "--preset", "atac", "-x", idx, "-r", pre + ".fa", "-1", pre + "_1" + sfx, "-2", pre + "_2" + sfx, "-o", out,
                                "--batch-pairs", "500000"] + extra, stderr=subprocess.PIPE, env=dict(os.environ, **env))
            assert r.returncode == 0, r.stderr.decode()[-1500:]
            assert r.stderr.count(b"Mapped 500000 read pairs.") >= 6, r.stderr.decode()[-800:]
            outs[name] = ds.md5(out)
        assert os.path.getsize(out) > 50_000_000
        assert len(set(outs.values())) == 1, (extra, outs)
