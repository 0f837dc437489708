"""chromap-amd on a box without a GPU: argument handling works, and a mapping run refuses to start -- there is
no CPU path behind the CLI either."""
import os
import subprocess

import pytest

import datasets

CLI = os.path.join(datasets.ROOT, "chromap_amd", "chromap-amd")
pytestmark = pytest.mark.skipif(not os.path.exists(CLI), reason="chromap-amd not built (python -c 'import __graft_entry__ as g; g.build()')")


def _run(*args):
    return subprocess.run([CLI] + list(args), stdout=subprocess.PIPE, stderr=subprocess.PIPE)


def test_version_and_help():
    r = _run("-v")
    assert r.returncode == 0 and b"chromap-amd" in r.stdout
    r = _run("-h")
    assert r.returncode == 0 and b"Usage" in r.stdout


def test_unknown_option_and_mismatched_inputs_are_errors():
    r = _run("--PAF")
    assert r.returncode != 0 and b"unsupported option" in r.stderr
    fa, _, _ = datasets.case_inputs("toy_atac")
    idx = datasets.case_index("toy_atac")
    r = _run("-x", idx, "-r", fa, "-1", "a.fq,b.fq", "-2", "c.fq", "-o", "o")
    assert r.returncode != 0 and b"don't match" in r.stderr, r.stderr
    r = _run("-x", idx, "-r", fa, "-1", "a.fq,b.fq", "-2", "c.fq,d.fq", "-b", "e.fq", "-o", "o")
    assert r.returncode != 0 and b"don't match" in r.stderr, r.stderr


def test_mapping_refuses_to_run_without_a_gpu(tmp_path):
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("a GPU is present")
    except ImportError:
        pass
    case = "toy_atac"
    fa, r1, r2 = datasets.case_inputs(case)
    out = str(tmp_path / "o.bed")
    r = _run("--preset", "atac", "-x", datasets.case_index(case), "-r", fa, "-1", r1, "-2", r2, "-o", out)
    assert r.returncode != 0
    assert not os.path.exists(out) or os.path.getsize(out) == 0
    assert b"HIP" in r.stderr or b"device" in r.stderr or b"GPU" in r.stderr, r.stderr[-500:]


def test_flag_combinations_without_a_record_type_are_refused():
    """argument checks happen before any device work: the refusals are testable without a GPU.  (--pairs on the ordinary pairing,
    pairs with cell barcodes and -n up to 8192 were refused in rounds 1-5; they are accepted now: tests/test_gpu_cli_golden.py)"""
    r = _run("--gpus", "0", "-x", "x", "-r", "y", "-1", "a", "-o", "o")
    assert r.returncode != 0 and b"--gpus" in r.stderr
    # -n: BED / TagAlign / pairs carry up to n records per read; one SAM slot per read on the device
    r = _run("--SAM", "-n", "2", "-x", "x", "-r", "y", "-1", "a", "-o", "o")
    assert r.returncode != 0 and b"--SAM with -n > 1" in r.stderr
    r = _run("-n", "8193", "-x", "x", "-r", "y", "-1", "a", "-o", "o")
    assert r.returncode != 0 and b"-n above 8192" in r.stderr
    r = _run("-n", "0", "-x", "x", "-r", "y", "-1", "a", "-o", "o")
    assert r.returncode != 0 and b"at least 1" in r.stderr
    # what used to be refused gets past the argument checks (and then fails on the missing files, not on the flags)
    for flags in (["--pairs"], ["--preset", "hic", "-b", "c"], ["-n", "100"]):
        r = _run(*flags, "-x", "x", "-r", "y", "-1", "a", "-2", "b", "-o", "o")
        assert r.returncode != 0 and b"outside this build" not in r.stderr, (flags, r.stderr)


def test_ingest_reader_inflates_bgzf_gzip_and_plain_text(tmp_path):
    """the CLI's chunk reader on its own (--inflate-only): BGZF is recognised and inflated block-parallel, ordinary gzip and
    plain text go through gzread; the bytes are the file's; a truncated BGZF file is an error, not a short read"""
    import gzip
    import sys
    sys.path.insert(0, os.path.join(datasets.ROOT, "tools"))
    import bgzf
    import numpy as np
    rng = np.random.default_rng(5)
    lines = []
    for i in range(120000):
        s = bytes(rng.choice(list(b"ACGT"), 50).astype(np.uint8))
        lines.append(b"@r%d\n%s\n+\n%s\n" % (i, s, b"I" * 50))
    text = b"".join(lines)
    plain = str(tmp_path / "r.fq")
    open(plain, "wb").write(text)
    bg = str(tmp_path / "r.bgzf.gz")
    bgzf.compress_file(plain, bg)
    assert gzip.open(bg, "rb").read() == text  # the writer produces valid multi-member gzip
    gz = str(tmp_path / "r.gz")
    with gzip.open(gz, "wb", compresslevel=1) as f:
        f.write(text)
    for path, kind in ((bg, b"bgzf"), (gz, b"gzread"), (plain, b"gzread")):
        r = _run("--inflate-only", path)
        assert r.returncode == 0 and r.stdout == text and r.stderr.strip() == kind, (path, r.stderr[-200:])
    cut = str(tmp_path / "cut.gz")
    open(cut, "wb").write(open(bg, "rb").read()[:-5000])
    r = _run("--inflate-only", cut)
    assert r.returncode != 0 and b"corrupted" in r.stderr


def test_ordinary_gzip_inflated_by_several_threads(tmp_path):
    """cm_pargz.h (the CLI's reader for single-stream gzip of 16 MiB and more): chunks searched for block starts and decoded
    speculatively with made-up windows, accepted only where they start exactly at the end of the decode before them.  The
    bytes must be gzread's for FASTQ text at several compression levels (guesses accepted), for concatenated members, for
    stored blocks and binary data (no guess counts: the serial path), and a truncated or damaged file must be an error"""
    import gzip
    import numpy as np
    rng = np.random.default_rng(11)
    n = 150000  # (20 MB of text: CM_PARGZ_MIN_KB lets files of a few MB through and 512 KiB chunks keep several groups -- fresh memory is dear on the CPU box)
    seq = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, (n, 60))]
    qual = (rng.integers(2, 41, (n, 60)) + 33).astype(np.uint8)
    rec = np.empty((n, 8 + 1 + 60 + 3 + 60 + 1), np.uint8)
    rec[:, :8] = np.frombuffer(b"".join(b"@r%06d" % i for i in range(n)), np.uint8).reshape(n, 8)
    rec[:, 8] = 10
    rec[:, 9:69] = seq
    rec[:, 69:72] = np.frombuffer(b"\n+\n", np.uint8)
    rec[:, 72:132] = qual
    rec[:, 132] = 10
    text = rec.tobytes()
    env = dict(os.environ, CM_PARGZ_THREADS="4", CM_PARGZ_MIN_KB="2048", CM_PARGZ_CHUNK_KB="512")

    def run(path):
        return subprocess.run([CLI, "--inflate-only", path], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)

    def stats(r):
        f = r.stderr.decode().strip().splitlines()[-1].split()
        assert f[0] == "pargz", r.stderr[-300:]
        return {k: int(v) for k, v in (x.split("=") for x in f[1:])}

    files = {}
    for lvl in (1, 6):
        p = str(tmp_path / ("t%d.gz" % lvl))
        with gzip.open(p, "wb", compresslevel=lvl) as g:
            g.write(text)
        assert os.path.getsize(p) > (4 << 20)
        files[lvl] = p
        r = run(p)
        assert r.returncode == 0 and r.stdout == text
        st = stats(r)
        assert st["accepted"] >= st["chunks"] - 1 and st["accepted"] > 3 and st["serial"] <= 1, st  # the guesses counted
    # the same file through gzread (CM_PARGZ=0) -- and a small file never takes the parallel reader
    r = subprocess.run([CLI, "--inflate-only", files[6]], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, CM_PARGZ="0"))
    assert r.returncode == 0 and r.stdout == text and r.stderr.strip() == b"gzread"
    # members one behind the other: deflate, stored (level 0: no dynamic block to find), deflate
    stored = str(tmp_path / "l0.gz")
    with gzip.open(stored, "wb", compresslevel=0) as g:
        g.write(text[:6_000_000])
    cat = str(tmp_path / "cat.gz")
    with open(cat, "wb") as f:
        for p in (files[1], stored, files[6]):
            f.write(open(p, "rb").read())
    r = run(cat)
    assert r.returncode == 0 and r.stdout == text + text[:6_000_000] + text
    assert stats(r)["accepted"] > 3
    # binary data (bytes of all values: the third decode that tells a literal >= 128 from a window byte), partly incompressible
    blob = bytearray()
    base = rng.integers(0, 256, 1 << 20, dtype=np.uint8).tobytes()
    words = [rng.integers(128, 256, int(rng.integers(3, 40)), dtype=np.uint8).tobytes() for _ in range(200)]
    while len(blob) < 20_000_000:
        blob += base[:int(rng.integers(1000, 200000))]
        for _ in range(20000):
            blob += words[int(rng.integers(0, 200))]
    blob = bytes(blob)
    bz = str(tmp_path / "bin.gz")
    with gzip.open(bz, "wb", compresslevel=6) as g:
        g.write(blob)
    if os.path.getsize(bz) > (4 << 20):
        r = run(bz)
        assert r.returncode == 0 and r.stdout == blob
    # damage: a truncated file, a flipped bit in the middle, a wrong CRC in the trailer
    z = open(files[6], "rb").read()
    for name, data in (("trunc.gz", z[:-200000]), ("flip.gz", z[:len(z) // 2] + bytes([z[len(z) // 2] ^ 0x10]) + z[len(z) // 2 + 1:]),
                       ("crc.gz", z[:-6] + bytes([z[-6] ^ 1]) + z[-5:])):
        p = str(tmp_path / name)
        open(p, "wb").write(data)
        r = run(p)
        assert r.returncode != 0 and b"corrupted" in r.stderr, name


def test_pipelined_gunzip_over_members_threads_and_chunk_sizes(tmp_path):
    """the round-6 form of cm_pargz.h -- the next group decoded while the last is finished, teams that live with the file, output split
    between the caller's buffer and the spill -- on files made of members of many sizes (a member's end inside a group, at a group's
    edge, members smaller than a chunk, a stored member, an empty member), for several team and chunk sizes, with and without decoding
    ahead: the bytes are Python's gzip.decompress of the same file every time"""
    import gzip
    import hashlib
    import zlib
    import numpy as np
    rng = np.random.default_rng(2026)

    def fastq_like(n_rec):
        names = rng.integers(0, 8, n_rec)
        L = 44
        seq = np.frombuffer(b"ACGTN", np.uint8)[np.minimum(rng.integers(0, 41, (n_rec, L)) // 10, 4)]
        qual = (rng.integers(2, 41, (n_rec, L)) + 33).astype(np.uint8)
        out = bytearray()
        for i in range(n_rec):
            out += b"@lane%d:%d\n" % (names[i], i)
            out += seq[i].tobytes() + b"\n+\n" + qual[i].tobytes() + b"\n"
        return bytes(out)

    base = fastq_like(30000)  # ~3.3 MB of text, cut and repeated below (CM_PARGZ_MIN_KB lets files this small through: fresh memory is dear on the CPU box)
    members = []
    sizes = [len(base), 1, 70_000, 1_000_000, 0, len(base) // 2, 300_000, len(base), 12_345, len(base)]
    levels = [6, 6, 1, 9, 6, 0, 6, 1, 6, 4]
    want = bytearray()
    for sz, lvl in zip(sizes, levels):
        off = int(rng.integers(0, len(base) - sz + 1)) if sz < len(base) else 0
        piece = base[off:off + sz]
        want += piece
        co = zlib.compressobj(lvl, zlib.DEFLATED, 31)
        members.append(co.compress(piece) + co.flush())
    path = str(tmp_path / "members.gz")
    with open(path, "wb") as f:
        for m in members:
            f.write(m)
    assert os.path.getsize(path) > (4 << 20)
    data = open(path, "rb").read()
    assert gzip.decompress(data) == bytes(want)
    md5 = hashlib.md5(bytes(want)).hexdigest()
    for threads, chunk_kb, ahead in ((2, 512, True), (3, 256, True), (4, 64, True), (5, 128, False), (8, 128, True), (2, 64, False)):
        env = dict(os.environ, CM_PARGZ_THREADS=str(threads), CM_PARGZ_CHUNK_KB=str(chunk_kb), CM_PARGZ_MIN_KB="1024")
        if not ahead:
            env["CM_PARGZ_NO_AHEAD"] = "1"
        r = subprocess.run([CLI, "--inflate-only", path], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        assert r.returncode == 0, (threads, chunk_kb, ahead, r.stderr[-300:])
        assert hashlib.md5(r.stdout).hexdigest() == md5, (threads, chunk_kb, ahead)
        assert r.stderr.strip().splitlines()[-1].startswith(b"pargz"), r.stderr[-200:]
    # a flipped bit deep inside is an error for every configuration (CRC or invalid data), never different bytes
    bad = bytearray(data)
    bad[len(bad) * 2 // 3] ^= 0x10
    badp = str(tmp_path / "bad.gz")
    open(badp, "wb").write(bad)
    for threads, chunk_kb in ((3, 256), (8, 64)):
        env = dict(os.environ, CM_PARGZ_THREADS=str(threads), CM_PARGZ_CHUNK_KB=str(chunk_kb), CM_PARGZ_MIN_KB="1024")
        r = subprocess.run([CLI, "--inflate-only", badp], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        assert r.returncode != 0 and b"corrupted" in r.stderr, r.stderr[-300:]
