"""chromap-amd against golden outputs the REFERENCE made (tests/golden/cli_only/make_cli_golden.py) for the option combinations
rounds 1-5 refused: --pairs on the ordinary pairing, pairs with cell barcodes, duplicate removal on pairs, --SAM with
--barcode-translate, -n above 64.  The command line is the one the reference was run with; the bytes must be equal."""
import gzip
import glob
import hashlib
import json
import os
import subprocess
import sys

import pytest

import datasets as ds

pytestmark = pytest.mark.gpu
HERE = os.path.join(ds.ROOT, "tests", "golden", "cli_only")
CLI = os.path.join(ds.ROOT, "chromap_amd", "chromap-amd")
CASES = sorted(os.path.basename(p)[:-5] for p in glob.glob(os.path.join(HERE, "*.json")))


def _inputs(meta):
    key = hashlib.md5(" ".join(meta["generator_args"]).encode()).hexdigest()[:12]
    d = os.path.join(ds.CACHE, "cli_" + key)
    pre = os.path.join(d, "d")
    if not os.path.exists(pre + "_2.fq"):
        os.makedirs(d, exist_ok=True)
        subprocess.check_call([sys.executable, os.path.join(ds.ROOT, "tools", "gen_synth.py"), "--out", pre] + meta["generator_args"])
    got = {"fa": ds.md5(pre + ".fa"), "r1": ds.md5(pre + "_1.fq"), "r2": ds.md5(pre + "_2.fq")}
    if "bc" in meta["input_md5"]:
        got.update({"bc": ds.md5(pre + "_bc.fq"), "whitelist": ds.md5(pre + ".whitelist.txt")})
    assert got == meta["input_md5"], "regenerated inputs differ from the ones the golden output was made from"
    return pre


def test_there_are_cases():
    assert len(CASES) >= 8


@pytest.mark.parametrize("name", CASES)
def test_cli_equals_reference_made_golden(name, tmp_path):
    with open(os.path.join(HERE, name + ".json")) as f:
        meta = json.load(f)
    pre = _inputs(meta)
    idx = pre + ".amd.idx"
    if not os.path.exists(idx):
        subprocess.run([CLI, "-i", "-r", pre + ".fa", "-o", idx], check=True, stderr=subprocess.PIPE)
    reads = ["-1", pre + "_1.fq", "-2", pre + "_2.fq"]
    if "bc" in meta["input_md5"]:
        reads += ["-b", pre + "_bc.fq", "--barcode-whitelist", pre + ".whitelist.txt"]
    flags = list(meta["chromap_flags"])
    if "@TABLE@" in flags:
        table = str(tmp_path / "table.tsv")
        with open(pre + ".whitelist.txt") as f, open(table, "w") as g:
            for i, ln in enumerate(f):
                g.write("CELL%05d\t%s\n" % (i, ln.strip()))
        flags[flags.index("@TABLE@")] = table
    out = str(tmp_path / "out.txt")
    r = subprocess.run([CLI] + flags + ["-x", idx, "-r", pre + ".fa"] + reads + ["-o", out], stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    got = open(out, "rb").read()
    with gzip.open(os.path.join(HERE, name + ".out.gz"), "rb") as f:
        want = f.read()
    assert len(want) == meta["output_bytes"] and hashlib.md5(want).hexdigest() == meta["output_md5"]
    assert hashlib.md5(got).hexdigest() == meta["output_md5"]
    assert got == want
