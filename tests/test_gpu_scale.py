"""BASELINE.json's configurations at multi-million-pair scale against the REFERENCE BINARY on the GPU box (slow tests):
tools/ref_baseline.py builds a 400-Mbp / 24-sequence synthetic genome and its index on the device, writes them in the
reference's file formats, writes >= 4 M device-generated pairs as FASTQ, runs oracle/_ref/chromap (all host threads) and
chromap-amd on the same files and compares the output byte for byte.

  chip     --preset chip, 2 x 50                       BED
  atac     --preset atac, 2 x 50 with adapter read-through, + the same reads as BGZF through the CLI's block-parallel inflate
  scATAC   --preset atac + 16-base cell barcodes against a 737 280-entry whitelist (10 % of the barcodes one substitution off)
  hic      --preset hic, 2 x 150 Hi-C shaped pairs, 35 % with a ligation junction inside a read, 0.1 % indels: pairs file
  atac on a genome with planted repeat families (32 x 600 copies of 3 kb at 2 %) and on the mosaic genome (profile 2: 47 % of the
  bases repeat-derived): the long-list paths of every stage

oracle/_ref/chromap travels with the repository like the built libraries (tests are skipped where it is absent)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.slow]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "chromap")

CONFIGS = {
    "chip": ["--preset", "chip", "--pairs", "2000000", "--batches", "2"],
    "atac_bgzf": ["--preset", "atac", "--pairs", "2000000", "--batches", "2", "--bgzf-check"],
    "scatac_737k": ["--preset", "atac", "--pairs", "2000000", "--batches", "2", "--barcodes", "737280"],
    # repeat-bearing genomes (round 4): every cooperative size class, the wave rescue searches and their pool at scale
    "atac_planted_repeats": ["--preset", "atac", "--pairs", "2000000", "--batches", "2", "--repeats", "32,600,3000,0.02"],
    "atac_mosaic_genome": ["--preset", "atac", "--pairs", "2000000", "--batches", "2", "--repeats", "profile:2"],
    "hic_chimeric": ["--preset", "hic", "--pairs", "2000000", "--batches", "2", "--readlen", "150", "--hic", "0.35", "--indel-rate", "0.001"],
}


@pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref/chromap not built")
@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_four_million_pairs_equal_reference_binary(name, tmp_path):
    cmd = [sys.executable, os.path.join(ROOT, "tools", "ref_baseline.py"), "--genome", "400000000", "--nseq", "24", "--dir", str(tmp_path / "w"),
           "--seed0", "9000"] + CONFIGS[name]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-2000:]
    r = json.loads(p.stdout.decode().strip().splitlines()[-1])
    assert "error" not in r, r
    assert "error" not in r["reference"], r["reference"]
    assert "error" not in r["chromap_amd"], r["chromap_amd"]
    assert r["pairs"] >= 4_000_000
    assert r["reference"]["bed_lines"] > 0.8 * r["pairs"] * (1 if name != "scatac_737k" else 0.8)
    assert r["bed_identical_to_reference"] is True, (r["reference"]["bed_md5"], r["chromap_amd"]["bed_md5"])
    if name == "atac_bgzf":
        assert r["bgzf"]["identical"] is True, r["bgzf"]
