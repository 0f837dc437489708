#!/usr/bin/env python3
"""Regenerates tests/golden/*.json + *.bed.gz by running the REFERENCE itself
(oracle/_ref/chromap, built unchanged from /root/reference by `make -C oracle ref`).

Only runs in the development container.  Inputs are synthesised by tools/gen_synth.py
(seeded) or are the data files of the reference's own test/ directory (copied to
tests/golden/toy/: ref.fa, read1.fq, read2.fq -- data, not code).  What is committed:
expected outputs (BED), the reference's own stderr counters, and md5s of the inputs so a
test can tell when its regenerated input is not the one the golden output belongs to.
"""
import gzip
import hashlib
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref", "chromap")
GEN = os.path.join(ROOT, "tools", "gen_synth.py")

# single-end cases: which mate file is mapped alone
SINGLE_END = {"s1_se_chip": 1, "s4_se_atac_q0": 2, "s4_se_inmem_q0": 1, "s1_se_sam": 1,
              "b1_se_bc": 1, "b3_se_bc_bulk_q0": 1, "b3_se_bc_inmem_q0": 2, "b1_se_bc_tagalign_q0": 1,
              "s4_se_tagalign_q0": 2, "s4_se_n2_q0": 2, "s4_se_drop2_q0": 2,
              "p1_se_e5_q0": 1, "p5_se_n5_e5_q0": 2}

# name -> (generator args or None for the toy data, chromap mapping flags)
CASES = {
    "toy_chip": (None, ["--preset", "chip"]),
    "toy_atac": (None, ["--preset", "atac"]),
    "s1_atac": (["--genome", "2000000", "--chroms", "4", "--pairs", "20000", "--readlen", "50", "--frag-min", "35"],
                ["--preset", "atac"]),
    "s1_chip_q0": (["--genome", "2000000", "--chroms", "4", "--pairs", "20000", "--readlen", "50", "--frag-min", "35"],
                   ["--preset", "chip", "-q", "0"]),
    "s2_atac_q0": (["--genome", "2000000", "--chroms", "3", "--pairs", "20000", "--readlen", "60", "--frag-min", "35",
                    "--varlen", "--seed", "7"], ["--preset", "atac", "-q", "0"]),
    "s3_chip": (["--genome", "6000000", "--chroms", "5", "--pairs", "30000", "--readlen", "100", "--seed", "99",
                 "--indel", "0.003", "--sub", "0.02"], ["--preset", "chip"]),
    "h1_hic": (["--genome", "3000000", "--chroms", "4", "--pairs", "20000", "--readlen", "150", "--frag-min", "300",
                "--frag-max", "800", "--hic", "--seed", "21", "--indel", "0.001"], ["--preset", "hic"]),
    "h2_hic_q0": (["--genome", "1000000", "--chroms", "3", "--pairs", "15000", "--readlen", "100", "--frag-min", "200",
                   "--frag-max", "500", "--hic", "--seed", "22", "--indel", "0.004", "--sub", "0.02"],
                  ["--preset", "hic", "-q", "0"]),
    "b1_atac_bc": (["--genome", "2000000", "--chroms", "4", "--pairs", "20000", "--readlen", "50", "--frag-min", "35",
                    "--barcodes", "500", "--seed", "31"], ["--preset", "atac"]),
    "b2_atac_bc2_q0": (["--genome", "1000000", "--chroms", "3", "--pairs", "15000", "--readlen", "50", "--frag-min", "35",
                        "--barcodes", "300", "--seed", "32"], ["--preset", "atac", "-q", "0", "--bc-error-threshold", "2"]),
    "s1_se_chip": (["--genome", "2000000", "--chroms", "4", "--pairs", "20000", "--readlen", "50", "--frag-min", "35"],
                   ["--preset", "chip"]),
    "s4_se_atac_q0": (["--genome", "300000", "--chroms", "2", "--pairs", "20000", "--readlen", "50", "--frag-min", "40",
                       "--seed", "5"], ["--preset", "atac", "-q", "0"]),
    # in-memory post-processing (no --low-mem): Tn5 shift before the sort, RemovePCRDuplicate keeps the last of a run
    "s4_inmem_q0": (["--genome", "300000", "--chroms", "2", "--pairs", "20000", "--readlen", "50", "--frag-min", "40",
                     "--seed", "5"], ["-l", "2000", "--remove-pcr-duplicates", "--Tn5-shift", "--trim-adapters", "-q", "0"]),
    "s4_se_inmem_q0": (["--genome", "300000", "--chroms", "2", "--pairs", "20000", "--readlen", "50", "--frag-min", "40",
                        "--seed", "5"], ["--remove-pcr-duplicates", "--Tn5-shift", "-q", "0"]),
    "b1_inmem_bc": (["--genome", "2000000", "--chroms", "4", "--pairs", "20000", "--readlen", "50", "--frag-min", "35",
                     "--barcodes", "500", "--seed", "31"], ["-l", "2000", "--remove-pcr-duplicates", "--Tn5-shift", "--trim-adapters"]),
    # single-cell, duplicate removal at bulk level (default without --preset atac)
    "b3_bulk_level_bc_q0": (["--genome", "300000", "--chroms", "2", "--pairs", "30000", "--readlen", "50", "--frag-min", "40",
                             "--barcodes", "40", "--seed", "33", "--dup-frac", "0.3"],
                            ["--preset", "atac", "--remove-pcr-duplicates-at-bulk-level", "-q", "0"]),
    "s1_inmem_nodedup": (["--genome", "2000000", "--chroms", "4", "--pairs", "20000", "--readlen", "50", "--frag-min", "35"],
                         ["-l", "2000", "--Tn5-shift"]),
    # --SAM: ksw alignment, CIGAR / NM / MD, SAMMapping sort + dedup
    "s1_chip_sam": (["--genome", "2000000", "--chroms", "4", "--pairs", "20000", "--readlen", "50", "--frag-min", "35"],
                    ["--preset", "chip", "--SAM"]),
    "s3_sam_q0": (["--genome", "6000000", "--chroms", "5", "--pairs", "30000", "--readlen", "100", "--seed", "99",
                   "--indel", "0.003", "--sub", "0.02"], ["-l", "2000", "--SAM", "-q", "0"]),
    "s2_atac_sam": (["--genome", "2000000", "--chroms", "3", "--pairs", "20000", "--readlen", "60", "--frag-min", "35",
                     "--varlen", "--seed", "7"], ["--preset", "atac", "--SAM"]),
    "s1_se_sam": (["--genome", "2000000", "--chroms", "4", "--pairs", "20000", "--readlen", "50", "--frag-min", "35"],
                  ["--SAM", "--remove-pcr-duplicates"]),
    # single-cell --SAM: barcode in the sort / duplicate keys, CB:Z tag
    "b1_bc_sam": (["--genome", "2000000", "--chroms", "4", "--pairs", "20000", "--readlen", "50", "--frag-min", "35",
                   "--barcodes", "500", "--seed", "31"], ["--preset", "atac", "--SAM"]),
    "b3_bc_sam_q0": (["--genome", "300000", "--chroms", "2", "--pairs", "30000", "--readlen", "50", "--frag-min", "40",
                      "--barcodes", "40", "--seed", "33", "--dup-frac", "0.3"], ["--preset", "atac", "--SAM", "-q", "0"]),
    # single-end reads with cell barcodes (MappingWithBarcode): cell-level and bulk-level low-memory merge, in-memory, TagAlign
    "b1_se_bc": (["--genome", "2000000", "--chroms", "4", "--pairs", "20000", "--readlen", "50", "--frag-min", "35",
                  "--barcodes", "500", "--seed", "31"], ["--preset", "atac"]),
    "b3_se_bc_bulk_q0": (["--genome", "300000", "--chroms", "2", "--pairs", "30000", "--readlen", "50", "--frag-min", "40",
                          "--barcodes", "40", "--seed", "33", "--dup-frac", "0.3"],
                         ["--preset", "atac", "--remove-pcr-duplicates-at-bulk-level", "-q", "0"]),
    "b3_se_bc_inmem_q0": (["--genome", "300000", "--chroms", "2", "--pairs", "30000", "--readlen", "50", "--frag-min", "40",
                           "--barcodes", "40", "--seed", "33", "--dup-frac", "0.3"], ["--remove-pcr-duplicates", "--Tn5-shift", "-q", "0"]),
    "b1_se_bc_tagalign_q0": (["--genome", "2000000", "--chroms", "4", "--pairs", "20000", "--readlen", "50", "--frag-min", "35",
                              "--barcodes", "500", "--seed", "31"], ["--TagAlign", "--remove-pcr-duplicates", "-q", "0"]),
    # --TagAlign: paired-end bulk (num_dups on the second line), paired-end single-cell (no barcode, no num_dups), single-end bulk
    "s2_tagalign_q0": (["--genome", "2000000", "--chroms", "3", "--pairs", "20000", "--readlen", "60", "--frag-min", "35",
                        "--varlen", "--seed", "7"], ["--preset", "atac", "--TagAlign", "-q", "0"]),
    "b2_tagalign_bc": (["--genome", "1000000", "--chroms", "3", "--pairs", "15000", "--readlen", "50", "--frag-min", "35",
                        "--barcodes", "300", "--seed", "32"], ["--preset", "atac", "--TagAlign"]),
    "s4_se_tagalign_q0": (["--genome", "300000", "--chroms", "2", "--pairs", "20000", "--readlen", "50", "--frag-min", "40",
                           "--seed", "5"], ["--preset", "chip", "--TagAlign", "-q", "0"]),
    # --chr-order: reference reordered, candidate rids re-ranked before verification (flag value: comma list,
    # written to a file for the reference; unlisted chromosomes follow in reference order)
    "s3_chip_chrorder": (["--genome", "6000000", "--chroms", "5", "--pairs", "30000", "--readlen", "100", "--seed", "99",
                          "--indel", "0.003", "--sub", "0.02"], ["--preset", "chip", "--chr-order", "chr4,chr2,chr5"]),
    "h1_hic_chrorder_q0": (["--genome", "3000000", "--chroms", "4", "--pairs", "20000", "--readlen", "150", "--frag-min", "300",
                            "--frag-max", "800", "--hic", "--seed", "21", "--indel", "0.001"],
                           ["--preset", "hic", "-q", "0", "--chr-order", "chr3,chr1"]),
    "h2_hic_natural_q0": (["--genome", "1000000", "--chroms", "3", "--pairs", "15000", "--readlen", "100", "--frag-min", "200",
                           "--frag-max", "500", "--hic", "--seed", "22", "--indel", "0.004", "--sub", "0.02"],
                          ["--preset", "hic", "-q", "0", "--pairs-natural-chr-order", "chr3,chr1,chr2"]),
    "s4_atac_q0": (["--genome", "300000", "--chroms", "2", "--pairs", "20000", "--readlen", "50", "--frag-min", "40",
                    "--seed", "5"], ["--preset", "atac", "-q", "0"]),
    # -n > 1: up to n best mappings per read (pair), reservoir sampling when there are more (mapping_generator.h:121-139,199-214)
    "s4_atac_n3_q0": (["--genome", "300000", "--chroms", "2", "--pairs", "20000", "--readlen", "50", "--frag-min", "40",
                       "--seed", "5"], ["--preset", "atac", "-q", "0", "-n", "3"]),
    "s4_se_n2_q0": (["--genome", "300000", "--chroms", "2", "--pairs", "20000", "--readlen", "50", "--frag-min", "40",
                     "--seed", "5"], ["--preset", "chip", "-q", "0", "-n", "2"]),
    # --drop-repetitive-reads: pairs with more best mappings than this are dropped (mapping_generator.h:190-193); single-end
    # reads are NOT (GenerateBestMappingsForSingleEndRead has no such test)
    "s4_atac_drop2_q0": (["--genome", "300000", "--chroms", "2", "--pairs", "20000", "--readlen", "50", "--frag-min", "40",
                          "--seed", "5"], ["--preset", "atac", "-q", "0", "--drop-repetitive-reads", "2"]),
    "s4_se_drop2_q0": (["--genome", "300000", "--chroms", "2", "--pairs", "20000", "--readlen", "50", "--frag-min", "40",
                        "--seed", "5"], ["--preset", "chip", "-q", "0", "--drop-repetitive-reads", "2"]),
    # 24 sequences (more owners than a 4- or 8-rank exchange has ranks to give them to, and empty owners at 8): bulk and single-cell
    # with duplicate removal at bulk level (the end-of-output MAPQ rule belongs to the last rank that owns records)
    "s5_atac_24chr_q0": (["--genome", "6000000", "--chroms", "24", "--pairs", "30000", "--readlen", "50", "--frag-min", "35", "--seed", "41"],
                         ["--preset", "atac", "-q", "0"]),
    "b4_bulk_level_bc_24chr_q0": (["--genome", "3000000", "--chroms", "24", "--pairs", "30000", "--readlen", "50", "--frag-min", "40",
                                   "--barcodes", "60", "--seed", "43", "--dup-frac", "0.3"],
                                  ["--preset", "atac", "--remove-pcr-duplicates-at-bulk-level", "-q", "0"]),
    # --preset hic --SAM: split alignment with ksw on the aligned part of the read, AdjustGapBeginning on the CIGAR, SEQ cut to the
    # CIGAR's query length (sam_mapping.h:186-193), all four strand combinations
    "h2_hic_sam_q0": (["--genome", "1000000", "--chroms", "3", "--pairs", "15000", "--readlen", "100", "--frag-min", "200",
                       "--frag-max", "500", "--hic", "--seed", "22", "--indel", "0.004", "--sub", "0.02"],
                      ["--preset", "hic", "--SAM", "-q", "0"]),
    "h1_hic_sam": (["--genome", "3000000", "--chroms", "4", "--pairs", "20000", "--readlen", "150", "--frag-min", "300",
                    "--frag-max", "800", "--hic", "--seed", "21", "--indel", "0.001"], ["--preset", "hic", "--SAM"]),
    "b2_atac_bc_n2_q0": (["--genome", "1000000", "--chroms", "3", "--pairs", "15000", "--readlen", "50", "--frag-min", "35",
                          "--barcodes", "300", "--seed", "32"], ["--preset", "atac", "-q", "0", "-n", "2"]),
    "h2_hic_n2_q0": (["--genome", "1000000", "--chroms", "3", "--pairs", "15000", "--readlen", "100", "--frag-min", "200",
                      "--frag-max", "500", "--hic", "--seed", "22", "--indel", "0.004", "--sub", "0.02"],
                     ["--preset", "hic", "-q", "0", "-n", "2"]),
    # off-preset parameters (round 5): the 8-lane verification form (error threshold below 8 without split alignment,
    # alignment.cc:503-654, mapping_parameters.h:80-88), a wide band, min seeds, seed-frequency caps that force the second
    # round (draft_mapping_generator / candidate_processor), a short insert limit, --min-read-length either side of 30
    "p1_e5_q0": (["--genome", "2000000", "--chroms", "4", "--pairs", "20000", "--readlen", "70", "--seed", "51", "--sub", "0.02",
                  "--indel", "0.003"], ["-e", "5", "-q", "0"]),
    "p1_e5_atac": (["--genome", "2000000", "--chroms", "4", "--pairs", "20000", "--readlen", "50", "--frag-min", "35"],
                   ["--preset", "atac", "-e", "5"]),
    "p1_e3_chip_q0": (["--genome", "2000000", "--chroms", "4", "--pairs", "20000", "--readlen", "70", "--seed", "51", "--sub", "0.02",
                       "--indel", "0.003"], ["--preset", "chip", "-e", "3", "-q", "0"]),
    "p1_se_e5_q0": (["--genome", "2000000", "--chroms", "4", "--pairs", "20000", "--readlen", "70", "--seed", "51", "--sub", "0.02",
                     "--indel", "0.003"], ["-e", "5", "-q", "0"]),
    "p1_e5_sam_q0": (["--genome", "2000000", "--chroms", "4", "--pairs", "20000", "--readlen", "70", "--seed", "51", "--sub", "0.02",
                      "--indel", "0.003"], ["-e", "5", "-q", "0", "--SAM"]),
    "p1_e12_chip_q0": (["--genome", "2000000", "--chroms", "4", "--pairs", "20000", "--readlen", "100", "--seed", "52", "--sub", "0.04",
                        "--indel", "0.005"], ["--preset", "chip", "-e", "12", "-q", "0"]),
    "p2_s3_l300_q0": (["--genome", "2000000", "--chroms", "4", "--pairs", "20000", "--readlen", "50", "--frag-min", "35"],
                      ["--preset", "chip", "-s", "3", "-l", "300", "-q", "0"]),
    "p2_s1_q0": (["--genome", "2000000", "--chroms", "4", "--pairs", "20000", "--readlen", "50", "--frag-min", "35"],
                 ["--preset", "atac", "-s", "1", "-q", "0"]),
    "p3_f40_90_q0": (["--genome", "2000000", "--chroms", "4", "--pairs", "20000", "--readlen", "50", "--frag-min", "35"],
                     ["--preset", "atac", "-f", "40,90", "-q", "0"]),
    "p3_f5_20_q0": (["--genome", "2000000", "--chroms", "4", "--pairs", "20000", "--readlen", "50", "--frag-min", "35"],
                    ["--preset", "chip", "-f", "5,20", "-q", "0"]),
    "p4_minlen40_q0": (["--genome", "2000000", "--chroms", "3", "--pairs", "20000", "--readlen", "60", "--frag-min", "35",
                        "--varlen", "--seed", "7"], ["--preset", "atac", "--min-read-length", "40", "-q", "0"]),
    "p4_minlen20_q0": (["--genome", "2000000", "--chroms", "3", "--pairs", "20000", "--readlen", "60", "--frag-min", "35",
                        "--varlen", "--seed", "7"], ["--preset", "atac", "--min-read-length", "20", "-q", "0"]),
    "p5_n5_q0": (["--genome", "300000", "--chroms", "2", "--pairs", "20000", "--readlen", "50", "--frag-min", "40",
                  "--seed", "5"], ["--preset", "atac", "-n", "5", "-q", "0"]),
    # combinations: off-preset values together, with cell barcodes, with split alignment
    "p6_combo_q0": (["--genome", "2000000", "--chroms", "4", "--pairs", "20000", "--readlen", "70", "--seed", "51", "--sub", "0.02",
                     "--indel", "0.003"], ["--preset", "chip", "-e", "10", "-s", "3", "-f", "20,60", "-l", "500", "--min-read-length", "35", "-q", "0"]),
    "b5_bc_e5_f40_q0": (["--genome", "2000000", "--chroms", "4", "--pairs", "20000", "--readlen", "50", "--frag-min", "35",
                         "--barcodes", "500", "--seed", "31"], ["--preset", "atac", "-e", "5", "-f", "40,90", "-q", "0"]),
    "h3_hic_e6_q0": (["--genome", "1000000", "--chroms", "3", "--pairs", "15000", "--readlen", "100", "--frag-min", "200",
                      "--frag-max", "500", "--hic", "--seed", "22", "--indel", "0.004", "--sub", "0.02"], ["--preset", "hic", "-e", "6", "-q", "0"]),
    "p5_se_n5_e5_q0": (["--genome", "300000", "--chroms", "2", "--pairs", "20000", "--readlen", "50", "--frag-min", "40",
                        "--seed", "5"], ["--preset", "chip", "-n", "5", "-e", "5", "-q", "0"]),
}


def md5(path):
    h = hashlib.md5()
    with open(path, "rb") as f:
        for b in iter(lambda: f.read(1 << 20), b""):
            h.update(b)
    return h.hexdigest()


def main():
    only = set(sys.argv[1:])
    for name, (gen, flags) in CASES.items():
        if only and name not in only:
            continue
        tmp = tempfile.mkdtemp(prefix="golden_")
        try:
            if gen is None:
                fa, r1, r2 = (os.path.join(HERE, "toy", f) for f in ("ref.fa", "read1.fq", "read2.fq"))
            else:
                subprocess.check_call([sys.executable, GEN, "--out", os.path.join(tmp, "d")] + gen)
                fa, r1, r2 = (os.path.join(tmp, f) for f in ("d.fa", "d_1.fq", "d_2.fq"))
            idx = os.path.join(tmp, "d.idx")
            subprocess.check_call([REF, "-i", "-r", fa, "-o", idx], stderr=subprocess.DEVNULL)
            out = os.path.join(tmp, "out.bed")
            extra = []
            if gen is not None and "--barcodes" in gen:
                extra = ["-b", os.path.join(tmp, "d_bc.fq"), "--barcode-whitelist", os.path.join(tmp, "d.whitelist.txt")]
            reads = ["-1", r1, "-2", r2]
            if name in SINGLE_END:
                reads = ["-1", r1 if SINGLE_END[name] == 1 else r2]
            run_flags = list(flags)
            for of in ("--chr-order", "--pairs-natural-chr-order"):
                if of in run_flags:
                    k = run_flags.index(of)
                    order_file = os.path.join(tmp, of.strip("-") + ".txt")
                    with open(order_file, "w") as f:
                        f.write("\n".join(run_flags[k + 1].split(",")) + "\n")
                    run_flags[k + 1] = order_file
            log = subprocess.run([REF] + run_flags + extra + ["-x", idx, "-r", fa] + reads + ["-o", out, "-t", "1"],
                                 stderr=subprocess.PIPE, check=True).stderr.decode()
            stats = {}
            for key, pat in (("num_reads", r"Number of reads: (\d+)"), ("num_mapped_reads", r"Number of mapped reads: (\d+)"),
                             ("num_uniquely_mapped_reads", r"Number of uniquely mapped reads: (\d+)"),
                             ("num_candidates", r"Number of candidates: (\d+)"),
                             ("num_mappings", r"Number of mappings: (\d+)"),
                             ("num_output", r"Number of output mappings \(passed filters\): (\d+)"),
                             ("num_barcode_in_whitelist", r"Number of barcodes in whitelist: (\d+)"),
                             ("num_corrected_barcode", r"Number of corrected barcodes: (\d+)")):
                m = re.search(pat, log)
                stats[key] = int(m.group(1)) if m else None
            meta = {"generator_args": gen, "chromap_flags": flags, "single_end_mate": SINGLE_END.get(name, 0), "reference_version": "0.3.3-r521",
                    "input_md5": dict({"fa": md5(fa), "r1": md5(r1), "r2": md5(r2)},
                                      **({"bc": md5(extra[1]), "whitelist": md5(extra[3])} if extra else {})),
                    "index_md5_reference_build": md5(idx), "bed_md5": md5(out), "reference_stderr_counters": stats}
            ext = ".sam.gz" if "--SAM" in flags else ".pairs.gz" if "hic" in flags else ".bed.gz"
            with open(out, "rb") as f, gzip.GzipFile(os.path.join(HERE, name + ext), "wb", mtime=0) as g:
                shutil.copyfileobj(f, g)
            with open(os.path.join(HERE, name + ".json"), "w") as f:
                json.dump(meta, f, indent=1, sort_keys=True)
            print(name, stats)
        finally:
            shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
