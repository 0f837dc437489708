#!/usr/bin/env python3
"""The REFERENCE itself (oracle/_ref/chromap, built unchanged from the reference sources by
`make -C oracle ref`; the binary travels to the GPU box) on the bench workload, on the GPU box's host
cores: GRCh38-sized synthetic genome + index written in the reference's formats by the device
builder, synthetic pairs as FASTQ, `chromap --preset atac -t <nproc>`; then the same files through
chromap-amd, and the two BED files compared.  Prints one JSON object."""
import argparse
import ctypes as C
import json
import os
import re
import shutil
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome", type=int, default=3_100_000_000)
    ap.add_argument("--nseq", type=int, default=24)
    ap.add_argument("--pairs", type=int, default=4_000_000)
    ap.add_argument("--batches", type=int, default=2)
    ap.add_argument("--readlen", type=int, default=50)
    ap.add_argument("--dir", default="/tmp/chromap_amd_ref")
    ap.add_argument("--threads", type=int, default=0, help="-t of the reference (0: the processors the container's CPU quota gives this process)")
    ap.add_argument("--preset", default="atac")
    ap.add_argument("--repeats", default="", help="families,copies,element_len,divergence of the planted repeats (bench.py --repeats)")
    ap.add_argument("--indel-rate", type=float, default=0.0)
    ap.add_argument("--frag-min", type=int, default=30)
    ap.add_argument("--frag-max", type=int, default=600)
    ap.add_argument("--seed0", type=int, default=1000)
    ap.add_argument("--barcodes", type=int, default=0, help="single-cell run: a whitelist of this many random 16-mers, every pair draws one, "
                                                            "10 %% of them with one substitution (BASELINE config 4: 737280)")
    ap.add_argument("--bgzf-check", action="store_true", help="also run chromap-amd on BGZF-compressed copies of the read files: same output")
    ap.add_argument("--hic", type=float, default=-1.0, help="Hi-C shaped pairs with this fraction of chimeric reads (default: fragments)")
    args = ap.parse_args()
    if args.threads <= 0:
        from chromap_amd.cpus import cpu_budget
        args.threads = cpu_budget()
    rep = None
    if args.repeats.startswith("profile:"):
        rep = args.repeats
    elif args.repeats:
        f = args.repeats.split(",")
        rep = (int(f[0]), int(f[1]), int(f[2]), float(f[3]))
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "chromap")
    if not os.path.exists(ref_bin):
        print(json.dumps({"error": "oracle/_ref/chromap not present"}))
        return
    os.makedirs(args.dir, exist_ok=True)
    free = shutil.disk_usage(args.dir).free
    need = int(args.genome * 7.5) + args.batches * args.pairs * 300
    if free < need:
        print(json.dumps({"error": "not enough disk in %s: %.0f GB free, %.0f GB needed" % (args.dir, free / 1e9, need / 1e9)}))
        return
    from e2e_bench import write_fastq
    from chromap_amd import ChromapGPU
    t0 = time.time()
    g = ChromapGPU(synthetic=(args.genome, args.nseq, 12345, rep), preset=args.preset)
    idx = os.path.join(args.dir, "g.index")
    fa = os.path.join(args.dir, "g.fa")
    g.save_index(idx)
    nseq = C.c_uint32(0)
    g.L.cmgpu_reference_lengths(g.ctx, None, 0, C.byref(nseq))
    lens = (C.c_uint32 * nseq.value)()
    g.L.cmgpu_reference_lengths(g.ctx, lens, nseq.value, C.byref(nseq))
    with open(fa, "wb") as f:
        for i in range(nseq.value):
            buf = C.create_string_buffer(lens[i])
            assert g.L.cmgpu_export_reference(g.ctx, i, buf, lens[i]) == 0
            f.write(b">chr%d\n" % (i + 1))
            f.write(buf.raw[:lens[i]])
            f.write(b"\n")
    r1 = os.path.join(args.dir, "r1.fq")
    r2 = os.path.join(args.dir, "r2.fq")
    for f in (r1, r2):
        if os.path.exists(f):
            os.remove(f)
    for b in range(args.batches):
        g.generate_resident(args.pairs, read_length=args.readlen, frag_min=args.frag_min, frag_max=args.frag_max, sub_rate=0.01,
                            seed=args.seed0 + b, indel_rate=args.indel_rate, hic=args.hic if args.hic >= 0 else None)
        b1, o1, b2, o2 = g.download_batch(args.pairs)
        for path, bases in ((r1, b1), (r2, b2)):
            tmp = path + ".part"
            write_fastq(tmp, bases, args.pairs, args.readlen)
            with open(path, "ab") as dst, open(tmp, "rb") as src:
                shutil.copyfileobj(src, dst, 1 << 24)
            os.remove(tmp)
    g.close()
    extra = []
    if args.barcodes:
        import numpy as np
        rng = np.random.default_rng(args.seed0)
        n_all = args.pairs * args.batches
        wl = np.unique(rng.integers(0, 1 << 32, size=args.barcodes, dtype=np.uint64))  # 16-mers as 32-bit codes
        acgt = np.frombuffer(b"ACGT", np.uint8)
        shifts = (2 * np.arange(15, -1, -1)).astype(np.uint64)

        def letters(codes):
            return acgt[((codes[:, None] >> shifts[None, :]) & np.uint64(3)).astype(np.int64)]
        wl_path = os.path.join(args.dir, "whitelist.txt")
        with open(wl_path, "wb") as f:
            f.write(b"\n".join(bytes(x) for x in letters(wl)) + b"\n")
        pick = wl[rng.integers(0, len(wl), size=n_all)]
        seq = letters(pick)
        hit = rng.random(n_all) < 0.10
        col = rng.integers(0, 16, size=n_all)
        rows = np.nonzero(hit)[0]
        seq[rows, col[rows]] = acgt[rng.integers(0, 4, size=len(rows))]
        rec = np.empty((n_all, 4 + 17 + 2 + 17), np.uint8)  # "@b\n" + 16 + "\n+\n" + 16 x 'I' + "\n"
        rec[:, 0:3] = np.frombuffer(b"@b\n", np.uint8)
        rec[:, 3:19] = seq
        rec[:, 19:22] = np.frombuffer(b"\n+\n", np.uint8)
        rec[:, 22:38] = ord("I")
        rec[:, 38] = ord("\n")
        bc_path = os.path.join(args.dir, "bc.fq")
        rec[:, :39].tofile(bc_path)
        extra = ["-b", bc_path, "--barcode-whitelist", wl_path]
    t_setup = time.time() - t0
    n_pairs = args.pairs * args.batches
    res = {"setup_s": round(t_setup, 1), "pairs": n_pairs, "threads": args.threads,
           "index_bytes": os.path.getsize(idx), "fastq_bytes": os.path.getsize(r1) + os.path.getsize(r2)}
    # ---- the reference
    out_ref = os.path.join(args.dir, "ref.out")
    t0 = time.time()
    p = subprocess.run([ref_bin, "--preset", args.preset, "-x", idx, "-r", fa, "-1", r1, "-2", r2, "-o", out_ref, "-t", str(args.threads)] + extra,
                       stderr=subprocess.PIPE)
    wall = time.time() - t0
    log = p.stderr.decode(errors="replace")
    if p.returncode != 0:
        res["reference"] = {"error": log[-2000:]}
    else:
        m_all = re.search(r"Mapped all reads in ([0-9.]+)s", log)
        per_batch = [float(x) for x in re.findall(r"Mapped \d+ read pairs in ([0-9.]+)s", log)]
        res["reference"] = {"wall_s": round(wall, 2), "mapped_all_reads_s": float(m_all.group(1)) if m_all else None,
                            "sum_of_batch_times_s": round(sum(per_batch), 3), "batches": len(per_batch),
                            "M_pairs_per_s_mapping_loop": round(n_pairs / float(m_all.group(1)) / 1e6, 3) if m_all else None,
                            "M_pairs_per_s_batches": round(n_pairs / sum(per_batch) / 1e6, 3) if per_batch else None,
                            "bed_md5": subprocess.check_output(["md5sum", out_ref]).split()[0].decode(),
                            "bed_lines": int(subprocess.check_output(["wc", "-l", out_ref]).split()[0])}
    # ---- chromap-amd on the same files
    out_gpu = os.path.join(args.dir, "gpu.out")
    cli = os.path.join(ROOT, "chromap_amd", "chromap-amd")
    t0 = time.time()
    p = subprocess.run([cli, "--preset", args.preset, "-x", idx, "-r", fa, "-1", r1, "-2", r2, "-o", out_gpu] + extra, stderr=subprocess.PIPE)
    wall = time.time() - t0
    log = p.stderr.decode(errors="replace")
    if p.returncode != 0:
        res["chromap_amd"] = {"error": log[-2000:]}
    else:
        tail = [ln for ln in log.splitlines() if ln.startswith("Mapped all reads")]
        res["chromap_amd"] = {"wall_s_incl_index_load": round(wall, 2), "cli": tail,
                              "bed_md5": subprocess.check_output(["md5sum", out_gpu]).split()[0].decode()}
    if "bed_md5" in res.get("reference", {}) and "bed_md5" in res.get("chromap_amd", {}):
        res["bed_identical_to_reference"] = res["reference"]["bed_md5"] == res["chromap_amd"]["bed_md5"]
    if args.bgzf_check and "bed_md5" in res.get("chromap_amd", {}):
        # the same reads as BGZF (block-parallel inflate in the CLI's reader): the output must not change
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bgzf
        z1, z2 = r1 + ".bgz", r2 + ".bgz"
        bgzf.compress_file(r1, z1)
        bgzf.compress_file(r2, z2)
        out_z = os.path.join(args.dir, "gpu_bgzf.out")
        zextra = list(extra)
        p = subprocess.run([cli, "--preset", args.preset, "-x", idx, "-r", fa, "-1", z1, "-2", z2, "-o", out_z] + zextra, stderr=subprocess.PIPE)
        res["bgzf"] = {"rc": p.returncode, "identical": p.returncode == 0 and
                       subprocess.check_output(["md5sum", out_z]).split()[0].decode() == res["chromap_amd"]["bed_md5"]}
    shutil.rmtree(args.dir, ignore_errors=True)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
