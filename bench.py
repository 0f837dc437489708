#!/usr/bin/env python3
"""bench.py -- M read pairs mapped per second through the hot path (K0-K5) on MI355X.

One "step" = one pass of the hot path over one batch of synthetic read pairs that is
already resident in HBM (inputs are generated on the device; the PCIe-inclusive rate is
noted in DESIGN.md).  Workload (BASELINE.json metric / configs[2], per-GPU weak scaling):
--preset atac, synthetic 2x50 bp pairs, GRCh38-sized synthetic index (3.1e9 random bases in
24 sequences, k=17, w=7) replicated on every GPU; at N>1 every rank maps its own batch and
the records are all-gathered over RCCL inside the step (the exchange the final global
sort/dedup needs).  Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpu_baseline_reference(args):
    """the reference binary itself (oracle/_ref/chromap, built unchanged from the reference sources; it
    travels with the repo) on the host cores: tools/ref_baseline.py writes the same GRCh38-sized
    synthetic index / genome in the reference's file formats and two bench batches as FASTQ, runs
    `chromap --preset atac -t <nproc>` and chromap-amd on them and compares the BED files"""
    import subprocess
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "chromap")
    if not os.path.exists(ref_bin):
        return None
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ref_baseline.py"), "--genome", str(args.genome), "--nseq",
                        str(args.nseq), "--pairs", str(args.pairs), "--batches", "2", "--readlen", str(args.readlen)],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500)
    try:
        r = json.loads(p.stdout.decode().strip().splitlines()[-1])
    except Exception:
        return None
    ref = r.get("reference", {})
    if "error" in r or "error" in ref or not ref.get("mapped_all_reads_s"):
        log("[bench] reference baseline unavailable: %s" % (r.get("error") or ref.get("error")))
        return None
    return {"value": ref["M_pairs_per_s_mapping_loop"], "unit": "M pairs/s", "cores": r["threads"], "kind": "reference",
            "sample": "%d pairs (two bench batches, seeds 1000/1001) as FASTQ through oracle/_ref/chromap --preset atac -t %d with the "
                      "GRCh38-sized index written by the device builder: %.2f s in its mapping loop (\"Mapped all reads in\"), "
                      "%.2f s summed over its %d batches" % (r["pairs"], r["threads"], ref["mapped_all_reads_s"],
                                                             ref["sum_of_batch_times_s"], ref["batches"]),
            "bed_identical_to_reference": r.get("bed_identical_to_reference"),
            "bed_lines": ref.get("bed_lines"), "chromap_amd_cli_same_files": r.get("chromap_amd", {}).get("cli")}


def cpu_baseline(g, args, n_sample_hint):
    """oracle ("port") on the host cores, same index, same reads, bounded sample"""
    import numpy as np
    import oracle_lib as ol
    import psutil
    L = g.L
    k = C.c_int32()
    w = C.c_int32()
    nb = C.c_uint32()
    nocc = C.c_uint32()
    L.cmgpu_index_info(g.ctx, C.byref(k), C.byref(w), C.byref(nb), C.byref(nocc), None, None)
    need = nb.value * 32 + nocc.value * 16 + args.genome * 1.1 + (2 << 30)
    avail = psutil.virtual_memory().available
    if avail < need:
        return {"value": None, "unit": "M pairs/s", "cores": 0, "kind": "port",
                "sample": "skipped: host has %.0f GB free, %.0f GB needed to hold the exported index" % (avail / 2**30, need / 2**30)}
    t0 = time.time()
    bk = np.empty(nb.value * 2, np.uint64)
    oc = np.empty(max(1, nocc.value), np.uint64)
    assert L.cmgpu_export_index(g.ctx, bk.ctypes.data, oc.ctypes.data) == 0
    O = ol.lib()
    idx = ol.OraIndex()
    assert O.ora_index_from_buckets(bk.ctypes.data, nb.value, oc.ctypes.data, nocc.value, k.value, w.value, C.byref(idx)) == 0
    del bk
    # reference sequences
    nseq = C.c_uint32()
    L.cmgpu_reference_lengths(g.ctx, None, 0, C.byref(nseq))
    lens = (C.c_uint32 * nseq.value)()
    L.cmgpu_reference_lengths(g.ctx, lens, nseq.value, C.byref(nseq))
    ref = ol.OraRef()
    ref.n_seq = nseq.value
    names = (C.c_char_p * nseq.value)(*[b"chr%d" % (i + 1) for i in range(nseq.value)])
    seqs = (C.c_void_p * nseq.value)()
    bufs = []
    for i in range(nseq.value):
        b = C.create_string_buffer(lens[i] + 64)
        assert L.cmgpu_export_reference(g.ctx, i, b, lens[i]) == 0
        bufs.append(b)
        seqs[i] = C.addressof(b)
    ref.name = names
    ref.seq = seqs
    ref.len = lens
    p = ol.params(args.preset)
    ctx = O.ora_create(C.byref(idx), C.byref(ref), C.byref(p))
    log("[bench] exported index to host in %.1fs" % (time.time() - t0))
    # the resident batch
    n = args.pairs
    rl = args.readlen
    b1 = np.empty(n * rl, np.uint8)
    b2 = np.empty(n * rl, np.uint8)
    o1 = np.empty(n + 1, np.uint32)
    o2 = np.empty(n + 1, np.uint32)
    assert L.cmgpu_download_batch(g.ctx, b1.ctypes.data, o1.ctypes.data, b2.ctypes.data, o2.ctypes.data) == 0
    cores = len(os.sched_getaffinity(0))

    def run(m):
        rec = (ol.OraRecord * m)()
        st = ol.OraStats()
        t = time.time()
        kk = O.ora_map_pairs_mt(ctx, cores, m, 0, b1.ctypes.data, o1.ctypes.data, b2.ctypes.data, o2.ctypes.data,
                                C.cast(rec, C.c_void_p), C.byref(st))
        return time.time() - t, kk, rec, st

    m = n  # the whole timed batch, repeated until ~12 s of CPU work have been measured
    t_run, kk, rec, st = run(m)
    reps, t_total = 1, t_run
    while t_total < 12.0 and reps < 64:
        t_more, _, _, _ = run(m)
        t_total += t_more
        reps += 1
    t_run = t_total / reps
    # parity of the sample: the GPU records of the same pairs must be identical
    grec, gk = g.download_records(n)
    gset = {}
    for i in range(gk):
        r = grec[i]
        if r.read_id < m:
            gset[r.read_id] = (r.rid, r.fragment_start, r.fragment_length, r.mapq, r.direction, r.is_unique,
                               r.positive_alignment_length, r.negative_alignment_length)
    oset = {}
    for i in range(kk):
        r = rec[i]
        oset[r.read_id] = (r.rid, r.fragment_start, r.fragment_length, r.mapq, r.direction, r.is_unique,
                           r.pos_aln_len, r.neg_aln_len)
    O.ora_destroy(ctx)
    return {"value": round(m / t_run / 1e6, 4), "unit": "M pairs/s", "cores": cores, "kind": "port",
            "sample": "the %d pairs of the timed batch x %d repeats (%.1f s of CPU work), same GRCh38-sized index exported from "
                      "HBM, OpenMP over %d host threads" % (m, reps, t_total, cores),
            "records_identical_to_gpu": gset == oset, "sample_records": len(oset),
            "algorithmic_probe_steps_per_pair": round(st.probe_steps / m, 2)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--pairs", type=int, default=4_000_000, help="read pairs per GPU per step")
    ap.add_argument("--genome", type=int, default=3_100_000_000)
    ap.add_argument("--nseq", type=int, default=24)
    ap.add_argument("--readlen", type=int, default=50)
    ap.add_argument("--preset", default="atac")
    ap.add_argument("--skip-cpu", action="store_true")
    ap.add_argument("--cpu-baseline", choices=["auto", "reference", "port"], default="auto",
                    help="reference: oracle/_ref/chromap itself; port: the oracle restatement with OpenMP; auto: reference if its binary is here")
    ap.add_argument("--probe-repeat", type=int, default=10)
    ap.add_argument("--sam", action="store_true", help="--SAM mode: every reported read is aligned with the banded affine-gap DP "
                                                         "(CIGAR / NM / MD); not the headline metric")
    ap.add_argument("--force-exchange", action="store_true",
                    help="run the RCCL record exchange even with one rank (exercises the N>1 code path on a 1-GPU box)")
    args = ap.parse_args()

    # RCCL prints a version banner on the process's stdout; the contract is ONE JSON line there.
    # Everything below writes to stderr; the JSON goes to the saved descriptor at the very end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (chromap_amd has no CPU path)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or args.force_exchange:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus

    from chromap_amd import ChromapGPU, Stats
    t0 = time.time()
    g = ChromapGPU(synthetic=(args.genome, args.nseq, 12345), preset=args.preset, device=local_rank,
                   **({"output_format": 1} if args.sam else {}))
    t_index = time.time() - t0
    if rank == 0:
        log("[bench] synthetic genome + index on device in %.1fs" % t_index)
    g.generate_resident(args.pairs, read_length=args.readlen, frag_min=30, frag_max=600, sub_rate=0.01, seed=1000 + rank)
    ex = None
    if world > 1 or args.force_exchange:
        from chromap_amd.distributed import RecordExchange
        ex = RecordExchange(args.pairs, torch.device("cuda", local_rank))

    def step(stats):
        k = g.map_resident(stats)
        if ex is not None:
            # records grouped by chromosome owner -> all-to-all -> the owner's device-side record store
            counts = (C.c_uint64 * world)()
            rc = g.L.cmgpu_records_partition(g.ctx, world, C.c_void_p(ex.send.data_ptr()), args.pairs, counts)
            assert rc == 0
            nrecv = ex.all_to_all(list(counts))
            g.store_append(ex.recv.data_ptr(), nrecv, on_device=True)
        return k

    for _ in range(args.warmup):
        step(Stats())
    g.store_clear()
    stage_ms = {}
    st = Stats()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    mapped = 0
    for _ in range(args.steps):
        mapped += step(st)
        for name, ms in g.timings():
            stage_ms[name] = stage_ms.get(name, 0.0) + ms
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    total_pairs = args.pairs * args.steps * world
    value = total_pairs / dt / 1e6

    if rank == 0:
        s = st.as_dict()
        steps = max(1, args.steps)
        probe_ms = stage_ms.get("s2_probe", 0.0) / steps
        probe_steps = s["probe_steps"] / steps
        alg_bytes = 16.0 * probe_steps  # one 8-B key + one 8-B value per visited bucket
        achieved = alg_bytes / (probe_ms * 1e-3) / 1e9 if probe_ms > 0 else 0.0
        # kernel-only re-measurement of the same launch (HIP events around `repeat` launches)
        avg = C.c_double(0)
        ps = C.c_uint64(0)
        hits = C.c_uint64(0)
        n_mm = s["num_minimizers"] // steps
        rc = g.L.cmgpu_probe_bench(g.ctx, None, n_mm, args.probe_repeat, C.byref(avg), C.byref(ps), C.byref(hits), None)
        probe_only = None
        if rc == 0 and avg.value > 0:
            probe_only = {"lookups": int(n_mm), "avg_ms": round(avg.value, 4), "probe_steps": int(ps.value),
                          "hits": int(hits.value), "GB/s": round(16.0 * ps.value / (avg.value * 1e-3) / 1e9, 1),
                          "G_lookups/s": round(n_mm / (avg.value * 1e-3) / 1e9, 2)}
            achieved = probe_only["GB/s"]
            probe_ms = avg.value
        gavg = C.c_double(0)
        gather = None
        ng = 1 << 28
        if g.L.cmgpu_gather_bench(g.ctx, ng, 5, C.byref(gavg)) == 0 and gavg.value > 0:
            gather = {"accesses": ng, "avg_ms": round(gavg.value, 4), "useful_GB/s": round(16.0 * ng / (gavg.value * 1e-3) / 1e9, 1),
                      "sector_GB/s": round(64.0 * ng / (gavg.value * 1e-3) / 1e9, 1)}
        traffic = None
        prof = os.path.join(ROOT, "profiles", "probe_traffic.json")
        if os.path.exists(prof):
            try:
                traffic = json.load(open(prof)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        roof = {"kernel": "k_probe", "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "algorithmic_bytes_per_launch": int(alg_bytes), "launch_ms": round(probe_ms, 4),
                "probe_only": probe_only, "random_gather_16B": gather,
                "frac_of_measured_gather": round(achieved / gather["useful_GB/s"], 3) if gather and gather["useful_GB/s"] else None}
        # device-side post-processing, outside the timed region: the last batch's
        # records plus three freshly mapped batches go to the record store; one call sorts,
        # de-duplicates, filters and renders the BED text in HBM
        post = None
        try:
            g.store_clear()
            t1 = time.perf_counter()
            nrec = g.store_append_resident()
            for extra in range(3):
                g.generate_resident(args.pairs, read_length=args.readlen, frag_min=30, frag_max=600, sub_rate=0.01,
                                    seed=5000 + extra)
                g.map_resident(Stats())
                nrec = g.store_append_resident()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            lines, nbytes = g.store_format(0)
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            post = {"records": int(nrec), "bed_lines": int(lines), "text_bytes": int(nbytes),
                    "sort_dedup_format_ms": round((t3 - t2) * 1e3, 3),
                    "M_records/s": round(nrec / (t3 - t2) / 1e6, 1)}
            g.store_clear()
        except Exception as e:
            post = {"error": repr(e)}
        # back to the timed batch (the CPU baseline compares its records with the GPU's)
        g.generate_resident(args.pairs, read_length=args.readlen, frag_min=30, frag_max=600, sub_rate=0.01, seed=1000 + rank)
        g.map_resident(Stats())
        # PCIe-inclusive rate of the host-buffer boundary (cmgpu_map_pairs: H2D reads, map, D2H records);
        # reported beside `value`, never as `value`
        pcie = None
        try:
            import numpy as np
            n = args.pairs
            o1 = np.zeros(n + 1, np.uint32)
            o2 = np.zeros(n + 1, np.uint32)
            b1 = np.zeros(n * args.readlen, np.uint8)
            b2 = np.zeros(n * args.readlen, np.uint8)
            assert g.L.cmgpu_download_batch(g.ctx, b1.ctypes.data, o1.ctypes.data, b2.ctypes.data, o2.ctypes.data) == 0
            g.map_pairs(b1, o1, b2, o2)  # warm the host-side buffers
            t1 = time.perf_counter()
            _, kk = g.map_pairs(b1, o1, b2, o2)
            t2 = time.perf_counter()
            pcie = {"M pairs/s": round(n / (t2 - t1) / 1e6, 2), "ms": round((t2 - t1) * 1e3, 2), "records": int(kk),
                    "note": "pageable host buffers, synchronous hipMemcpy, includes allocating the host record array"}
        except Exception as e:
            pcie = {"error": repr(e)}
        cpu = None
        if world == 1 and not args.skip_cpu and not args.sam:
            try:
                if args.cpu_baseline in ("auto", "reference"):
                    cpu = cpu_baseline_reference(args)
                if cpu is None and args.cpu_baseline in ("auto", "port"):
                    cpu = cpu_baseline(g, args, args.pairs)
            except Exception as e:  # the GPU number must still be reported
                cpu = {"value": None, "unit": "M pairs/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}
        out = {
            "metric": "M paired reads mapped/s (ATAC preset, GRCh38 index)",
            "value": round(value, 4), "unit": "M pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": "--preset %s, synthetic 2x%d bp pairs (fragments 30-600 bp, 1%% substitutions), "
                                   "GRCh38-sized synthetic index (%.2e bases, %d sequences, k=17 w=7) resident per GPU, "
                                   "%d pairs per GPU per step, reads resident in HBM" % (args.preset, args.readlen, args.genome, args.nseq, args.pairs),
                       "pairs_per_gpu_per_step": args.pairs, "parallelism": "read-shard x%d + RCCL all-to-all of records to chromosome owners" % world if world > 1 else "single GPU"},
            "roofline": roof, "cpu_baseline": cpu, "postprocess_on_device": post, "pcie_inclusive": pcie,
            "stage_ms_per_step": {k: round(v / steps, 3) for k, v in stage_ms.items()},
            "counters_per_step": {k: v // steps for k, v in s.items()},
            "mapped_pairs_per_step": mapped // steps, "index_build_s": round(t_index, 1),
        }
        result_line = json.dumps(out)
    else:
        result_line = None
    g.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    sys.stdout.flush()
    if result_line is not None:
        os.write(json_fd, (result_line + "\n").encode())
    os.close(json_fd)


if __name__ == "__main__":
    main()
