#!/usr/bin/env python3
"""bench.py -- M read pairs mapped per second through the hot path (K0-K5) on MI355X.

One "step" = one pass of the hot path over one batch of synthetic read pairs that is already
resident in HBM (inputs are generated on the device; the PCIe-inclusive rate of the host-buffer
boundary is reported beside it, never as `value`).  Workload (BASELINE.json metric / configs[2],
per-GPU weak scaling): --preset atac, synthetic 2x50 bp pairs, GRCh38-sized synthetic index
(3.1e9 random bases in 24 sequences, k=17, w=7) replicated on every GPU.  Four distinct batches per
rank stay resident and take turns.  At N>1 every rank maps its own batches and, inside the step,
the records go to the ranks that own their chromosomes (device partition -> RCCL all-to-all issued
by the library on its mapping stream -> the owner's device-side record store): the exchange the
final sort / duplicate removal needs.

`python bench.py --gpus N` starts its N ranks itself when it was not launched under
torch.distributed.run (WORLD_SIZE unset); under the launcher it reads RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_*.  torch.distributed (gloo) carries the control plane only: the rendezvous of
the RCCL unique id, the barriers around the timed region and the max over ranks.
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
N_SLOTS = 4            # distinct resident batches per rank that take turns


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(args):
    """python bench.py --gpus N without a launcher: re-run under torch.distributed.run, one rank per GPU"""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.call(cmd, env=env)


def host_info():
    """hardware threads the box reports against the processors the container's CPU quota gives the job (the CPU baseline's threads)"""
    from chromap_amd.cpus import cpu_budget
    return {"hardware_threads": os.cpu_count(), "cpu_budget": cpu_budget()}


def cpu_baseline_reference(args, repeats=None, tag="", seed0=1000, pairs=None):
    """the reference binary itself (oracle/_ref/chromap, built unchanged from the reference sources; it
    travels with the repo) on the host cores: tools/ref_baseline.py writes the same GRCh38-sized
    synthetic index / genome in the reference's file formats and two bench batches as FASTQ, runs
    `chromap --preset atac -t <nproc>` and chromap-amd on them and compares the BED files"""
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "chromap")
    if not os.path.exists(ref_bin):
        return None
    cmd = [sys.executable, os.path.join(ROOT, "tools", "ref_baseline.py"), "--genome", str(args.genome), "--nseq",
           str(args.nseq), "--pairs", str(pairs or args.pairs), "--batches", "2", "--readlen", str(args.readlen), "--preset", args.preset,
           "--seed0", str(seed0), "--indel-rate", str(args.indel_rate)]
    from chromap_amd.cpus import cpu_budget
    cmd += ["--threads", str(cpu_budget())]  # (-t for the reference: the processors the container's quota really gives it, not the 256 it reports)
    if repeats:
        cmd += ["--repeats", repeats if isinstance(repeats, str) else ",".join(str(x) for x in repeats)]
    if args.hic >= 0:
        cmd += ["--hic", str(args.hic)]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500)
    try:
        r = json.loads(p.stdout.decode().strip().splitlines()[-1])
    except Exception:
        log("[bench] reference baseline%s: no result (%s)" % (tag, p.stderr.decode(errors="replace")[-500:]))
        return None
    ref = r.get("reference", {})
    if "error" in r or "error" in ref or not ref.get("mapped_all_reads_s"):
        log("[bench] reference baseline%s unavailable: %s" % (tag, r.get("error") or ref.get("error")))
        return None
    return {"value": ref["M_pairs_per_s_mapping_loop"], "unit": "M pairs/s", "cores": r["threads"], "kind": "reference",
            "sample": "%d pairs (two bench batches, seeds %d/%d) as FASTQ through oracle/_ref/chromap --preset %s -t %d with the "
                      "GRCh38-sized index written by the device builder: %.2f s in its mapping loop (\"Mapped all reads in\"), "
                      "%.2f s summed over its %d batches" % (r["pairs"], seed0, seed0 + 1, args.preset, r["threads"], ref["mapped_all_reads_s"],
                                                             ref["sum_of_batch_times_s"], ref["batches"]),
            "bed_identical_to_reference": r.get("bed_identical_to_reference"),
            "bed_lines": ref.get("bed_lines"), "chromap_amd_cli_same_files": r.get("chromap_amd", {}).get("cli")}


def cpu_baseline_port(g, args):
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
    n = args.pairs
    rl = args.readlen
    b1 = np.empty(n * rl, np.uint8)
    b2 = np.empty(n * rl, np.uint8)
    o1 = np.empty(n + 1, np.uint32)
    o2 = np.empty(n + 1, np.uint32)
    assert L.cmgpu_download_batch(g.ctx, b1.ctypes.data, o1.ctypes.data, b2.ctypes.data, o2.ctypes.data) == 0
    from chromap_amd.cpus import cpu_budget
    cores = cpu_budget()  # (the affinity mask cut to the container's CPU quota: 16 on the measurement boxes, which report 256 hardware threads)

    def run(m):
        rec = (ol.OraRecord * m)()
        st = ol.OraStats()
        t = time.time()
        kk = O.ora_map_pairs_mt(ctx, cores, m, 0, b1.ctypes.data, o1.ctypes.data, b2.ctypes.data, o2.ctypes.data,
                                C.cast(rec, C.c_void_p), C.byref(st))
        return time.time() - t, kk, rec, st

    m = n
    t_run, kk, rec, st = run(m)
    reps, t_total = 1, t_run
    while t_total < 12.0 and reps < 64:
        t_more, _, _, _ = run(m)
        t_total += t_more
        reps += 1
    t_run = t_total / reps
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


def gen(g, args, seed, pairs=None):
    g.generate_resident(pairs or args.pairs, read_length=args.readlen, frag_min=args.frag_min, frag_max=args.frag_max, sub_rate=0.01, seed=seed,
                        indel_rate=args.indel_rate, hic=args.hic if args.hic >= 0 else None)


SETUP_PASSES = 2
MAPPED = {"pairs": 0}  # pairs this process has mapped so far (the profiler passes divide a kernel's summed counters by it)
SUB_STEP_MAX = 25_000_000  # pairs mapped by one call inside a step (--strong: a rank's share of the total is cut into sub-steps of at most this)
MAX_SLOTS = 8              # cmgpu_swap_resident_batch has eight parking slots


def sub_steps(pairs):
    """a rank's pairs of one step as the sizes of its sub-steps: [pairs] up to SUB_STEP_MAX, else equal parts (the last takes the remainder)"""
    n = max(1, -(-pairs // SUB_STEP_MAX))
    if n > MAX_SLOTS:
        raise SystemExit("--strong: %d pairs per GPU and step need more than %d sub-steps of %d" % (pairs, MAX_SLOTS, SUB_STEP_MAX))
    s = -(-pairs // n)
    return [s] * (n - 1) + [pairs - s * (n - 1)]


def timed_run(g, args, rank, world, dist, seed0, exchange):
    """parks the distinct batches, then warm-up + K timed steps; returns (dt of this rank, stage sums, Stats, mapped).
    A step maps args.pairs pairs: one resident batch (N_SLOTS distinct ones take turns), or -- more than SUB_STEP_MAX pairs,
    --strong at few GPUs -- every one of its sub-steps' batches in turn, all of them distinct and resident"""
    import torch
    from chromap_amd import Stats
    subs = sub_steps(args.pairs)
    n_slots = N_SLOTS if len(subs) == 1 else len(subs)
    for b in range(n_slots):
        gen(g, args, seed0 + rank * 64 + b, subs[b % len(subs)])
        g.swap_resident(b)

    def step(i, stats):
        k, tms = 0, {}
        for sl in ([i % n_slots] if len(subs) == 1 else range(n_slots)):
            g.swap_resident(sl)
            k += g.map_resident(stats)
            MAPPED["pairs"] += subs[sl % len(subs)]
            if exchange:
                g.exchange_step()
            for name, ms in g.timings():
                tms[name] = tms.get(name, 0.0) + ms
            g.swap_resident(sl)
        return k, list(tms.items())

    if exchange:  # a run knows how many pairs it maps: the owner's store is sized once, not doubled on the way
        g.store_reserve(int((SETUP_PASSES + args.warmup + args.steps + 1) * args.pairs * 1.3))
    # set-up, before the W warm-up steps: two untimed passes that let the library SIZE its device buffers (the candidate arrays follow the
    # previous batch's totals, the rescue-hit pool grows to its demand -- a hipFree + hipMalloc of gigabytes, 0.3-0.5 s, that landed in
    # the first timed step of repeat-rich workloads when W was 1).  Reported as "setup_passes"; the K timed steps are untouched.
    for i in range(SETUP_PASSES):
        step(i, Stats())
    for i in range(args.warmup):
        step(i, Stats())
    g.store_clear()
    stage_ms = {}
    st = Stats()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    mapped = 0
    for i in range(args.steps):
        k, tm = step(args.warmup + i, st)
        mapped += k
        for name, ms in tm:
            stage_ms[name] = stage_ms.get(name, 0.0) + ms
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    return dt, stage_ms, st, mapped


GRADED_LOOKUPS_MIN = 100_000_000  # the graded launch of the probe kernel has at least 10^8 lookups
GRADED_LAUNCHES = 30              # its duration is the MEDIAN of this many single launches, each bracketed by HIP events


def probe_source_sha():
    """sha of the sources the probe kernel is compiled from: profiles/probe_traffic.json names the one it was measured on"""
    import hashlib
    h = hashlib.sha256()
    for f in ("cm_kernels.hip", "cm_stages.h", "cm_coop.h", "cm_types.h"):
        h.update(open(os.path.join(ROOT, "chromap_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def s3b_roofline(g, occurrences, pairs, label):
    """SURVEY 8(d)'s second term: 8 B x occurrence entries read, over the time of the candidate stage's kernels (S3b: k_s3b_*) that
    read them -- HIP events around the stage on the mapping stream, of the one-lane pass over the resident batch mapped last"""
    ms = dict(g.timings()).get("s3b_candidates", 0.0)
    if ms <= 0:
        return None
    traffic, src = None, None
    prof = os.path.join(ROOT, "profiles", "s3b_traffic.json")
    if os.path.exists(prof):
        try:
            pj = json.load(open(prof)).get(label)
            if pj and pj.get("source_sha") == probe_source_sha():
                traffic = int(pj["hbm_bytes_per_pair"] * pairs)
                src = ("profiles/s3b_traffic.json: %.1f HBM bytes per pair (FETCH_SIZE + WRITE_SIZE of every k_s3b_* dispatch in separate rocprofv3 --pmc passes "
                       "over %d pairs of this workload, tools/profile_bench.sh; same sources, sha %s, commit %s) x the pairs of this pass"
                       % (pj["hbm_bytes_per_pair"], pj.get("pairs_mapped_in_pmc_pass", 0), pj.get("source_sha"), pj.get("commit")))
            elif pj:
                src = "profiles/s3b_traffic.json is stale (measured on other sources: %s, this run %s): not reported" % (pj.get("source_sha"), probe_source_sha())
        except Exception:
            pass
    ach = 8.0 * occurrences / (ms * 1e-3) / 1e9
    return {"kernels": "k_s3b_candidates + k_s3b_heavy<*> + k_s3b_coop<*> (the candidate stage of one batch)", "bound": "hbm", "achieved": round(ach, 1),
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5), "algorithmic_bytes_per_pass": int(8 * occurrences),
            "occurrences_read": int(occurrences), "pairs": int(pairs), "stage_ms": round(ms, 3), "traffic": traffic, "traffic_source": src,
            "note": "the stage sorts and sweeps the occurrences it reads (merge sort in shared memory): its bound is instruction issue / latency, not "
                    "HBM -- the fraction is reported because SURVEY 8(d) counts these bytes, not as a claim that HBM limits the stage"}


def roofline(g, args, steps):
    """kernel-only measurement of the index probe (HIP events on the launch stream) against the swept random-gather
    ceiling of the same table.  The graded launch looks up the minimizers of ONE resident batch of >= 10^8 lookups (a batch of
    enough pairs is generated and mapped in one piece); its duration is the median of GRADED_LAUNCHES single launches"""
    from chromap_amd import Stats
    import statistics
    # a batch with >= 10^8 minimizers, mapped in one piece so that its hashes are resident together
    per_pair = 15.2 * args.readlen / 50.0
    gp = int(-(-GRADED_LOOKUPS_MIN * 1.03 // per_pair))
    gp = max(args.pairs, -(-gp // 1_000_000) * 1_000_000)
    gen(g, args, 999, gp)
    gst = Stats()
    g.map_resident(gst)
    MAPPED["pairs"] += gp
    gs = gst.as_dict()
    n_mm = gs["num_minimizers"]
    s3b = s3b_roofline(g, gs["occurrences_read"], gp, "headline" if not args.headline_repeats else args.headline_repeats)
    # the file's table (khash layout, load 0.7): its probe steps are SURVEY 8(d)'s algorithmic unit -- what kh_get visits for these lookups
    favg, fps, fhits = C.c_double(0), C.c_uint64(0), C.c_uint64(0)
    file_layout = None
    if not args.graded_probe_only and g.L.cmgpu_probe_bench_variant(g.ctx, n_mm, args.probe_repeat, 1, 2, C.byref(favg), C.byref(fps), C.byref(fhits)) == 0 and favg.value > 0:
        file_layout = {"lookups": int(n_mm), "avg_ms": round(favg.value, 4), "probe_steps": int(fps.value), "hits": int(fhits.value),
                       "GB/s": round(16.0 * fps.value / (favg.value * 1e-3) / 1e9, 1), "buckets": g.get_option("probe_table_buckets") >> max(0, args.probe_table_shift)}
    if args.graded_probe_only:
        # the profiler passes: the file table's probe steps (a property of the data set) are counted by ONE launch of another shape
        # (two lookups per lane: k_probe<2, false>), so that every k_probe<1, false> in the trace is the graded launch
        g.L.cmgpu_probe_bench_variant(g.ctx, n_mm, 1, 2, 2, C.byref(favg), C.byref(fps), C.byref(fhits))
    probe_steps = fps.value if fps.value else gs["probe_steps"]
    alg_bytes = 16.0 * probe_steps  # one 8-B key + one 8-B value per bucket kh_get visits
    avg, ps, hits = C.c_double(0), C.c_uint64(0), C.c_uint64(0)
    probe_only, achieved, probe_ms = None, 0.0, 0.0
    singles = []
    for _ in range(1 if args.graded_probe_only else GRADED_LAUNCHES):
        # (every call: one counted launch, then `repeat` launches between two events; under the profiler one call of probe_repeat launches)
        if g.L.cmgpu_probe_bench(g.ctx, None, n_mm, args.probe_repeat if args.graded_probe_only else 1, C.byref(avg), C.byref(ps), C.byref(hits), None) != 0 or avg.value <= 0:
            break
        singles.append(avg.value)
    if singles:
        med = statistics.median(singles)
        probe_only = {"lookups": int(n_mm), "pairs_of_the_batch": int(gp), "launches_timed": len(singles), "median_ms": round(med, 4),
                      "mean_ms": round(sum(singles) / len(singles), 4), "min_ms": round(min(singles), 4), "max_ms": round(max(singles), 4),
                      "buckets_visited": int(ps.value), "hits": int(hits.value),
                      "table_buckets": g.get_option("probe_table_buckets"),
                      "GB/s": round(alg_bytes / (med * 1e-3) / 1e9, 1), "GB/s_of_buckets_visited": round(16.0 * ps.value / (med * 1e-3) / 1e9, 1),
                      "G_lookups/s": round(n_mm / (med * 1e-3) / 1e9, 2)}
        achieved, probe_ms = probe_only["GB/s"], med
        avg.value = med
        assert hits.value == fhits.value or not fps.value, "the re-hashed table answers differently"
    variants = []
    for u in (() if args.graded_probe_only else (1, 2, 4, 8)):
        for pair in (0, 1):
            a = C.c_double(0)
            if g.L.cmgpu_probe_bench_variant(g.ctx, n_mm, max(3, args.probe_repeat // 2), u, pair, C.byref(a), None, None) == 0 and a.value > 0:
                variants.append({"lookups_per_lane": u, "pair_prefetch": pair, "avg_ms": round(a.value, 4)})
    ng = 1 << 28
    sweep = []
    for loads, width in (((4, 16),) if args.graded_probe_only else ((1, 16), (2, 16), (4, 16), (8, 16), (16, 16), (1, 64), (2, 64), (4, 64), (8, 64))):
        a = C.c_double(0)
        if g.L.cmgpu_gather_sweep(g.ctx, ng, 3, loads, width, C.byref(a)) == 0 and a.value > 0:
            sweep.append({"loads_per_lane": loads, "access_bytes": width, "avg_ms": round(a.value, 4),
                          "G_accesses/s": round(ng / (a.value * 1e-3) / 1e9, 2), "sector_GB/s": round(64.0 * ng / (a.value * 1e-3) / 1e9, 1)})
    best = max(sweep, key=lambda x: x["G_accesses/s"]) if sweep else None
    traffic, sectors_per_lookup, traffic_src = None, None, "no profiles/probe_traffic.json"
    prof = os.path.join(ROOT, "profiles", "probe_traffic.json")
    if os.path.exists(prof):
        try:
            pj = json.load(open(prof))
            # the file names the sources and the launch it was measured on: another kernel text or launch size and it is refused
            if pj.get("source_sha") == probe_source_sha() and pj.get("lookups") == n_mm:
                traffic = pj.get("hbm_bytes_per_launch")
                sectors_per_lookup = pj.get("sectors_per_lookup")
                traffic_src = ("from_profile: profiles/probe_traffic.json (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this kernel on this launch "
                               "-- same sources (sha %s, commit %s), same %d lookups --, tools/profile_bench.sh), not measured in this run"
                               % (pj.get("source_sha"), pj.get("commit"), n_mm))
            else:
                traffic_src = ("profiles/probe_traffic.json is stale (measured on sources %s / %s lookups, this run: %s / %d): not reported"
                               % (pj.get("source_sha"), pj.get("lookups"), probe_source_sha(), n_mm))
        except Exception:
            pass
    shape = {"lookups_per_lane": g.get_option("probe_lookups_per_lane"), "pair_prefetch": g.get_option("probe_pair_prefetch")}
    roof = {"kernel": "k_probe", "kernel_shape": shape, "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            # the same launch counted in the buckets it really read (the re-hashed table's walk is shorter than kh_get's in the file's table)
            "frac_visited": round(16.0 * ps.value / (avg.value * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if avg.value > 0 else None,
            "extra_hbm_bytes_of_rehashed_table": int(16 * g.get_option("probe_table_buckets")) if args.probe_table_shift else 0,
            "traffic": traffic,
            "traffic_source": traffic_src,
            "algorithmic_bytes_per_launch": int(alg_bytes),
            "algorithmic_bytes_cover": "16 B per bucket kh_get visits; the occurrence bytes of SURVEY 8(d) (8 B x occurrences) are read by the candidate "
                                       "stage (S3b: k_s3b_*), not by k_probe, and are not in this figure",
            "launch_ms": round(probe_ms, 4), "launch_ms_is": "median of %d single launches" % len(singles), "probe_only": probe_only,
            "probe_on_file_layout": file_layout,
            "table_note": "the graded launch probes the device's re-hashed copy of the table (cmgpu_set_option probe_table_shift = %d: %d buckets; same "
                          "keys, values, hash and probe sequence, so hit / miss / value of every lookup are the file table's -- asserted here -- and "
                          "fewer buckets are visited); algorithmic bytes = 16 B x the buckets kh_get visits in the FILE's table for the same lookups "
                          "(SURVEY 8(d)); probe_on_file_layout is the same kernel on that table"
                          % (args.probe_table_shift, g.get_option("probe_table_buckets")) if args.probe_table_shift else "the graded launch probes the file's table",
            "probe_variants": variants, "random_gather_sweep": sweep, "random_gather_ceiling": best, "s3b": s3b}
    if best and probe_only:
        # useful: bucket reads of the probe against 16-byte accesses of the ceiling shape;
        # sector: 64-byte sectors the probe fetched (PMC, profiles/probe_traffic.json) against the ceiling's sectors
        roof["useful_frac"] = round((ps.value / (avg.value * 1e-3) / 1e9) / best["G_accesses/s"], 3)  # bucket reads of this launch / ceiling
        if sectors_per_lookup:
            roof["sector_frac"] = round((n_mm * sectors_per_lookup / (avg.value * 1e-3) / 1e9) / best["G_accesses/s"], 3)
        roof["frac_of_measured_gather"] = roof["useful_frac"]
    return roof


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
    ap.add_argument("--frag-min", type=int, default=30)
    ap.add_argument("--frag-max", type=int, default=600)
    ap.add_argument("--indel-rate", type=float, default=0.0, help="1-base indels per base in the synthetic reads (SURVEY 8(d): 0.001 for config 5)")
    ap.add_argument("--repeats", default="32,600,3000,0.02",
                    help="planted repeat families of the second, repeat-bearing workload: families,copies,element_len,divergence ('' = skip it)")
    ap.add_argument("--hic", type=float, default=-1.0, help="Hi-C shaped pairs (mates from independent loci) with this fraction of reads chimeric "
                                                             "across a ligation junction; default: fragments")
    ap.add_argument("--hic-workload", default="150,0.001,0.35,2000000",
                    help="the fourth workload, BASELINE config 5: --preset hic, read length, indel rate, chimeric fraction, pairs per step ('' = skip it)")
    ap.add_argument("--harsh", default="profile:1",
                    help="the third workload's genome: profile:1 = 22.5 %% of the bases repeat-derived -- SINE-like families of ~10^4 copies at "
                         "5-15 %% divergence, LINE-like families of ~360 copies at 1-5 %%, satellite arrays ('' = skip it)")
    ap.add_argument("--harsh-ref-pairs", type=int, default=2_000_000, help="pairs per batch of the reference-binary check on the third / fourth workload (two batches)")
    ap.add_argument("--harsh2", default="profile:2",
                    help="the fourth workload's genome: profile:2 = the same element kinds with 47 %% of the bases repeat-derived, as GRCh38 is "
                         "(SINE-like families of ~19 000 copies, LINE-like of ~830) ('' = skip it)")
    ap.add_argument("--headline-repeats", default="", help="plant repeats in the headline workload's genome too (not the default)")
    ap.add_argument("--skip-cpu", action="store_true")
    ap.add_argument("--skip-extras", action="store_true", help="timed region and roofline only")
    ap.add_argument("--cpu-baseline", choices=["auto", "reference", "port"], default="auto",
                    help="reference: oracle/_ref/chromap itself; port: the oracle restatement with OpenMP; auto: reference if its binary is here")
    ap.add_argument("--probe-repeat", type=int, default=10)
    ap.add_argument("--graded-probe-only", action="store_true",
                    help="roofline section: only the graded launch of k_probe and the gather calibration (the PMC passes of tools/profile_bench.sh: "
                         "their per-kernel averages must not mix launches on different tables or shapes)")
    ap.add_argument("--sam", action="store_true", help="--SAM mode: every reported read is aligned with the banded affine-gap DP "
                                                         "(CIGAR / NM / MD); not the headline metric")
    ap.add_argument("--force-exchange", action="store_true",
                    help="run the RCCL record exchange even with one rank (exercises the N>1 code path on a 1-GPU box)")
    ap.add_argument("--lanes", type=int, default=3, help="ranges of a batch mapped side by side (cmgpu_set_option lanes); measured best for resident batches")
    ap.add_argument("--probe-table-shift", type=int, default=1,
                    help="the pipeline probes a device copy of the index table re-hashed into 2^shift times as many buckets (same keys, values, "
                         "hash and probe sequence: identical lookups, fewer buckets visited); 0: the file's table")
    ap.add_argument("--exchange-lanes", type=int, default=1, help="lanes of a rank that exchanges records (see make_ctx)")
    ap.add_argument("--strong", type=int, default=0, metavar="TOTAL_PAIRS",
                    help="strong scaling: TOTAL_PAIRS read pairs per step over ALL ranks (each maps TOTAL_PAIRS / gpus), e.g. "
                         "--strong 100000000 --steps 1 for BASELINE config 3's fixed 100 M pairs at 1/2/4/8 GPUs; the JSON line then says scaling: strong")
    ap.add_argument("--config3-pairs", type=int, default=100_000_000,
                    help="extra (N = 1): one timed pass over this many DISTINCT resident pairs, BASELINE config 3's size (0 = skip it)")
    ap.add_argument("--option", action="append", default=[], help="name=value for cmgpu_set_option (measurement knobs)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

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
    if args.strong:
        if args.strong % world:
            raise SystemExit("--strong: TOTAL_PAIRS must divide by the number of GPUs")
        args.pairs = args.strong // world
        sub_steps(args.pairs)  # (refuses what the parking slots cannot hold)
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE=%d but --gpus %d: launch with torch.distributed.run --nproc-per-node %d" % (world, args.gpus, args.gpus))
    torch.cuda.set_device(local_rank)
    exchange = world > 1 or args.force_exchange
    dist = None
    if world > 1:
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        if os.environ["MASTER_ADDR"] in ("127.0.0.1", "localhost"):
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(minutes=30))

    from chromap_amd import ChromapGPU, Stats

    def parse_rep(txt):
        if not txt:
            return None
        if txt.startswith("profile:"):
            return txt
        f = txt.split(",")
        return (int(f[0]), int(f[1]), int(f[2]), float(f[3]))

    def make_ctx(rep, preset=None):
        g_ = ChromapGPU(synthetic=(args.genome, args.nseq, 12345, rep), preset=preset or args.preset, device=local_rank,
                        **({"output_format": 1} if args.sam else {}))
        # lanes and the record exchange do not mix well on one GPU (measured: 3 lanes 429 -> 366 M pairs/s with the exchange,
        # 1 lane 404 -> 388): ranks that exchange map their batch in one piece
        g_.set_option("lanes", args.exchange_lanes if exchange else args.lanes)
        if args.probe_table_shift != 1:  # (1 is the library's own default: cmgpu_create* re-hash the table into twice the buckets)
            g_.set_option("probe_table_shift", args.probe_table_shift)
        for o in args.option:
            k_, v_ = o.split("=")
            g_.set_option(k_, int(v_, 0))
        return g_

    t0 = time.time()
    g = make_ctx(parse_rep(args.headline_repeats))
    t_index = time.time() - t0
    if rank == 0:
        log("[bench] synthetic genome + index on device in %.1fs" % t_index)
    if exchange:
        uid = [g.exchange_unique_id() if rank == 0 else None]
        if dist is not None:
            dist.broadcast_object_list(uid, src=0)
        g.exchange_init(uid[0], rank, world)
    dt, stage_ms, st, mapped = timed_run(g, args, rank, world, dist, 1000, exchange)
    info = {"rank": rank, "mapped_pairs": int(mapped), "dt_s": dt}
    if exchange:
        info.update(g.exchange_info())
        info["records_owned"] = info.pop("records_received")
    infos = [info]
    if dist is not None:
        infos = [None] * world
        dist.all_gather_object(infos, info)
        dt = max(x["dt_s"] for x in infos)
    if exchange:
        g.exchange_finalize()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        g.close()
        os.close(json_fd)
        return
    steps = max(1, args.steps)
    subs = sub_steps(args.pairs)
    n_slots = N_SLOTS if len(subs) == 1 else len(subs)
    total_pairs = args.pairs * args.steps * world
    value = total_pairs / dt / 1e6
    s = st.as_dict()
    # the kernel-only probe measurement wants one batch's minimizers resident in one piece
    g.set_option("lanes", 1)
    roof = roofline(g, args, steps)
    g.set_option("lanes", args.lanes)
    post = pcie = cpu = rep_out = harsh_out = harsh2_out = hic_out = cfg3 = None
    # (on the headline genome only: a repeat-rich genome's candidate arrays for 25 M-pair sub-steps do not fit three lanes' worth of HBM)
    if not args.skip_extras and world == 1 and args.config3_pairs and not args.strong and not args.headline_repeats:
        # BASELINE config 3 at its stated size on this one GPU: 100 M DISTINCT pairs, generated on the device and resident, mapped
        # once inside one timed step as sub-steps of <= SUB_STEP_MAX pairs (what --strong 100000000 --gpus 1 runs)
        try:
            import copy
            ca = copy.copy(args)
            ca.pairs, ca.steps, ca.warmup = args.config3_pairs, 1, 0
            cdt, _, cst, cmapped = timed_run(g, ca, 0, 1, None, 2000, False)
            cfg3 = {"distinct_pairs": ca.pairs, "sub_steps": sub_steps(ca.pairs), "value": round(ca.pairs / cdt / 1e6, 4), "unit": "M pairs/s",
                    "ms": round(cdt * 1e3, 2), "mapped_pairs": int(cmapped), "lanes": args.lanes,
                    "note": "one timed pass over all of the pairs (after %d untimed set-up passes over the same batches)" % SETUP_PASSES}
            for b in range(len(sub_steps(ca.pairs))):  # the headline's batches back into their slots for what follows
                gen(g, args, 1000 + b)
                g.swap_resident(b)
        except Exception as e:
            cfg3 = {"error": repr(e)}
    if not args.skip_extras:
        # device-side post-processing, outside the timed region: the records of the four resident
        # batches go to the record store; one call sorts, de-duplicates, filters and renders the BED text in HBM
        try:
            g.store_clear()
            nrec = 0
            for b in range(n_slots):
                g.swap_resident(b)
                g.map_resident(Stats())
                nrec = g.store_append_resident()
                g.swap_resident(b)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            lines, nbytes = g.store_format(0)
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            post = {"records": int(nrec), "bed_lines": int(lines), "text_bytes": int(nbytes),
                    "sort_dedup_format_ms": round((t3 - t2) * 1e3, 3), "M_records/s": round(nrec / (t3 - t2) / 1e6, 1)}
            g.store_clear()
        except Exception as e:
            post = {"error": repr(e)}
        # PCIe-inclusive rate of the host-buffer boundary (cmgpu_map_pairs: H2D reads, map, D2H records);
        # reported beside `value`, never as `value`
        try:
            import numpy as np
            n = subs[0]
            g.swap_resident(0)
            o1 = np.zeros(n + 1, np.uint32)
            o2 = np.zeros(n + 1, np.uint32)
            b1 = np.zeros(n * args.readlen, np.uint8)
            b2 = np.zeros(n * args.readlen, np.uint8)
            assert g.L.cmgpu_download_batch(g.ctx, b1.ctypes.data, o1.ctypes.data, b2.ctypes.data, o2.ctypes.data) == 0
            g.swap_resident(0)
            g.set_option("lanes", 1)  # the copy engine and the lanes' host threads get in each other's way (measured)
            g.map_pairs(b1, o1, b2, o2)  # warm the host-side buffers
            t1 = time.perf_counter()
            _, kk = g.map_pairs(b1, o1, b2, o2)
            t2 = time.perf_counter()
            pcie = {"M pairs/s": round(n / (t2 - t1) / 1e6, 2), "ms": round((t2 - t1) * 1e3, 2), "records": int(kk),
                    "note": "cmgpu_map_pairs on pageable host buffers (upload, map, record download)"}
            if hasattr(g, "map_pairs_pipelined"):
                g.set_option("lanes", 1 if exchange else args.lanes)  # the upload runs on a stream (and hardware queue) of its own
                pcie["pipelined"] = g.map_pairs_pipelined(b1, o1, b2, o2, repeats=10)
                pcie["pipelined"]["lanes"] = 1 if exchange else args.lanes
        except Exception as e:
            pcie = {"error": repr(e)}
        if world == 1 and not args.skip_cpu and not args.sam:
            try:
                if args.cpu_baseline in ("auto", "reference"):
                    g.close()  # the baseline tool builds its own copy of the index on the device
                    g = None
                    cpu = cpu_baseline_reference(args, parse_rep(args.headline_repeats))
                if cpu is None and args.cpu_baseline in ("auto", "port"):
                    if g is None:
                        g = make_ctx(parse_rep(args.headline_repeats))
                    gen(g, args, 1000)
                    g.map_resident(Stats())
                    cpu = cpu_baseline_port(g, args)
            except Exception as e:  # the GPU number must still be reported
                cpu = {"value": None, "unit": "M pairs/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}
        # ---- the same measurement on a genome with planted repeat families: frequent seeds,
        #      multi-mappers and mate rescue as on a real genome; reported beside the headline
        def side_workload(rep, seed0, what, ref_pairs=None):
            """the headline measurement on another genome (timed the same way), with the reference-binary BED check"""
            nonlocal g
            try:
                if g is not None:
                    g.close()
                g = None
                gr = make_ctx(rep)
                rdt, rstage, rst, rmapped = timed_run(gr, args, 0, 1, None, seed0, False)
                rs = rst.as_dict()
                o = {"value": round(args.pairs * args.steps / rdt / 1e6, 4), "unit": "M pairs/s", "ms_per_step": round(rdt / steps * 1e3, 3),
                     "workload": what, "lanes": args.lanes,
                     "counters_per_step": {k: v // steps for k, v in rs.items()},
                     "candidates_per_read": round(rs["num_candidates"] / (2.0 * args.pairs * args.steps), 3),
                     "stage_ms_per_step": {k: round(v / steps, 3) for k, v in rstage.items()},
                     "mapped_pairs_per_step": rmapped // steps}
                try:  # SURVEY 8(d)'s occurrence bytes over the candidate stage's time, one-lane pass over one of the batches
                    gr.set_option("lanes", 1)
                    gr.swap_resident(0)
                    ost = Stats()
                    gr.map_resident(ost)
                    o["roofline_s3b"] = s3b_roofline(gr, ost.as_dict()["occurrences_read"], args.pairs, rep if isinstance(rep, str) else ",".join(str(x) for x in rep))
                    gr.swap_resident(0)
                except Exception as e:
                    o["roofline_s3b"] = {"error": repr(e)}
                gr.close()
                if not args.skip_cpu and args.cpu_baseline in ("auto", "reference"):
                    o["cpu_baseline"] = cpu_baseline_reference(args, rep, " (%s)" % what[:24], seed0=seed0, pairs=ref_pairs)
                return o
            except Exception as e:
                return {"error": repr(e)}

        rep = parse_rep(args.repeats)
        if world == 1 and rep and not args.sam:
            rep_out = side_workload(rep, 3000, "the headline workload on a genome with %d planted repeat families x %d copies of %d bases at %.1f %% "
                                               "divergence (%.1f %% of the genome)" % (rep[0], rep[1], rep[2], rep[3] * 100,
                                                                                       100.0 * rep[0] * rep[1] * rep[2] / args.genome))
        harsh = parse_rep(args.harsh)
        if world == 1 and harsh and not args.sam:
            harsh_out = side_workload(harsh, 5000, "the headline workload on a genome with a mammalian-like repeat landscape (%s): 22.5 %% of the bases "
                                                   "repeat-derived -- SINE-like 300-base elements in 128 families of ~10^4 copies at 5-15 %% divergence, "
                                                   "LINE-like 3-kb elements in 256 families of ~360 copies at 1-5 %%, satellite arrays of 171-base "
                                                   "units" % harsh, ref_pairs=args.harsh_ref_pairs)
        harsh2 = parse_rep(args.harsh2)
        if world == 1 and harsh2 and not args.sam:
            harsh2_out = side_workload(harsh2, 6000, "the headline workload on a genome with the same repeat kinds at GRCh38's share (%s): 47 %% of the bases "
                                                     "repeat-derived -- 40 %% of the 64-kb tiles SINE-like (128 families of ~19 000 copies at 5-15 %%), 28 %% "
                                                     "LINE-like (256 families of ~830 copies at 1-5 %%), 3 %% satellite arrays" % harsh2,
                                       ref_pairs=args.harsh_ref_pairs)
        if world == 1 and args.hic_workload and not args.sam and args.hic < 0:
            # BASELINE config 5: --preset hic, 2 x 150 with 0.1 % indels, Hi-C shaped pairs with chimeric reads (split alignment)
            try:
                import copy
                f = args.hic_workload.split(",")
                ha = copy.copy(args)
                ha.readlen, ha.indel_rate, ha.hic, ha.pairs, ha.preset = int(f[0]), float(f[1]), float(f[2]), int(f[3]), "hic"
                if g is not None:
                    g.close()
                g = None
                gh = make_ctx(None, "hic")
                hdt, hstage, hst, hmapped = timed_run(gh, ha, 0, 1, None, 7000, False)
                hs = hst.as_dict()
                hic_out = {"value": round(ha.pairs * ha.steps / hdt / 1e6, 4), "unit": "M pairs/s", "ms_per_step": round(hdt / steps * 1e3, 3),
                           "workload": "--preset hic, synthetic 2x%d bp Hi-C shaped pairs (mates from independent loci, 60 %% within 1 Mb; %.0f %% of the "
                                       "pairs with a ligation junction inside a read; 1 %% substitutions, %.2f %% 1-base indels), the headline's "
                                       "GRCh38-sized index, %d pairs per step" % (ha.readlen, ha.hic * 100, ha.indel_rate * 100, ha.pairs),
                           "lanes": args.lanes, "counters_per_step": {k: v // steps for k, v in hs.items()},
                           "stage_ms_per_step": {k: round(v / steps, 3) for k, v in hstage.items()}, "mapped_pairs_per_step": hmapped // steps}
                gh.close()
                if not args.skip_cpu and args.cpu_baseline in ("auto", "reference"):
                    hic_out["cpu_baseline"] = cpu_baseline_reference(ha, None, " (hic)", seed0=7000, pairs=500_000)
            except Exception as e:
                hic_out = {"error": repr(e)}
    out = {
        "metric": "M paired reads mapped/s (ATAC preset, GRCh38 index)",
        "value": round(value, 4), "unit": "M pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / steps * 1e3, 3), "higher_is_better": True, "scaling": "strong" if args.strong else "weak", "vs_baseline": None, "setup_passes": SETUP_PASSES,
        "dtype": "u64", "data": "synthetic",
        "config": {"workload": "--preset %s, synthetic 2x%d bp pairs (fragments %d-%d bp, 1%% substitutions%s), "
                               "GRCh38-sized synthetic index (%.2e bases, %d sequences, k=17 w=7%s) resident per GPU, "
                               "%d pairs per GPU per step, %s (%.0f M distinct pairs per GPU)"
                               % (args.preset, args.readlen, args.frag_min, args.frag_max,
                                  ", %.2f%% 1-base indels" % (args.indel_rate * 100) if args.indel_rate else "", args.genome, args.nseq,
                                  ", planted repeats " + args.headline_repeats if args.headline_repeats else "", args.pairs,
                                  "%d distinct batches per GPU resident in HBM taking turns" % N_SLOTS if len(subs) == 1 else
                                  "mapped as %d sub-steps of <= %d pairs, every sub-step a distinct batch resident in HBM" % (len(subs), subs[0]),
                                  (args.pairs * N_SLOTS if len(subs) == 1 else args.pairs) / 1e6),
                   "pairs_per_gpu_per_step": args.pairs, "lanes": args.exchange_lanes if exchange else args.lanes,
                   "parallelism": ("read-shard x%d, records to chromosome owners by device partition + RCCL all-to-all on the library's "
                                   "mapping stream inside every step" % world) if exchange else "single GPU"},
        "roofline": roof, "cpu_baseline": cpu, "repeat_workload": rep_out, "harsh_repeat_workload": harsh_out, "harsh2_repeat_workload": harsh2_out, "hic_workload": hic_out, "config3_full_size_pass": cfg3, "postprocess_on_device": post, "pcie_inclusive": pcie,
        "stage_ms_per_step": {k: round(v / steps, 3) for k, v in stage_ms.items()},
        "stage_ms_note": "HIP events of the calling thread's lane (1 / %d of the batch when lanes > 1; the lanes overlap)" % args.lanes,
        "counters_per_step": {k: v // steps for k, v in s.items()},
        "mapped_pairs_per_step": mapped // steps, "index_build_s": round(t_index, 1),
        "pairs_mapped_in_process": MAPPED["pairs"], "s3b_label": args.headline_repeats or "headline",
        "host": host_info(),
    }
    if exchange:
        out["exchange"] = {"ranks": world, "transport": "RCCL (ncclAllGather of counts + grouped ncclSend/ncclRecv) on the mapping stream",
                           "per_rank": [{k: x[k] for k in ("rank", "mapped_pairs", "records_sent", "records_owned")} for x in infos]}
    result_line = json.dumps(out)
    if g is not None:
        g.close()
    sys.stdout.flush()
    os.write(json_fd, (result_line + "\n").encode())
    os.close(json_fd)


if __name__ == "__main__":
    main()
