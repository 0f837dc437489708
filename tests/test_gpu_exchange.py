"""Multi-GPU record exchange behind the C ABI (cm_exchange.hip) on the GPU box:

* the whole chain HIP map -> device partition by chromosome owner -> exchange -> owner's device record store
  -> device sort / duplicate removal / BED text, with TWO ranks (two processes sharing the one GPU of the
  box; RCCL needs one GPU per rank, so the ranks use the host-staged gloo transport through
  cmgpu_exchange_init_external -- everything but the wire is the product path), against the reference's
  golden BED;
* the library's own RCCL transport with a one-rank communicator (ncclAllGather + ncclSend/ncclRecv to self on
  the mapping stream), bulk and single-cell records;
* sub-batch mapping when a dense intermediate would pass the item limit;
* the measurement helpers (parked batches, probe kernel shapes)."""
import ctypes as C
import os
import socket
import sys

import numpy as np
import pytest

import datasets
import oracle_lib as ol

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _gpu(case, **extra):
    from chromap_amd import ChromapGPU
    meta = datasets.case_meta(case)
    fa, r1, r2 = datasets.case_inputs(case)
    preset, kw = datasets.flags_to_params(meta["chromap_flags"])
    kw.update(extra)
    return ChromapGPU(datasets.case_index(case), fa, preset=preset, **kw), meta, r1, r2


def _worker(rank, world, port, case, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    torch.zeros(1, device="cuda")  # torch's HIP runtime first (tests/conftest.py)
    from chromap_amd import _capi
    from chromap_amd.distributed import HostStagedTransport, owned_rids, shard_batches
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    g, meta, r1, r2 = _gpu(case)
    b1, o1 = ol.read_fastx(r1)
    b2, o2 = ol.read_fastx(r2)
    n = len(o1) - 1
    g.exchange_init_external(HostStagedTransport(g), rank, world)
    shard = 10000  # stands in for the 500000-pair reference batch; keeps the 5000-pair task chunks whole
    mine = shard_batches(n, rank, world, ref_batch=shard)
    rounds = (n + shard * world - 1) // (shard * world)
    sent_total = 0
    for rd in range(rounds):  # the exchange is collective: a rank without a batch takes part with an empty one
        if rd < len(mine):
            lo, hi = mine[rd]
            oo1 = (o1[lo:hi + 1] - o1[lo]).astype(np.uint32)
            oo2 = (o2[lo:hi + 1] - o2[lo]).astype(np.uint32)
            g.upload(b1[o1[lo]:o1[hi]].copy(), oo1, b2[o2[lo]:o2[hi]].copy(), oo2, first_read_id=lo)
        else:
            g.upload(np.zeros(0, np.uint8), np.zeros(1, np.uint32), np.zeros(0, np.uint8), np.zeros(1, np.uint32))
        k = g.map_resident()
        sent, nrecv = g.exchange_step()
        assert sum(sent) == k
        sent_total += k
    info = g.exchange_info()
    assert info["records_sent"] == sent_total and info["world"] == world
    # every record landed on its chromosome's owner: the store holds owned rids only (checked through the text)
    lines, nbytes = g.store_format(_capi.TEXT_BED_PE)
    text = g.store_text()
    own = {g.names[r] for r in owned_rids(list(g.reference_lengths()), rank, world)}
    assert {ln.split(b"\t")[0] for ln in text.splitlines()} <= own
    with open(os.path.join(outdir, "part%d.bed" % rank), "wb") as f:
        f.write(text)
    tot = torch.tensor([info["records_received"], info["records_sent"]], dtype=torch.int64)
    dist.all_reduce(tot)
    assert int(tot[0]) == int(tot[1])  # nothing lost on the way
    g.exchange_finalize()
    g.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("case", ["s1_atac", "s4_atac_q0"])
def test_two_ranks_map_partition_exchange_store_format_equals_golden(case, tmp_path):
    import torch.multiprocessing as mp
    datasets.case_inputs(case)
    datasets.case_index(case)
    mp.spawn(_worker, args=(2, _free_port(), case, str(tmp_path)), nprocs=2, join=True)
    got = b"".join(open(str(tmp_path / ("part%d.bed" % r)), "rb").read() for r in range(2))
    assert got == datasets.case_golden_bed(case)


def _worker_wide(rank, world, port, case, outdir):
    """one of `world` ranks sharing the box's GPU: bulk or single-cell records of a 24-sequence case, shards of 5000 pairs dealt
    in rounds (ranks run out of input at different rounds and keep taking part with empty batches), every record to its
    sequence's owner, the owner's device-side sort / duplicate removal / text"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    torch.zeros(1, device="cuda")
    from chromap_amd import _capi
    from chromap_amd.distributed import HostStagedTransport, owned_rids, shard_batches
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    g, meta, r1, r2 = _gpu(case)
    b1, o1 = ol.read_fastx(r1)
    b2, o2 = ol.read_fastx(r2)
    n = len(o1) - 1
    bc = None
    if datasets.has_barcodes(case):
        bcf, wlf = datasets.case_barcode_inputs(case)
        bc, bcq, bco = ol.read_fastq_qual(bcf)
        g.set_whitelist_file(wlf, int(bco[1] - bco[0]))
        g.compute_barcode_abundance(bc, bco)  # the pre-pass sees the whole input on every rank
    g.exchange_init_external(HostStagedTransport(g), rank, world)
    shard = 5000
    mine = shard_batches(n, rank, world, ref_batch=shard)
    rounds = (n + shard * world - 1) // (shard * world)
    sent_total = 0
    for rd in range(rounds):
        lo, hi = mine[rd] if rd < len(mine) else (0, 0)
        oo1 = (o1[lo:hi + 1] - o1[lo]).astype(np.uint32)
        oo2 = (o2[lo:hi + 1] - o2[lo]).astype(np.uint32)
        if bc is None:
            g.upload(b1[o1[lo]:o1[hi]].copy(), oo1, b2[o2[lo]:o2[hi]].copy(), oo2, first_read_id=lo)
            k = g.map_resident()
        else:
            oob = (bco[lo:hi + 1] - bco[lo]).astype(np.uint32)
            _, k = g.map_pairs_barcoded(b1[o1[lo]:o1[hi]].copy(), oo1, b2[o2[lo]:o2[hi]].copy(), oo2, bc[bco[lo]:bco[hi]].copy(),
                                        bcq[bco[lo]:bco[hi]].copy(), oob, first_read_id=lo)
        sent, nrecv = g.exchange_step()
        assert sum(sent) == k
        sent_total += k
    info = g.exchange_info()
    assert info["records_sent"] == sent_total and info["world"] == world
    if bc is None:
        g.store_format(_capi.TEXT_BED_PE)
    else:
        g.store_format(_capi.TEXT_BED_PE_BC, barcode_length=g.barcode_length)
    text = g.store_text()
    own = {g.names[r] for r in owned_rids(list(g.reference_lengths()), rank, world)}
    assert {ln.split(b"\t")[0] for ln in text.splitlines()} <= own
    with open(os.path.join(outdir, "part%d.bed" % rank), "wb") as f:
        f.write(text)
    tot = torch.tensor([info["records_received"], info["records_sent"], 1 if len(text) == 0 else 0], dtype=torch.int64)
    dist.all_reduce(tot)
    assert int(tot[0]) == int(tot[1])
    with open(os.path.join(outdir, "empty%d" % rank), "w") as f:
        f.write(str(int(tot[2])))
    g.exchange_finalize()
    g.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [4, 8])
@pytest.mark.parametrize("case", ["s5_atac_24chr_q0", "b4_bulk_level_bc_24chr_q0"])
def test_four_and_eight_ranks_equal_golden(case, world, tmp_path):
    """SURVEY 8(e) at the widths the driver's scaling run uses: 24 sequences over 4 / 8 owners, ranks that run out of input
    early, bulk records and single-cell records with duplicate removal at bulk level (its end-of-output MAPQ rule is applied by
    the last rank that owns records); the concatenated sections equal the reference's BED"""
    import torch.multiprocessing as mp
    datasets.case_inputs(case)
    datasets.case_index(case)
    mp.spawn(_worker_wide, args=(world, _free_port(), case, str(tmp_path)), nprocs=world, join=True)
    got = b"".join(open(str(tmp_path / ("part%d.bed" % r)), "rb").read() for r in range(world))
    assert got == datasets.case_golden_bed(case)


@pytest.mark.parametrize("case", ["s1_atac", "s3_chip"])
def test_rccl_exchange_one_rank_equals_golden(case):
    """the library's RCCL transport: communicator of one rank, collectives issued on the mapping stream"""
    from chromap_amd import _capi
    g, meta, r1, r2 = _gpu(case)
    b1, o1 = ol.read_fastx(r1)
    b2, o2 = ol.read_fastx(r2)
    g.exchange_init(g.exchange_unique_id(), 0, 1)
    n = len(o1) - 1
    half = n // 2 // 5000 * 5000
    tot = 0
    for lo, hi in ((0, half), (half, n)):  # two rounds: the second receive lands behind the first in the store
        oo1 = (o1[lo:hi + 1] - o1[lo]).astype(np.uint32)
        oo2 = (o2[lo:hi + 1] - o2[lo]).astype(np.uint32)
        g.upload(b1[o1[lo]:o1[hi]].copy(), oo1, b2[o2[lo]:o2[hi]].copy(), oo2, first_read_id=lo)
        k = g.map_resident()
        sent, nrecv = g.exchange_step()
        assert sent == [k] and nrecv == k
        tot += k
    g.store_format(_capi.TEXT_BED_PE)
    assert g.store_text() == datasets.case_golden_bed(case)
    assert g.exchange_info()["records_received"] == tot
    g.exchange_finalize()
    g.close()


def test_rccl_exchange_barcoded_records():
    """single-cell records travel as 32-byte {record, barcode} entries and are split into the store on arrival"""
    from chromap_amd import _capi
    case = "b1_atac_bc"
    g, meta, r1, r2 = _gpu(case)
    bcf, wlf = datasets.case_barcode_inputs(case)
    b1, o1 = ol.read_fastx(r1)
    b2, o2 = ol.read_fastx(r2)
    bc, bcq, bco = ol.read_fastq_qual(bcf)
    g.set_whitelist_file(wlf, int(bco[1] - bco[0]))
    g.compute_barcode_abundance(bc, bco)
    g.exchange_init(g.exchange_unique_id(), 0, 1)
    rec, k = g.map_pairs_barcoded(b1, o1, b2, o2, bc, bcq, bco)
    sent, nrecv = g.exchange_step()
    assert sent == [k] and nrecv == k
    g.store_format(_capi.TEXT_BED_PE_BC, barcode_length=g.barcode_length)
    assert g.store_text() == datasets.case_golden_bed(case)
    g.close()


def test_owner_table_follows_chr_order():
    """with --chr-order the records carry ranks: ownership is decided on the lengths in rank order"""
    from chromap_amd.distributed import owner_table
    case = "s3_chip"
    g, meta, r1, r2 = _gpu(case)
    names = list(g.names)
    g.set_chr_order(list(reversed(names)))
    lens = list(g.reference_lengths())  # in rank order
    for world in (1, 2, 3):
        assert g.exchange_owner_table(world) == [int(x) for x in owner_table(lens, world)]
    g.close()


def test_item_limit_maps_in_sub_batches():
    """a batch whose hit / candidate arrays would pass the item limit is mapped in sub-batches cut on
    reference-batch boundaries; records and counters are those of the one-piece run"""
    from chromap_amd import Stats
    case = "s4_atac_q0"  # repeat families: multi-mappers, so the per-chunk sampling is exercised as well
    out = []
    # 20000 pairs of 50 bases: a range of n pairs sizes its minimizer arrays at 30 n entries, so 450000 cuts the batch into
    # two ranges of 10000 pairs and 250000 into four of 5000 (the reference batch here; chunks of 5000 pairs either way)
    for limit in (0, 450000, 250000):
        g, meta, r1, r2 = _gpu(case, read_batch_size=5000)
        if limit:
            g.set_option("item_limit", limit)
        b1, o1 = ol.read_fastx(r1)
        b2, o2 = ol.read_fastx(r2)
        st = Stats()
        g.upload(b1, o1, b2, o2)
        k = g.map_resident(st)
        rec, k2 = g.download_records(len(o1) - 1)
        assert k2 == k
        raw = bytes(rec)[:k * 24]
        out.append((sorted(raw[i * 24:(i + 1) * 24] for i in range(k)), {x: v for x, v in st.as_dict().items() if x != "probe_steps"}))
        g.close()
    assert out[0][0] == out[1][0] == out[2][0]
    assert out[0][1] == out[1][1] == out[2][1]
    # a single reference batch that cannot fit is an error, not a wrap-around
    from chromap_amd.mapper import ChromapError
    g, meta, r1, r2 = _gpu(case)
    g.set_option("item_limit", 1000)
    b1, o1 = ol.read_fastx(r1)
    b2, o2 = ol.read_fastx(r2)
    g.upload(b1, o1, b2, o2)
    with pytest.raises(ChromapError):
        g.map_resident()
    g.close()


def test_parked_batches_and_probe_shapes():
    """cmgpu_swap_resident_batch keeps distinct batches resident; every shape of the probe kernel gives kh_get's results"""
    from chromap_amd import ChromapGPU, Stats
    g = ChromapGPU(synthetic=(2_000_000, 4, 99), preset="atac")
    n = 20000
    want = []
    for b in range(3):
        g.generate_resident(n, read_length=50, frag_min=30, frag_max=500, sub_rate=0.01, seed=50 + b)
        k = g.map_resident(Stats())
        rec, _ = g.download_records(n)
        want.append(bytes(rec)[:k * 24])
        g.swap_resident(b)
    assert len({w for w in want}) == 3
    ref_counts = None
    for u in (1, 2, 4, 8):
        for pair in (0, 1):
            g.set_option("probe_lookups_per_lane", u)
            g.set_option("probe_pair_prefetch", pair)
            for b in (2, 0, 1):
                g.swap_resident(b)
                st = Stats()
                k = g.map_resident(st)
                rec, _ = g.download_records(n)
                assert bytes(rec)[:k * 24] == want[b], (u, pair, b)
                g.swap_resident(b)
            # the kernel alone on the last batch's minimizers: same buckets visited, same hits
            a, ps, hits = C.c_double(0), C.c_uint64(0), C.c_uint64(0)
            n_mm = st.num_minimizers
            assert g.L.cmgpu_probe_bench_variant(g.ctx, n_mm, 1, u, pair, C.byref(a), C.byref(ps), C.byref(hits)) == 0
            if ref_counts is None:
                ref_counts = (ps.value, hits.value)
                assert ps.value == st.probe_steps
            assert (ps.value, hits.value) == ref_counts, (u, pair)
    # gather sweep entry points answer for every shape bench.py asks for
    for loads, width in ((1, 16), (16, 16), (1, 64), (8, 64)):
        a = C.c_double(0)
        assert g.L.cmgpu_gather_sweep(g.ctx, 1 << 16, 1, loads, width, C.byref(a)) == 0 and a.value > 0
    g.close()


def test_synthetic_repeats_and_indels_match_oracle(tmp_path):
    """planted repeat families + reads with 1-base indels (bench.py's second workload) against the oracle"""
    from chromap_amd import ChromapGPU
    from test_gpu_synthetic import _export_fasta, _tuples
    g = ChromapGPU(synthetic=(3_000_000, 5, 4242, (4, 40, 1500, 0.02)), preset="atac")
    fa = str(tmp_path / "rep.fa")
    _export_fasta(g, fa)
    n, readlen = 30000, 50
    g.generate_resident(n, read_length=readlen, frag_min=30, frag_max=600, sub_rate=0.01, seed=11, indel_rate=0.002)
    b1 = np.zeros(n * readlen, np.uint8)
    b2 = np.zeros(n * readlen, np.uint8)
    o1 = np.zeros(n + 1, np.uint32)
    o2 = np.zeros(n + 1, np.uint32)
    assert g.L.cmgpu_download_batch(g.ctx, b1.ctypes.data, o1.ctypes.data, b2.ctypes.data, o2.ctypes.data) == 0
    k = g.map_resident()
    rec, k2 = g.download_records(n)
    o = ol.Oracle(None, fa, ol.params("atac"))
    orec, ok, ost, _ = o.map_pairs(b1, o1, b2, o2)
    assert ok == k == k2
    assert _tuples(rec, k, True) == _tuples(orec, ok, False)
    s = g.stats.as_dict()
    assert s["num_multi_mappers"] > 0  # the repeats do what they are planted for
    assert s["num_candidates"] == ost.as_dict()["num_candidates"]
    g.close()
    o.close()


def test_lanes_map_ranges_side_by_side():
    """one batch cut on reference-batch boundaries into ranges that are mapped concurrently (own streams and
    intermediates): records and counters are those of the one-piece run"""
    from chromap_amd import ChromapGPU, Stats
    g = ChromapGPU(synthetic=(3_000_000, 5, 77, (2, 40, 1200, 0.02)), preset="atac", read_batch_size=300000)
    n = 1_200_000
    g.generate_resident(n, read_length=50, frag_min=30, frag_max=500, sub_rate=0.01, seed=5)
    out = []
    for lanes in (1, 2, 3, 4):
        g.set_option("lanes", lanes)
        st = Stats()
        k = g.map_resident(st)
        rec, k2 = g.download_records(n)
        assert k2 == k
        a = np.frombuffer(bytes(rec)[:k * 24], dtype=np.uint8).reshape(k, 24)
        key = a.view(np.uint32)[:, 0]
        out.append((a[np.argsort(key, kind="stable")].tobytes(), {x: v for x, v in st.as_dict().items()}))
    for i in (1, 2, 3):
        assert out[i][0] == out[0][0], i
        assert out[i][1] == out[0][1], i
    assert out[0][1]["num_multi_mappers"] > 0
    g.close()


def test_pipelined_submit_equals_map_pairs():
    """cmgpu_submit_pairs / cmgpu_map_submitted with page-locked buffers: two batches in the queue, records (compacted on the
    device, pair order kept) equal to cmgpu_map_pairs'"""
    from chromap_amd import ChromapGPU, Stats
    case = "s1_atac"
    fa, r1, r2 = datasets.case_inputs(case)
    g = ChromapGPU(datasets.case_index(case), fa, preset="atac")
    b1, o1 = ol.read_fastx(r1)
    b2, o2 = ol.read_fastx(r2)
    n = len(o1) - 1
    half = n // 2
    rec, k = g.map_pairs(b1, o1, b2, o2)
    want = bytes(rec)[:k * 24]
    parts = []
    for lo, hi in ((0, half), (half, n)):
        pb1, pb2 = g.host_array(int(o1[hi] - o1[lo]), np.uint8), g.host_array(int(o2[hi] - o2[lo]), np.uint8)
        po1, po2 = g.host_array(hi - lo + 1, np.uint32), g.host_array(hi - lo + 1, np.uint32)
        pb1[:] = b1[o1[lo]:o1[hi]]; pb2[:] = b2[o2[lo]:o2[hi]]
        po1[:] = o1[lo:hi + 1] - o1[lo]; po2[:] = o2[lo:hi + 1] - o2[lo]
        parts.append((pb1, po1, pb2, po2, lo, g.host_array((hi - lo) * 24, np.uint8), hi - lo))
    # uniquely mapped pairs do not depend on where a batch is cut (multi-mappers draw from a per-chunk generator)
    g.submit_pairs(*parts[0][:4], first_read_id=parts[0][4])
    g.submit_pairs(*parts[1][:4], first_read_id=parts[1][4])
    with pytest.raises(Exception):
        g.submit_pairs(*parts[0][:4])  # two are waiting already
    got = b""
    for p in parts:
        kk = g.map_submitted(p[5], p[6], Stats())
        got += p[5][:kk * 24].tobytes()
    assert got == want  # half is a multiple of the 5000-pair chunk: even the multi-mappers agree
    with pytest.raises(Exception):
        g.map_submitted()
    # the same with the record downloads left running (cmgpu_map_submitted_async / cmgpu_records_wait): both pending at once
    for p in parts:
        p[5][:] = 0
    g.submit_pairs(*parts[0][:4], first_read_id=parts[0][4])
    g.submit_pairs(*parts[1][:4], first_read_id=parts[1][4])
    with pytest.raises(Exception):
        g.records_wait()  # nothing pending yet
    g.map_submitted_async(parts[0][5], parts[0][6], Stats())
    g.map_submitted_async(parts[1][5], parts[1][6], Stats())
    k0 = g.records_wait()
    k1 = g.records_wait()
    assert parts[0][5][:k0 * 24].tobytes() + parts[1][5][:k1 * 24].tobytes() == want
    with pytest.raises(Exception):
        g.records_wait()
    # a synchronous call after it still works (and waits for nothing that is not there)
    rec2, k2 = g.map_pairs(b1, o1, b2, o2)
    assert bytes(rec2)[:k2 * 24] == want
    g.close()


def _n_gpus():
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:
        return 0


@pytest.mark.parametrize("world", [1, pytest.param(2, marks=pytest.mark.skipif(_n_gpus() < 2, reason="needs two GPUs (two real RCCL ranks)"))])
def test_bench_ranks_exchange_every_record(world):
    """bench.py launched the way the driver launches it (torch.distributed.run for two ranks; one rank with --force-exchange on a
    1-GPU box): each rank maps its own batches and the records change hands by owner chromosome over RCCL Send/Recv inside every
    step; what the ranks own after the run is what they sent, and the value is all ranks' pairs over the slowest rank's time"""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    args = [os.path.join(root, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1", "--skip-extras", "--skip-cpu", "--genome", "400000000",
            "--pairs", "1000000"]
    if world == 1:
        cmd = [sys.executable] + args + ["--force-exchange"]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port",
               str(_free_port())] + args
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=900)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-2000:]
    j = json.loads(p.stdout.decode().strip().splitlines()[-1])
    assert j["n_gpus"] == world and j["scaling"] == "weak"
    ex = j["exchange"]
    assert ex["ranks"] == world and len(ex["per_rank"]) == world
    sent = sum(r["records_sent"] for r in ex["per_rank"])
    owned = sum(r["records_owned"] for r in ex["per_rank"])
    assert sent > 0 and owned == sent  # every record a rank produced arrived at its owner (a rank also 'sends' to itself)
    assert all(r["mapped_pairs"] > 0.9 * 1000000 * 3 for r in ex["per_rank"])
    assert abs(j["value"] - world * 1000000 / (j["ms_per_step"] * 1e-3) / 1e6) < 0.01 * j["value"]
