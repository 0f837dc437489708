"""world_size-2 (gloo, CPU) run of the multi-GPU flow: shard pairs -> map each shard ->
all-to-all of the 24-byte records to their chromosome owners -> owners sort + dedup -> BED sections
in rid order.
The mapping itself runs through tests/hostemu (the product's stage functions compiled for the
host); the sharding / exchange / ownership code is chromap_amd/distributed.py, the one the GPU
path uses with the nccl (RCCL) backend."""
import ctypes as C
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, case, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import datasets
    import hostemu_lib as he
    import oracle_lib as ol
    from chromap_amd import _capi
    from chromap_amd.distributed import REC_DTYPE, RecordExchange, owned_rids, partition_by_owner, shard_batches
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    meta = datasets.case_meta(case)
    fa, r1, r2 = datasets.case_inputs(case)
    idx = datasets.case_index(case)
    preset, kw = datasets.flags_to_params(meta["chromap_flags"])
    p = he.params(preset, **kw)
    h = he.HostEmu(idx, fa, p)
    b1, o1 = ol.read_fastx(r1)
    b2, o2 = ol.read_fastx(r2)
    n = len(o1) - 1
    shard = 10000  # stands in for the 500000-pair reference batch; keeps the 5000-pair task chunks whole
    mine = []
    for lo, hi in shard_batches(n, rank, world, ref_batch=shard):
        rec, k, _, _ = h.map_pairs(b1, o1[lo:hi + 1], b2, o2[lo:hi + 1], first_read_id=lo)
        mine.append(np.frombuffer(bytes(rec)[: k * 24], dtype=REC_DTYPE).copy())
    mine = np.concatenate(mine) if mine else np.zeros(0, REC_DTYPE)
    nseq = len(h.names)
    lengths = [int(h.ref.lengths[i]) for i in range(nseq)]
    ex = RecordExchange(n, torch.device("cpu"))
    grouped, counts = partition_by_owner(mine, lengths, world)
    raw = torch.from_numpy(grouped.view(np.uint8).copy())
    ex.send[: raw.numel()] = raw
    ex.all_to_all(counts)
    sel = ex.received_records().copy()
    # owner-side sort + dedup + BED for the chromosomes this rank owns
    own = set(owned_rids(lengths, rank, world))
    assert set(np.unique(sel["rid"]).tolist()) <= own
    buf = (C.c_uint8 * max(1, sel.nbytes)).from_buffer_copy(sel.tobytes() if sel.nbytes else b"\0")
    out = os.path.join(outdir, "part%d.bed" % rank)
    names = (C.c_char_p * nseq)(*h.names)
    h.L.cmgpu_write_bed_pe(names, nseq, C.byref(p), C.cast(buf, C.c_void_p), len(sel), out.encode())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("case", ["s1_atac", "s4_atac_q0"])
def test_two_rank_shard_exchange_dedup_matches_golden(case, tmp_path):
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import datasets
    datasets.case_inputs(case)
    datasets.case_index(case)
    port = _free_port()
    mp.spawn(_worker, args=(2, port, case, str(tmp_path)), nprocs=2, join=True)
    got = b"".join(open(str(tmp_path / ("part%d.bed" % r)), "rb").read() for r in range(2))
    assert got == datasets.case_golden_bed(case)


def test_shard_helpers():
    from chromap_amd.distributed import owned_rids, shard_batches
    assert shard_batches(1_200_000, 0, 2) == [(0, 500000), (1000000, 1200000)]
    assert shard_batches(1_200_000, 1, 2) == [(500000, 1000000)]
    from chromap_amd.distributed import owner_table
    # equal lengths: the plain rid * world / n_seq split
    allr = sorted(sum((owned_rids([1000] * 24, r, 8) for r in range(8)), []))
    assert allr == list(range(24))
    assert owned_rids([1000] * 24, 0, 8) == [0, 1, 2]
    # GRCh38-like lengths: owners are contiguous rid ranges of nearly equal total length
    grch38 = [248, 242, 198, 190, 181, 171, 159, 145, 138, 133, 135, 133, 114, 107, 102, 90, 83, 80, 58, 64, 46, 50, 156, 57]
    own = owner_table(grch38, 8)
    assert list(own) == sorted(own) and own[0] == 0 and own[-1] == 7 and set(own) == set(range(8))
    assert list(owner_table([5, 5, 5], 8)) == [0, 1, 2]
    share = [sum(l for l, k in zip(grch38, own) if k == r) / sum(grch38) for r in range(8)]
    assert max(share) < 0.155 and min(share) > 0.08  # rid * world / n_seq gave rank 0 chr1-3 = 22 %
    assert list(owner_table(grch38, 1)) == [0] * 24
