#!/usr/bin/env python3
"""Times the pieces of bench.py's multi-GPU record exchange on one rank (partition by owner, the two
all_to_all_single calls, the append to the device-side store).  Run under torch.distributed.run."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist


def main():
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    lr = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(lr)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", lr))
    from chromap_amd import ChromapGPU, Stats
    from chromap_amd.distributed import RecordExchange
    n = 4_000_000
    g = ChromapGPU(synthetic=(400_000_000, 24, 12345), preset="atac", device=lr)
    g.generate_resident(n, read_length=50, frag_min=30, frag_max=600, sub_rate=0.01, seed=1)
    ex = RecordExchange(n, torch.device("cuda", lr))
    g.map_resident(Stats())
    counts = (C.c_uint64 * world)()
    t = {"partition": 0.0, "partition_again": 0.0, "all_to_all": 0.0, "append": 0.0, "map": 0.0}
    reps = 10
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        g.map_resident(Stats())
        torch.cuda.synchronize(); t1 = time.perf_counter()
        assert g.L.cmgpu_records_partition(g.ctx, world, C.c_void_p(ex.send.data_ptr()), n, counts) == 0
        torch.cuda.synchronize(); t2 = time.perf_counter()
        assert g.L.cmgpu_records_partition(g.ctx, world, C.c_void_p(ex.send.data_ptr()), n, counts) == 0
        t["partition_again"] += time.perf_counter() - t2
        t2 = time.perf_counter()
        nrecv = ex.all_to_all(list(counts))
        torch.cuda.synchronize(); t3 = time.perf_counter()
        g.store_append(ex.recv.data_ptr(), nrecv, on_device=True)
        torch.cuda.synchronize(); t4 = time.perf_counter()
        g.store_clear()
        t["map"] += t1 - t0; t["partition"] += t2 - t1; t["all_to_all"] += t3 - t2; t["append"] += t4 - t3
    if rank == 0:
        print({k: round(v / reps * 1e3, 3) for k, v in t.items()}, "ms per step, records", nrecv)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
