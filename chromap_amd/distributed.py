"""Multi-GPU record exchange: read pairs shard across ranks with no
data-path collective; the only exchange is one all-to-all of the fixed-size 24-byte records,
each routed to the rank that owns its chromosome, so that every rank sorts + PCR-dedups the
chromosome range it owns (on the device: cmgpu_records_partition -> all_to_all ->
cmgpu_store_append -> cmgpu_store_format).  Every record crosses xGMI once: (N-1)/N of a rank's
records leave it, against N-1 copies of everything with an all-gather.  torch.distributed is
plumbing here ("nccl" == RCCL over xGMI on ROCm, "gloo" in the CPU tests)."""
import numpy as np

RECORD_BYTES = 24
REC_DTYPE = np.dtype([("read_id", "<u4"), ("rid", "<u4"), ("fragment_start", "<u4"), ("fragment_length", "<u2"),
                      ("mapq", "u1"), ("direction", "u1"), ("is_unique", "u1"), ("num_dups", "u1"),
                      ("positive_alignment_length", "<u2"), ("negative_alignment_length", "<u2"), ("pad", "<u2")])
assert REC_DTYPE.itemsize == RECORD_BYTES


def shard_range(n_items, rank, world):
    """contiguous slice [lo, hi) of n_items owned by rank; slices start on multiples of
    `align` pairs when possible so reservoir-sampling chunks stay whole"""
    lo = (n_items * rank) // world
    hi = (n_items * (rank + 1)) // world
    return lo, hi


def shard_batches(n_pairs, rank, world, ref_batch=500000):
    """Deal whole reference batches (500000 pairs, chromap.h:182) round-robin so that every
    shard starts on a batch boundary and multi-mapper sampling reproduces the single-process
    result.  Returns the list of (lo, hi) this rank maps."""
    out = []
    b = 0
    i = 0
    while b < n_pairs:
        e = min(n_pairs, b + ref_batch)
        if i % world == rank:
            out.append((b, e))
        b = e
        i += 1
    return out


def owner_table(lengths, world):
    """owner rank of every sequence -- the twin of cmgpu_exchange_owner_table: contiguous rid ranges whose largest
    total length B is minimal (B by bisection on the greedy in-order packing into bins of B bases, then that packing)"""
    lengths = [int(x) for x in lengths]

    def bins(b):
        p, cur = 1, 0
        for ln in lengths:
            if cur + ln > b:
                p, cur = p + 1, ln
            else:
                cur += ln
        return p

    lo, hi = (max(lengths) if lengths else 0), sum(lengths)
    while lo < hi:
        mid = lo + (hi - lo) // 2
        if bins(mid) <= world:
            hi = mid
        else:
            lo = mid + 1
    out, k, cur = [], 0, 0
    for ln in lengths:
        if cur + ln > lo and cur > 0:
            k, cur = k + 1, 0
        cur += ln
        out.append(min(k, world - 1))
    return np.asarray(out, dtype=np.int64)


def owned_rids(lengths, rank, world):
    """chromosome ownership for the final sort/dedup: rid-major output order means owners
    write disjoint, ordered sections"""
    return [r for r, k in enumerate(owner_table(lengths, world)) if k == rank]


def partition_by_owner(records, lengths, world):
    """host twin of cmgpu_records_partition: (records grouped by owner rank, counts[world])"""
    own = owner_table(lengths, world)[records["rid"].astype(np.int64)] if len(records) else np.zeros(0, np.int64)
    order = np.argsort(own, kind="stable")
    return records[order], np.bincount(own, minlength=world).astype(np.int64)


class HostStagedTransport:
    """exchange transport for cmgpu_exchange_init_external over any torch.distributed group whose backend
    works on host tensors (gloo): device buffers are staged through host memory with hipMemcpy.  Used where
    RCCL cannot run -- several ranks sharing one GPU in the tests; the product path is the library's own RCCL."""

    def __init__(self, mapper, group=None):
        import ctypes as C
        import torch
        import torch.distributed as dist
        self.C, self.torch, self.dist, self.group = C, torch, dist, group
        self.world = dist.get_world_size(group)
        self.g = mapper  # copies go through the library's own HIP runtime (cmgpu_memcpy)

    def allgather_counts(self, mine):
        t = self.torch.tensor(mine, dtype=self.torch.int64)
        out = [self.torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t, group=self.group)
        return [o.tolist() for o in out]

    def alltoallv(self, send_dev, send_counts, recv_dev, recv_counts, record_bytes):
        torch, C = self.torch, self.C
        ns, nr = sum(send_counts) * record_bytes, sum(recv_counts) * record_bytes
        send = torch.zeros(max(1, ns), dtype=torch.uint8)
        recv = torch.zeros(max(1, nr), dtype=torch.uint8)
        if ns and self.g.L.cmgpu_memcpy(self.g.ctx, C.c_void_p(send.data_ptr()), C.c_void_p(send_dev), ns, 2) != 0:
            raise RuntimeError("device -> host copy failed")
        self.dist.all_to_all_single(recv[:nr], send[:ns], [c * record_bytes for c in recv_counts],
                                    [c * record_bytes for c in send_counts], group=self.group)
        if nr and self.g.L.cmgpu_memcpy(self.g.ctx, C.c_void_p(recv_dev), C.c_void_p(recv.data_ptr()), nr, 1) != 0:
            raise RuntimeError("host -> device copy failed")


class RecordExchange:
    """exchange of per-rank record buffers.  send: uint8 tensor [capacity*24] on the device the
    process group works with (all_to_all: grouped by destination rank); recv: world * capacity records."""

    def __init__(self, capacity, device, group=None):
        import torch
        import torch.distributed as dist
        self.torch = torch
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group)
        self.capacity = capacity
        self.send = torch.zeros(capacity * RECORD_BYTES, dtype=torch.uint8, device=device)
        self.recv = torch.zeros(self.world * capacity * RECORD_BYTES, dtype=torch.uint8, device=device)
        self.cnt = torch.zeros(1, dtype=torch.int64, device=device)
        self.cnts = torch.zeros(self.world, dtype=torch.int64, device=device)

    def all_gather(self, count):
        self.cnt[0] = count
        self.dist.all_gather_into_tensor(self.cnts, self.cnt, group=self.group)
        self.dist.all_gather_into_tensor(self.recv, self.send, group=self.group)
        return self.cnts

    def all_to_all(self, send_counts):
        """send buffer holds sum(send_counts) records grouped by destination rank; returns the number of
        records received (they sit at the front of self.recv, grouped by source rank)"""
        torch, dist = self.torch, self.dist
        sc = torch.tensor([int(x) for x in send_counts], dtype=torch.int64, device=self.send.device)
        rc = torch.zeros(self.world, dtype=torch.int64, device=self.send.device)
        dist.all_to_all_single(rc, sc, group=self.group)
        rcv = [int(x) for x in rc.cpu().tolist()]
        in_split = [int(x) * RECORD_BYTES for x in send_counts]
        out_split = [x * RECORD_BYTES for x in rcv]
        if sum(out_split) > self.recv.numel():
            raise RuntimeError("receive buffer too small for the records owned by this rank")
        dist.all_to_all_single(self.recv[:sum(out_split)], self.send[:sum(in_split)], out_split, in_split, group=self.group)
        if self.recv.is_cuda:
            # the caller hands recv to the C ABI (its own HIP stream) next: the collective must have landed
            torch.cuda.current_stream(self.recv.device).synchronize()
        self.n_recv = sum(rcv)
        return self.n_recv

    def received_records(self):
        raw = self.recv[:self.n_recv * RECORD_BYTES].cpu().numpy()
        return np.frombuffer(raw.tobytes(), dtype=REC_DTYPE)

    def gathered_records(self):
        """numpy structured array of all ranks' records (host copy)"""
        cnts = self.cnts.cpu().numpy()
        raw = self.recv.cpu().numpy()
        parts = []
        for r in range(self.world):
            seg = raw[r * self.capacity * RECORD_BYTES:(r * self.capacity + int(cnts[r])) * RECORD_BYTES]
            parts.append(np.frombuffer(seg.tobytes(), dtype=REC_DTYPE))
        return np.concatenate(parts) if parts else np.zeros(0, REC_DTYPE)
