"""Multi-GPU record exchange: read pairs shard across ranks with no
data-path collective; the only exchange is one all-gather of the fixed-size 24-byte records
so that every rank can sort + PCR-dedup the chromosome range it owns.  torch.distributed is
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


def owned_rids(n_seq, rank, world):
    """chromosome ownership for the final sort/dedup: rid-major output order means owners
    write disjoint, ordered sections"""
    return [r for r in range(n_seq) if (r * world) // n_seq == rank]


class RecordExchange:
    """all-gather of per-rank record buffers.  send: uint8 tensor [capacity*24] on the
    device the process group works with; count: number of valid records in it."""

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

    def gathered_records(self):
        """numpy structured array of all ranks' records (host copy)"""
        cnts = self.cnts.cpu().numpy()
        raw = self.recv.cpu().numpy()
        parts = []
        for r in range(self.world):
            seg = raw[r * self.capacity * RECORD_BYTES:(r * self.capacity + int(cnts[r])) * RECORD_BYTES]
            parts.append(np.frombuffer(seg.tobytes(), dtype=REC_DTYPE))
        return np.concatenate(parts) if parts else np.zeros(0, REC_DTYPE)
