"""The communication plan of one record-exchange round (cmgpu_exchange_plan, the function cmgpu_exchange_step posts its RCCL
calls from), replayed on the CPU for every rank of a world: the rounds cannot deadlock -- every rank calls the same collectives
in the same order whatever it holds, and inside the one grouped send / receive every send meets exactly one receive of the same
size, also when ranks are empty, own nothing or send to nobody.  No GPU, no RCCL."""
import ctypes as C
import random

import pytest

from chromap_amd import _capi

EX_ALLGATHER_COUNTS, EX_ALLGATHER_STATUS, EX_COPY_SELF, EX_SEND, EX_RECV = range(5)


class Op(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("peer", C.c_uint32), ("records", C.c_uint64)]


def plan(L, rank, world, matrix):
    flat = (C.c_uint64 * (world * world))(*[matrix[r][j] for r in range(world) for j in range(world)])
    n = C.c_uint32(0)
    assert L.cmgpu_exchange_plan(rank, world, C.cast(flat, C.c_void_p), None, 0, C.byref(n)) == 0
    ops = (Op * max(1, n.value))()
    assert L.cmgpu_exchange_plan(rank, world, C.cast(flat, C.c_void_p), C.cast(ops, C.c_void_p), n.value, C.byref(n)) == 0
    return [(ops[i].kind, ops[i].peer, ops[i].records) for i in range(n.value)]


def matrices(world, rng):
    yield "all zero", [[0] * world for _ in range(world)]
    yield "full", [[rng.randrange(1, 10_000) for _ in range(world)] for _ in range(world)]
    for trial in range(40):
        m = [[rng.randrange(0, 5000) if rng.random() < 0.6 else 0 for _ in range(world)] for _ in range(world)]
        for r in rng.sample(range(world), rng.randrange(0, world)):  # ranks that ran out of input: they send nothing
            m[r] = [0] * world
        for j in rng.sample(range(world), rng.randrange(0, world)):  # ranks that own no chromosome with records
            for r in range(world):
                m[r][j] = 0
        yield "random %d" % trial, m
    m = [[0] * world for _ in range(world)]
    m[world - 1][0] = 7  # one record stream in the whole world
    yield "single edge", m


def simulate_group(plans, world):
    """RCCL group semantics: all sends / receives of a rank's group are posted together and complete as they are matched.
    Returns the posts left unmatched (a non-empty result is a hang)."""
    sends, recvs = {}, {}
    for me in range(world):
        for kind, peer, n in plans[me]:
            if kind == EX_SEND:
                assert (me, peer) not in sends, "two sends to one peer in one round"
                sends[(me, peer)] = n
            elif kind == EX_RECV:
                assert (peer, me) not in recvs, "two receives from one peer in one round"
                recvs[(peer, me)] = n
    left = []
    for k, n in sends.items():
        if recvs.pop(k, None) != n:
            left.append(("send", k, n))
    left += [("recv", k, n) for k, n in recvs.items()]
    return left


@pytest.mark.parametrize("world", [1, 2, 3, 4, 8])
def test_rounds_cannot_deadlock(world):
    L = _capi.lib()
    rng = random.Random(1234 + world)
    for name, m in matrices(world, rng):
        plans = [plan(L, r, world, m) for r in range(world)]
        # the same collectives first, on every rank, whatever the rank holds
        want = [EX_ALLGATHER_COUNTS] + ([EX_ALLGATHER_STATUS] if world > 1 else [])
        for r in range(world):
            kinds = [k for k, _, _ in plans[r]]
            assert kinds[:len(want)] == want, (name, r, kinds)
            assert all(k in (EX_COPY_SELF, EX_SEND, EX_RECV) for k in kinds[len(want):]), (name, r, kinds)
            assert plans[r][0][2] == world  # words contributed: world counts (+ the record kind, added by the step)
        assert simulate_group(plans, world) == [], (name, m)
        # every matrix entry is moved exactly once
        for r in range(world):
            for j in range(world):
                moved = [n for k, p, n in plans[r] if (k == EX_SEND and p == j) or (k == EX_COPY_SELF and r == j)]
                assert moved == ([m[r][j]] if m[r][j] else []), (name, r, j)
            for kind, peer, n in plans[r]:
                if kind == EX_RECV:
                    assert m[peer][r] == n and peer != r
        # the stagger: at distance d a rank sends to rank + d and receives from rank - d -- no rank is everybody's first target
        if world > 2 and name == "full":
            first_targets = [next(p for k, p, _ in plans[r] if k == EX_SEND) for r in range(world)]
            assert sorted(first_targets) == list(range(world))


def test_plan_refuses_bad_arguments():
    L = _capi.lib()
    m = (C.c_uint64 * 4)(1, 2, 3, 4)
    n = C.c_uint32(0)
    mp = C.cast(m, C.c_void_p)
    assert L.cmgpu_exchange_plan(2, 2, mp, None, 0, C.byref(n)) != 0   # rank outside the world
    assert L.cmgpu_exchange_plan(0, 0, mp, None, 0, C.byref(n)) != 0
    ops = (Op * 1)()
    assert L.cmgpu_exchange_plan(0, 2, mp, C.cast(ops, C.c_void_p), 1, C.byref(n)) != 0 and n.value == 5  # too small: the count is still reported
