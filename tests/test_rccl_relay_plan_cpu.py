"""The schedule of the multi-path halos of the library's RCCL transport (csrc/comm_rccl.cpp, mugiq_hip_rccl_relay_plan) on a
SIMULATED network: every rank's operations of a phase are posted, sends and receives between a pair of ranks are paired in the order
they were posted (NCCL's rule inside a group), the bytes are moved, and after the two phases every rank must hold exactly its
source's message.  No GPU and no RCCL are involved: the plan is a pure function.  (More than one rank cannot be run on the one-GPU
boxes this was written on.)"""
import ctypes

import numpy as np
import pytest

from mugiq_amd import _lib

SEND_USER, RECV_USER, RECV_BOUNCE, SEND_BOUNCE = 0, 1, 2, 3


def plan(rank, grid, dim, direction, nbytes):
    lib = _lib.load()
    n = 64
    ph, kd, pr = (ctypes.c_int * n)(), (ctypes.c_int * n)(), (ctypes.c_int * n)()
    off, ln = (ctypes.c_size_t * n)(), (ctypes.c_size_t * n)()
    bb = ctypes.c_size_t()
    k = lib.mugiq_hip_rccl_relay_plan(rank, _lib.int4(grid), dim, direction, nbytes, n, ph, kd, pr, off, ln, ctypes.byref(bb))
    assert 0 < k <= n
    return [(ph[i], kd[i], pr[i], off[i], ln[i]) for i in range(k)], bb.value


def neighbour(grid, r, dim, direction):
    c = [0, 0, 0, 0]
    rr = r
    for d in (3, 2, 1):
        c[d] = rr % grid[d]
        rr //= grid[d]
    c[0] = rr
    c[dim] = (c[dim] + direction) % grid[dim]
    return ((c[0] * grid[1] + c[1]) * grid[2] + c[2]) * grid[3] + c[3]


@pytest.mark.parametrize("grid,messages,nbytes", [
    ((1, 1, 2, 4), [(2, -1), (3, -1)], 100003),        # configs[2]'s grid: the z and t halos of one transfer group
    ((1, 1, 2, 4), [(3, 1), (2, 1), (3, -1)], 4096),
    ((1, 1, 1, 4), [(3, -1)], 1000),
    ((1, 1, 1, 8), [(3, 1)], 777),
    ((2, 1, 2, 2), [(0, 1), (2, -1), (3, 1)], 5000),
    ((1, 1, 1, 3), [(3, -1)], 100),
    ((1, 1, 1, 2), [(3, -1)], 300),                    # two ranks: no relay exists, everything goes directly
    ((1, 1, 2, 4), [(2, -1)], 7)])                     # fewer bytes than paths: empty parts are not posted
def test_every_byte_arrives_over_the_relays(grid, messages, nbytes):
    size = int(np.prod(grid))
    rng = np.random.default_rng(5)
    send = [[rng.integers(0, 256, nbytes, dtype=np.uint8) for _ in messages] for _ in range(size)]
    recv = [[np.zeros(nbytes, dtype=np.uint8) for _ in messages] for _ in range(size)]
    plans = [[plan(r, grid, d, s, nbytes) for (d, s) in messages] for r in range(size)]
    bounce = [[np.zeros(max(plans[r][m][1], 1), dtype=np.uint8) for m in range(len(messages))] for r in range(size)]
    relayed = 0
    for phase in (1, 2):
        # what every rank posts in this phase, message by message (the order csrc/comm_rccl.cpp uses), per peer
        sends = {(a, b): [] for a in range(size) for b in range(size)}
        recvs = {(a, b): [] for a in range(size) for b in range(size)}
        for r in range(size):
            for m in range(len(messages)):
                for ph, kind, peer, off, ln in plans[r][m][0]:
                    if ph != phase:
                        continue
                    assert ln > 0 and 0 <= peer < size
                    if kind == SEND_USER:
                        sends[(r, peer)].append(send[r][m][off:off + ln])
                    elif kind == SEND_BOUNCE:
                        sends[(r, peer)].append(bounce[r][m][off:off + ln].copy())
                        relayed += ln
                    elif kind == RECV_USER:
                        recvs[(peer, r)].append(recv[r][m][off:off + ln])
                    else:
                        recvs[(peer, r)].append(bounce[r][m][off:off + ln])
        for key in sends:                               # NCCL pairs the k-th send to a peer with that peer's k-th receive from me
            assert len(sends[key]) == len(recvs[key]), (phase, key, len(sends[key]), len(recvs[key]))
            for s_, r_ in zip(sends[key], recvs[key]):
                assert len(s_) == len(r_), (phase, key)
                r_[:] = s_
    for r in range(size):
        for m, (d, s) in enumerate(messages):
            src = neighbour(grid, r, d, -s)
            assert np.array_equal(recv[r][m], send[src][m]), (r, m)
    if size > 2 and nbytes >= 4096:
        assert relayed > 0.5 * size * len(messages) * nbytes * (size - 2) / (size - 1) * 0.9      # most of every message took a relay


def test_relay_parts_use_distinct_links():
    """1 x 1 x 2 x 4: the z halo of a rank leaves over seven different links in phase 1 (one direct, six first hops)"""
    ops, bounce = plan(5, (1, 1, 2, 4), 2, -1, 7 * 256 * 1000)
    first = [(kind, peer, ln) for ph, kind, peer, off, ln in ops if ph == 1 and kind == SEND_USER]
    assert len(first) == 7 and len(set(p for _, p, _ in first)) == 7 and all(ln == 256 * 1000 for _, _, ln in first)
    assert bounce == 6 * 256 * 1000
