"""world_size-2 `gloo` tests of the N>1 path on CPU: the comm layer the driver's callbacks use (process grid,
nearest-neighbour face exchange, COMM_SPACE reduce, COMM_TIME gather, broadcast) with the oracle doing the
per-rank arithmetic, checked against the single-domain oracle."""
import socket

import pytest
import torch.multiprocessing as mp

import mp_workers


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("grid", [(1, 1, 1, 2), (1, 1, 2, 1), (2, 1, 1, 1)])
def test_two_rank_gloo_loop_equals_single_domain(grid):
    mp.spawn(mp_workers.cpu_worker, args=(2, free_port(), grid), nprocs=2, join=True)


def test_four_rank_gloo_z_and_t_partitioned():
    mp.spawn(mp_workers.cpu_worker, args=(4, free_port(), (1, 1, 2, 2)), nprocs=4, join=True)


@pytest.mark.parametrize("grid,G", [((1, 1, 1, 4), (4, 4, 4, 8)), ((1, 1, 4, 1), (4, 4, 8, 4))])
def test_four_rank_gloo_extent_four(grid, G):
    """Extent 4 along one axis: the +1 and -1 neighbours are different ranks (with extent 2 they coincide, and a swapped
    send direction could not fail)."""
    mp.spawn(mp_workers.cpu_worker, args=(4, free_port(), grid, G), nprocs=4, join=True)


def test_eight_rank_gloo_baseline_grid():
    """BASELINE.json configs[2]'s process grid, 1 x 1 x 2 x 4 (z and t partitioned, t extent 4), on 8 gloo ranks."""
    mp.spawn(mp_workers.cpu_worker, args=(8, free_port(), (1, 1, 2, 4)), nprocs=8, join=True)


@pytest.mark.parametrize("grid,world,force", [((1, 1, 1, 1), 1, (0, 0, 1, 1)), ((1, 1, 1, 1), 1, (1, 1, 0, 0)), ((1, 1, 1, 2), 2, (0, 0, 1, 0))])
def test_forced_partitioning_self_neighbour(grid, world, force):
    """comm_dim_partitioned forced on axes of extent 1 (QUDA's comm_dim_partitioned_set): ghost zones, face exchange and
    gauge borders run with the rank as its own neighbour and must reproduce the single-domain result."""
    mp.spawn(mp_workers.cpu_worker, args=(world, free_port(), grid, (4, 4, 4, 8), force), nprocs=world, join=True)
