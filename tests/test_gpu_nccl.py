"""GPU test of the RCCL transport (torch.distributed backend "nccl") of mugiq_amd.comm.GridComm with ONE rank: the
one-GPU box cannot host two nccl ranks (one rank per device), but every nccl-only branch of GridComm -- `_wire`, the
device-side reduce / all_gather / broadcast, batch_isend_irecv on an ExternalStream, the transfer group -- runs on the
hardware here.  N > 1 over nccl is the round-end driver's 8-GPU run; N > 1 semantics are covered over gloo
(tests/test_multi_rank_cpu.py, tests/test_gpu_driver.py)."""
import pytest
import torch.multiprocessing as mp

import mp_workers
from test_multi_rank_cpu import free_port

pytestmark = pytest.mark.gpu


def test_gridcomm_on_nccl_world_size_one():
    mp.spawn(mp_workers.nccl_world1_worker, args=(1, free_port(), False), nprocs=1, join=True)


def test_gridcomm_on_nccl_isend_irecv_to_self():
    """The halo message goes through RCCL's grouped isend / irecv (a send to self inside one batch)."""
    mp.spawn(mp_workers.nccl_world1_worker, args=(1, free_port(), True), nprocs=1, join=True)


def test_native_rccl_transport_one_rank_forced_partition():
    """The library's own transport (csrc/comm_rccl.cpp, mugiq_hip_rccl_comm_create / _fill; Python face: hip.RcclComm) with ONE rank
    and z, t forced-partitioned: the gauge borders and the eigenvector halos of the OPT plan travel as ncclSend / ncclRecv to self
    inside ncclGroupStart / End on the driver's halo stream (no Python callback in the data path), and the loops equal the
    unpartitioned run.  More than one rank cannot be had on a one-GPU box (RCCL refuses two ranks on one device); the reduce /
    gather / broadcast members are therefore covered by construction only (same semantics as GridComm's, which the gloo tests check)."""
    mp.spawn(mp_workers.native_rccl_worker, args=(1, free_port()), nprocs=1, join=True)
