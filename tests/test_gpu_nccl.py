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
