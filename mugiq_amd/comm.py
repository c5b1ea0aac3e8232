"""Process-grid communication for the loop engine over torch.distributed (backend "nccl" = RCCL over xGMI on
MI355X; "gloo" on CPU, or with device buffers staged through the host).

This is the role QUDA's comm layer + MPI play in the reference:
  * the 4-d process grid, comm_dim / comm_coord (tests/loop.cpp:781; include/contract_util.cuh:64,89);
  * the nearest-neighbour face exchange of ColorSpinorField::exchangeGhost (lib/contract_wrappers.cu:166-169);
  * the COMM_SPACE / COMM_TIME sub-communicators and MPI_Reduce / MPI_Gather / MPI_Bcast of
    Loop_Mugiq::setupComms + performMomentumProjection (lib/loop_mugiq.cpp:61-88, 406-424).
One process per GPU.  The rank <-> coordinate map is QUDA's default (t runs fastest).
"""
import ctypes

import numpy as np
import torch
import torch.distributed as dist

from . import _lib


class _DevPtr:
    """Expose a raw device pointer to torch through __cuda_array_interface__ (no copy, no ownership)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def device_bytes(ptr, nbytes, device):
    return torch.as_tensor(_DevPtr(ptr, nbytes), device=device)


def host_array(ptr, n_real, precision):
    ct = ctypes.c_double if precision == 8 else ctypes.c_float
    return np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ct)), shape=(int(n_real),))


class GridComm:
    def __init__(self, grid, device=None, force_partitioned=(0, 0, 0, 0), emulate_link_GBps=0.0):
        """grid = ranks along (x, y, z, t).  torch.distributed must be initialised; every rank must construct this
        (it creates the sub-groups collectively).
        force_partitioned[d] != 0 on an axis of extent 1: comm_dim_partitioned(d) is 1 there all the same (QUDA's
        comm_dim_partitioned_set / `--partition`), i.e. the driver runs its partitioned code path along d -- ghost zones,
        packed face layers, halo messages, interior / boundary tiles, gauge borders through sendrecv -- with this rank as
        its own neighbour (the message is a device copy, or a send-to-self with loopback_through_transport).  It lets ONE
        GPU run the halo machinery at the full per-GPU size; the result must equal the unpartitioned run.
        emulate_link_GBps > 0 (measurement aid, self-neighbour messages only): a message to self is a device copy, i.e. it is over in
        a fraction of the time the same bytes take over one xGMI link, and the overlap schedule of the driver (halos posted ahead,
        interior tiles first, boundary tiles after the halo event) is never put to the test.  With this set, the stream that carries
        the message is additionally held busy (a spin kernel on one wave) until bytes / rate have passed -- messages of different
        (axis, direction) inside one transfer group count as different links and run side by side, like different peers over xGMI.
        The data are the same; only the time at which the halo event fires changes."""
        self.grid = tuple(int(g) for g in grid)
        self.force_partitioned = tuple(1 if f else 0 for f in force_partitioned)
        self.device = torch.device(device) if device is not None else torch.device("cpu")
        self._cb = None
        self._group = None            # deferred (send, recv, staged recv target, dst, src) while a group is open
        # A grid of extent 1 along `dim` makes a rank its own neighbour; the driver exchanges along such an axis only under
        # forced partitioning, and sendrecv serves it with a device copy.  loopback_through_transport = True sends that
        # message through the transport as well (a send to self inside one batch): ONE rank then exercises the real
        # isend / irecv path of the backend.
        self.loopback_through_transport = False
        self.emulate_link_GBps = float(emulate_link_GBps)
        self._emu_bytes = {}          # (dim, direction) -> bytes sent to self since the last delay
        self._emu_cycles_per_s = None
        if not dist.is_initialized():
            # one process without a process group: only the 1x1x1x1 grid, whose every message is a copy to self (forced
            # partitioning on one device); the reduce / gather / bcast callbacks are never reached with size == 1
            assert int(np.prod(self.grid)) == 1, "a process grid %s needs torch.distributed to be initialised" % (self.grid,)
            self.size, self.rank, self.coord, self.backend = 1, 0, (0, 0, 0, 0), "local"
            self.space_group = self.time_group = None
            self.space_root, self.time_ranks, self.is_time_process = 0, [0], True
            return
        self.size = dist.get_world_size()
        self.rank = dist.get_rank()
        assert int(np.prod(self.grid)) == self.size, "process grid %s does not match world size %d" % (self.grid, self.size)
        self.coord = self.coords_of(self.rank)
        self.backend = dist.get_backend()
        # COMM_SPACE: ranks with equal t-coordinate, root = the one with x=y=z=0 (lowest world rank of the group)
        # COMM_TIME : ranks with x=y=z=0, ordered by t, root t=0                      lib/loop_mugiq.cpp:61-88
        self.space_group, self.space_root = None, None
        for t in range(self.grid[3]):
            ranks = [r for r in range(self.size) if self.coords_of(r)[3] == t]
            g = dist.new_group(ranks)
            if self.coord[3] == t:
                self.space_group, self.space_root = g, min(ranks)
        self.time_ranks = [self.rank_of((0, 0, 0, t)) for t in range(self.grid[3])]
        self.time_group = dist.new_group(self.time_ranks)
        self.is_time_process = self.coord[:3] == (0, 0, 0)

    # ---- topology (QUDA's comm_rank_from_coords: x slowest, t fastest) --------------------------------------
    def coords_of(self, rank):
        gx, gy, gz, gt = self.grid
        t = rank % gt
        z = (rank // gt) % gz
        y = (rank // (gt * gz)) % gy
        x = rank // (gt * gz * gy)
        return (x, y, z, t)

    def rank_of(self, c):
        gx, gy, gz, gt = self.grid
        return ((c[0] * gy + c[1]) * gz + c[2]) * gt + c[3]

    def comm_dim_partitioned(self, d):
        return 1 if (self.grid[d] > 1 or self.force_partitioned[d]) else 0

    def neighbour(self, dim, direction):
        c = list(self.coord)
        c[dim] = (c[dim] + direction) % self.grid[dim]
        return self.rank_of(c)

    # ---- halo -------------------------------------------------------------------------------------------------
    def sendrecv(self, send, recv, dim, direction):
        """Send `send` to the neighbour at coord[dim]+direction, receive `recv` from coord[dim]-direction.
        Tensors may live on the GPU; with a CPU-only backend (gloo) they are staged through host memory."""
        dst, src = self.neighbour(dim, direction), self.neighbour(dim, -direction)
        if dst == self.rank and not self.loopback_through_transport:
            # extent 1 along `dim` (forced partitioning): my own face is my ghost zone -- a copy on the current stream
            recv.copy_(send, non_blocking=True)
            if self.emulate_link_GBps > 0 and send.is_cuda:
                key = (dim, direction)
                self._emu_bytes[key] = self._emu_bytes.get(key, 0) + send.numel() * send.element_size()
                if self._group is None:
                    self._emu_delay()
            return
        stage = self.backend == "gloo" and send.is_cuda
        s = send.cpu() if stage else send
        r = torch.empty_like(recv, device="cpu") if stage else recv
        if self._group is not None:
            self._group.append((s, r, recv if stage else None, dst, src))     # issued together at group_end
            return
        ops = [dist.P2POp(dist.isend, s, dst), dist.P2POp(dist.irecv, r, src)]
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        if stage:
            recv.copy_(r)

    def group_begin(self):
        """Defer the sendrecv calls that follow: the halos of different axes go to different neighbours and can share one
        batch (ncclGroupStart/End under torch's batch_isend_irecv), i.e. use their xGMI links at the same time."""
        self._group = []

    def _emu_delay(self):
        """hold the current stream for what the slowest emulated link needs for the bytes handed to it (see __init__)"""
        if not self._emu_bytes:
            return
        seconds = max(self._emu_bytes.values()) / (self.emulate_link_GBps * 1e9)
        self._emu_bytes = {}
        if self._emu_cycles_per_s is None:           # calibrate the spin kernel's clock once
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            cal = torch.cuda.Stream(device=self.device)
            with torch.cuda.stream(cal):
                torch.cuda._sleep(1000)
                e0.record()
                torch.cuda._sleep(20_000_000)
                e1.record()
            cal.synchronize()
            self._emu_cycles_per_s = 20_000_000 / (e0.elapsed_time(e1) * 1e-3)
        torch.cuda._sleep(int(seconds * self._emu_cycles_per_s))

    def group_end(self):
        pending, self._group = self._group, None
        if self.emulate_link_GBps > 0:
            self._emu_delay()
        if not pending:
            return
        ops = []
        for s, r, _, dst, src in pending:
            ops += [dist.P2POp(dist.isend, s, dst), dist.P2POp(dist.irecv, r, src)]
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        for s, r, target, _, _ in pending:
            if target is not None:
                target.copy_(r)

    # ---- FT reduction (host-side payloads, tiny) -----------------------------------------------------------------
    def _wire(self, t):
        return t.to(self.device) if self.backend == "nccl" else t

    def reduce_space(self, send, recv):
        """recv = sum over the ranks sharing my t-coordinate of send, on the group's time process only."""
        t = self._wire(send.clone())
        dist.reduce(t, dst=self.space_root, op=dist.ReduceOp.SUM, group=self.space_group)
        if self.rank == self.space_root:
            recv.copy_(t.cpu())

    def gather_time(self, send, recv):
        """Concatenate the time processes' buffers in t order into recv (significant on the t=0 time process)."""
        if not self.is_time_process:
            return
        parts = [torch.empty_like(self._wire(send)) for _ in self.time_ranks]
        dist.all_gather(parts, self._wire(send.clone()), group=self.time_group)
        if self.coord[3] == 0:
            recv.copy_(torch.cat([p.cpu() for p in parts]))

    def bcast(self, buf):
        t = self._wire(buf.clone() if self.backend == "nccl" else buf)
        dist.broadcast(t, src=0)
        if self.backend == "nccl":
            buf.copy_(t.cpu())

    # ---- C callbacks for the driver (MugiqHipComm) ----------------------------------------------------------------
    def c_struct(self):
        if self._cb is not None:
            return self._cb[0]
        SENDRECV = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                    ctypes.c_int, ctypes.c_int, ctypes.c_void_p)
        REDUCE = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int)
        BCAST = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int)

        def guard(fn):
            def wrapped(*a):
                try:
                    fn(*a)
                    return 0
                except Exception as e:  # never let an exception cross the C boundary
                    import traceback
                    traceback.print_exc()
                    return 1
            return wrapped

        import contextlib

        @contextlib.contextmanager
        def _stream_ctx(stream):
            # `stream` is the stream the driver ordered the pack kernel on (its compute stream, or its halo stream
            # when the transfer overlaps the interior sites).  torch's collectives order against torch's CURRENT
            # stream, so make `stream` current for the duration of the exchange.
            cur = torch.cuda.current_stream(self.device)
            ptr = int(stream) if stream else 0
            if ptr != cur.cuda_stream:
                with torch.cuda.stream(torch.cuda.ExternalStream(ptr, device=self.device)):
                    yield
            else:
                yield

        def c_sendrecv(ctx, send_d, recv_d, nbytes, dim, direction, stream):
            s = device_bytes(send_d, nbytes, self.device)
            r = device_bytes(recv_d, nbytes, self.device)
            with _stream_ctx(stream):
                if self.backend == "gloo":
                    torch.cuda.current_stream(self.device).synchronize()      # staging through the host
                self.sendrecv(s, r, dim, direction)
                if self.backend == "gloo" and self._group is None:
                    torch.cuda.current_stream(self.device).synchronize()

        def c_reduce(ctx, send_h, recv_h, n, prec):
            s = torch.from_numpy(host_array(send_h, n, prec))
            r = torch.from_numpy(host_array(recv_h, n, prec))
            self.reduce_space(s, r)

        def c_gather(ctx, send_h, recv_h, n, prec):
            s = torch.from_numpy(host_array(send_h, n, prec))
            r = torch.from_numpy(host_array(recv_h, n * self.grid[3], prec))
            self.gather_time(s, r)

        def c_bcast(ctx, buf_h, n, prec):
            self.bcast(torch.from_numpy(host_array(buf_h, n, prec)))

        GBEGIN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p)
        GEND = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p)

        def c_group_begin(ctx):
            self.group_begin()

        def c_group_end(ctx, stream):
            with _stream_ctx(stream):
                if self.backend == "gloo":
                    torch.cuda.current_stream(self.device).synchronize()      # the pack kernels feeding the staged copies
                self.group_end()
                if self.backend == "gloo":
                    torch.cuda.current_stream(self.device).synchronize()

        class CComm(ctypes.Structure):
            _fields_ = [("ctx", ctypes.c_void_p), ("rank", ctypes.c_int), ("size", ctypes.c_int),
                        ("grid", ctypes.c_int * 4), ("coord", ctypes.c_int * 4),
                        ("sendrecv", SENDRECV), ("reduce_space", REDUCE), ("gather_time", REDUCE), ("bcast", BCAST),
                        ("group_begin", GBEGIN), ("group_end", GEND), ("partitioned", ctypes.c_int * 4)]

        fns = (SENDRECV(guard(c_sendrecv)), REDUCE(guard(c_reduce)), REDUCE(guard(c_gather)), BCAST(guard(c_bcast)),
               GBEGIN(guard(c_group_begin)), GEND(guard(c_group_end)))
        c = CComm()
        c.ctx = None
        c.rank, c.size = self.rank, self.size
        for d in range(4):
            c.grid[d] = self.grid[d]
            c.coord[d] = self.coord[d]
            c.partitioned[d] = self.force_partitioned[d]
        c.sendrecv, c.reduce_space, c.gather_time, c.bcast, c.group_begin, c.group_end = fns
        self._cb = (c, fns)          # keep the callbacks alive
        return c


class _CCommRaw(ctypes.Structure):
    """MugiqHipComm with the callbacks as raw pointers (filled in by the library itself)"""
    _fields_ = [("ctx", ctypes.c_void_p), ("rank", ctypes.c_int), ("size", ctypes.c_int),
                ("grid", ctypes.c_int * 4), ("coord", ctypes.c_int * 4),
                ("sendrecv", ctypes.c_void_p), ("reduce_space", ctypes.c_void_p), ("gather_time", ctypes.c_void_p), ("bcast", ctypes.c_void_p),
                ("group_begin", ctypes.c_void_p), ("group_end", ctypes.c_void_p), ("partitioned", ctypes.c_int * 4)]


class RcclComm:
    """The library's own RCCL transport (csrc/comm_rccl.cpp: ncclSend / ncclRecv groups on the halo stream, ncclReduce / AllGather /
    Broadcast over ncclCommSplit sub-communicators) behind the face of GridComm -- no Python in the data path; what a C++ host gets
    from mugiq_hip_rccl_comm_create.  One process: nothing else is needed.  Several: torch.distributed (any backend) must be
    initialised -- it only carries the 128-byte ncclUniqueId from rank 0 to the others; every rank must construct this (collective)."""

    def __init__(self, grid, device=None, force_partitioned=(0, 0, 0, 0), multipath=False):
        """multipath: halo messages over several xGMI paths at once (mugiq_hip_rccl_comm_set_multipath; more than two ranks)"""
        lib = _lib.load()
        self.grid = tuple(int(g) for g in grid)
        self.force_partitioned = tuple(1 if f else 0 for f in force_partitioned)
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        if dist.is_initialized():
            self.size, self.rank = dist.get_world_size(), dist.get_rank()
        else:
            self.size, self.rank = 1, 0
        assert int(np.prod(self.grid)) == self.size, "process grid %s does not match world size %d" % (self.grid, self.size)
        uid = (ctypes.c_ubyte * 128)()
        if self.rank == 0:
            _lib.check(lib.mugiq_hip_rccl_get_unique_id(ctypes.cast(uid, ctypes.c_void_p)))
        if self.size > 1:
            wire = self.device if dist.get_backend() == "nccl" else torch.device("cpu")
            t = torch.tensor(list(uid), dtype=torch.uint8, device=wire)
            dist.broadcast(t, src=0)
            uid = (ctypes.c_ubyte * 128)(*t.cpu().tolist())
        self._h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(lib.mugiq_hip_rccl_comm_create(ctypes.byref(self._h), ctypes.cast(uid, ctypes.c_void_p), self.rank, self.size,
                                                      _lib.int4(self.grid), _lib.int4(self.force_partitioned)))
        self._c = _CCommRaw()
        _lib.check(lib.mugiq_hip_rccl_comm_fill(self._h, ctypes.byref(self._c)))
        self.coord = tuple(self._c.coord)
        self.backend = "rccl-native"
        self.multipath = bool(multipath) and self.size > 2
        if multipath:
            _lib.check(lib.mugiq_hip_rccl_comm_set_multipath(self._h, 1))

    def comm_dim_partitioned(self, d):
        return 1 if (self.grid[d] > 1 or self.force_partitioned[d]) else 0

    def c_struct(self):
        return self._c

    def close(self):
        if self._h:
            _lib.load().mugiq_hip_rccl_comm_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
