"""ctypes wrapper of oracle/libmugiq_oracle.so (the plain-C restatement of the contraction).
TEST INFRASTRUCTURE ONLY -- see the header of oracle/mugiq_oracle.c."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "libmugiq_oracle.so")
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            raise ImportError("%s missing: run `make -C oracle`" % _PATH)
        _lib = ctypes.CDLL(_PATH)
        _lib.oracle_num_threads.restype = ctypes.c_int
        _lib.oracle_set_num_threads.restype = None
        _lib.oracle_set_num_threads.argtypes = [ctypes.c_int]
        for name in ("oracle_loop_contract_f64", "oracle_loop_contract_f32"):
            f = getattr(_lib, name)
            f.restype = None
            f.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p),
                          ctypes.POINTER(ctypes.c_double), ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int,
                          ctypes.c_int64, ctypes.c_int64, ctypes.c_int]
        _lib.oracle_first_touch_copy.restype = None
        _lib.oracle_first_touch_copy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int]
    return _lib


def first_touch_copy(src, n_planes, S):
    """A copy of the flat numpy array `src` ([2 parities][n_planes][S] items) whose pages are first written by the OpenMP
    threads that will read them in loop_contract_native (same static partition of the site index)."""
    lib = load()
    assert src.flags["C_CONTIGUOUS"] and src.nbytes % (2 * n_planes * S) == 0
    dst = np.empty_like(src)                     # freshly mapped, untouched
    lib.oracle_first_touch_copy(dst.ctypes.data, src.ctypes.data, n_planes, S, src.nbytes // (2 * n_planes * S))
    return dst


def usable_cpus():
    """CPUs this process may actually use: the affinity mask, capped by the cgroup CPU quota (a GPU box hands a job 16 of its 256
    hardware threads that way; OpenMP's default would start one thread per hardware thread and have them throttled)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p_ = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p_))
        except (OSError, ValueError):
            pass
    return n


def set_num_threads(n):
    load().oracle_set_num_threads(int(n))


def num_threads():
    return load().oracle_num_threads()


def loop_contract_native(loop, vL_bufs, vR_bufs, sigmas, volumeCB, stride, parity_offset, order,
                         site_begin=0, site_end=None):
    """loop (flat complex numpy, 16*V) += sum_n (1/sigma_n) vL_n^dag G vR_n over sites [site_begin, site_end);
    vL_bufs / vR_bufs: lists of flat native-layout complex numpy buffers (FLOAT2 / FLOAT4 order)."""
    lib = load()
    n = len(vL_bufs)
    f64 = loop.dtype == np.complex128
    for b in list(vL_bufs) + list(vR_bufs):
        assert b.dtype == loop.dtype and b.flags["C_CONTIGUOUS"]
    pl = (ctypes.c_void_p * n)(*[b.ctypes.data for b in vL_bufs])
    pr = (ctypes.c_void_p * n)(*[b.ctypes.data for b in vR_bufs])
    sg = (ctypes.c_double * n)(*[float(s) for s in sigmas])
    site_end = 2 * volumeCB if site_end is None else site_end
    fn = lib.oracle_loop_contract_f64 if f64 else lib.oracle_loop_contract_f32
    fn(loop.ctypes.data, pl, pr, sg, n, site_begin, site_end, volumeCB, stride, parity_offset, order)
    return loop
