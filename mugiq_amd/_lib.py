"""ctypes loader for libmugiq_hip.so (built in-tree by `make -C mugiq_amd/csrc` / __graft_entry__.build()).

There is NO fallback: if the HIP library is missing or a call fails, an exception is raised.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MUGIQ_HIP_LIB selects another build of the same library (e.g. a diagnostic build); never a different backend
LIB_PATH = os.environ.get("MUGIQ_HIP_LIB") or os.path.join(_HERE, "libmugiq_hip.so")


class MugiqHipError(RuntimeError):
    """Raised for every non-zero status of the C ABI (the reference aborts through errorQuda)."""


class SpinorDesc(ctypes.Structure):
    """MugiqHipSpinorField (include/mugiq_hip.h)."""
    _fields_ = [("data", ctypes.c_void_p),
                ("precision", ctypes.c_int),
                ("field_order", ctypes.c_int),
                ("nParity", ctypes.c_int),
                ("volumeCB", ctypes.c_int),
                ("stride", ctypes.c_int),
                ("X", ctypes.c_int * 4),
                ("parity_offset", ctypes.c_int64),
                ("ghost", (ctypes.c_void_p * 2) * 4)]


class GaugeDesc(ctypes.Structure):
    """MugiqHipGaugeField (include/mugiq_hip.h)."""
    _fields_ = [("data", ctypes.c_void_p),
                ("precision", ctypes.c_int),
                ("X", ctypes.c_int * 4),
                ("R", ctypes.c_int * 4),
                ("stride", ctypes.c_int),
                ("parity_offset", ctypes.c_int64)]


class CoarseDesc(ctypes.Structure):
    """MugiqHipCoarseField (include/mugiq_hip.h)."""
    _fields_ = [("data", ctypes.c_void_p), ("precision", ctypes.c_int), ("nSpin", ctypes.c_int), ("nColor", ctypes.c_int),
                ("volumeCB", ctypes.c_int), ("stride", ctypes.c_int), ("X", ctypes.c_int * 4), ("parity_offset", ctypes.c_int64)]


class TransferDesc(ctypes.Structure):
    """MugiqHipTransfer (include/mugiq_hip.h)."""
    _fields_ = [("V", ctypes.c_void_p), ("precision", ctypes.c_int), ("nVec", ctypes.c_int), ("geoBlockSize", ctypes.c_int * 4),
                ("spinBlockSize", ctypes.c_int), ("X", ctypes.c_int * 4), ("stride", ctypes.c_int), ("parity_offset", ctypes.c_int64)]


_I4 = ctypes.POINTER(ctypes.c_int)
_VP = ctypes.c_void_p
_SP = ctypes.POINTER(SpinorDesc)
_GP = ctypes.POINTER(GaugeDesc)

# name -> (restype, argtypes); every symbol include/mugiq_hip.h declares
SIGNATURES = {
    "mugiq_hip_version": (ctypes.c_int, []),
    "mugiq_hip_last_error": (ctypes.c_char_p, []),
    "mugiq_hip_device_count": (ctypes.c_int, []),
    "mugiq_hip_release_stream": (ctypes.c_int, [_VP]),
    "mugiq_hip_probe_read_bandwidth": (ctypes.c_int, [_VP, ctypes.c_size_t, ctypes.c_int, _VP]),
    "mugiq_hip_debug_poison_lds": (ctypes.c_int, [_VP]),
    "mugiq_hip_copy_gamma_coeff_to_symbol": (ctypes.c_int, [ctypes.c_int]),
    "mugiq_hip_copy_gamma_map_to_symbol": (ctypes.c_int, [ctypes.c_int]),
    "mugiq_hip_get_gamma_tables": (ctypes.c_int, [ctypes.POINTER(ctypes.c_double), _I4,
                                                  ctypes.POINTER(ctypes.c_double), _I4]),
    "mugiq_hip_gamma_name": (ctypes.c_char_p, [ctypes.c_int]),
    "mugiq_hip_perform_loop_contraction": (ctypes.c_int, [_VP, _SP, _SP, ctypes.c_double, _VP]),
    "mugiq_hip_perform_loop_contraction_batched": (ctypes.c_int, [_VP, _SP, _SP, ctypes.POINTER(ctypes.c_double),
                                                                  ctypes.c_int, _VP]),
    "mugiq_hip_perform_loop_contraction_batched_mixed": (ctypes.c_int, [_VP, ctypes.c_int, _SP, _SP, ctypes.POINTER(ctypes.c_double),
                                                                        ctypes.c_int, _VP]),
    "mugiq_hip_displaced_loop_contraction_fused_mixed": (ctypes.c_int, [_VP, ctypes.c_int, _SP, ctypes.POINTER(ctypes.c_double),
                                                                        ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), _I4,
                                                                        ctypes.c_int, ctypes.c_int, ctypes.c_int, _I4, _VP,
                                                                        ctypes.c_int, _VP]),
    "mugiq_hip_displaced_loop_contraction_fused_region": (ctypes.c_int, [_VP, ctypes.c_int, _SP, ctypes.POINTER(ctypes.c_double),
                                                                         ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), _I4,
                                                                         ctypes.c_int, ctypes.c_int, ctypes.c_int, _I4, _VP,
                                                                         ctypes.c_int, ctypes.c_int, _VP]),
    "mugiq_hip_displaced_loop_contraction_fused_carry": (ctypes.c_int, [_VP, ctypes.c_int, _SP, ctypes.POINTER(ctypes.c_double),
                                                                        ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), _I4,
                                                                        ctypes.c_int, ctypes.c_int, ctypes.c_int, _I4, _VP,
                                                                        ctypes.c_int, ctypes.c_int, _VP, _I4, _VP]),
    "mugiq_hip_perform_covariant_displacement_vector": (ctypes.c_int, [_SP, _SP, _GP, ctypes.c_int, ctypes.c_int,
                                                                       _I4, _VP]),
    "mugiq_hip_pack_face": (ctypes.c_int, [_VP, _SP, ctypes.c_int, ctypes.c_int, _VP]),
    "mugiq_hip_exchange_ghost_vec": (ctypes.c_int, [_SP, _VP, _VP]),
    "mugiq_hip_rccl_get_unique_id": (ctypes.c_int, [_VP]),
    "mugiq_hip_rccl_comm_create": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), _VP, ctypes.c_int, ctypes.c_int, _I4, _I4]),
    "mugiq_hip_rccl_comm_from_nccl": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), _VP, _I4, _I4]),
    "mugiq_hip_rccl_comm_fill": (ctypes.c_int, [_VP, _VP]),
    "mugiq_hip_rccl_comm_destroy": (ctypes.c_int, [_VP]),
    "mugiq_hip_rccl_comm_set_multipath": (ctypes.c_int, [_VP, ctypes.c_int]),
    "mugiq_hip_rccl_relay_plan": (ctypes.c_int, [ctypes.c_int, _I4, ctypes.c_int, ctypes.c_int, ctypes.c_size_t, ctypes.c_int, _I4, _I4, _I4,
                                                ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]),
    "mugiq_hip_alloc_spinor_like": (ctypes.c_int, [_SP, _SP, ctypes.c_int, _I4]),
    "mugiq_hip_free_spinor": (ctypes.c_int, [_SP]),
    "mugiq_hip_copy_spinor": (ctypes.c_int, [_SP, _SP, _VP]),
    "mugiq_hip_zero_spinor": (ctypes.c_int, [_SP, _VP]),
    "mugiq_hip_create_phase_matrix": (ctypes.c_int, [_VP, _I4, ctypes.c_longlong, ctypes.c_int, ctypes.c_int,
                                                     _I4, _I4, _I4, ctypes.c_int, _VP]),
    "mugiq_hip_convert_idx_order_map_gamma": (ctypes.c_int, [_VP, _VP, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                             ctypes.c_int, _I4, ctypes.c_int, _VP]),
    "mugiq_hip_momentum_projection_workspace": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int, ctypes.c_longlong,
                                                                  ctypes.c_int, ctypes.c_int]),
    "mugiq_hip_momentum_projection": (ctypes.c_int, [_VP, _VP, _VP, ctypes.c_int, ctypes.c_int, ctypes.c_longlong,
                                                     ctypes.c_int, ctypes.c_int, _VP, ctypes.c_size_t, _VP]),
    "mugiq_hip_momentum_projection_separable_workspace": (ctypes.c_size_t, [_I4, ctypes.c_int, _I4, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "mugiq_hip_momentum_projection_separable": (ctypes.c_int, [_VP, _VP, _I4, ctypes.c_int, ctypes.c_int, _I4, _I4, _I4, ctypes.c_int,
                                                               ctypes.c_int, ctypes.c_int, _VP, ctypes.c_size_t, _VP]),
    "mugiq_hip_convert_and_project": (ctypes.c_int, [_VP, _VP, ctypes.c_int, ctypes.c_int, _I4, ctypes.c_int, ctypes.c_int, _I4, _I4, _I4,
                                                     ctypes.c_int, _VP, ctypes.c_size_t, _VP]),
    "mugiq_hip_convert_and_project_slots": (ctypes.c_int, [_VP, _VP, ctypes.c_int, _I4, ctypes.c_int, _I4, ctypes.c_int, ctypes.c_int, _I4, _I4, _I4,
                                                           ctypes.c_int, _VP, ctypes.c_size_t, _VP]),
    "mugiq_hip_reflect_momentum_space": (ctypes.c_int, [_VP, ctypes.c_int, ctypes.c_int, _I4, ctypes.c_int, _I4, ctypes.c_int, ctypes.c_int,
                                                        ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "mugiq_hip_pack_face_layers": (ctypes.c_int, [_VP, _SP, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _VP]),
    "mugiq_hip_reflect_displaced_loop": (ctypes.c_int, [_VP, _VP, _VP, _I4, ctypes.c_int, ctypes.c_int, ctypes.c_int, _I4, ctypes.c_int, _VP]),
    "mugiq_hip_pack_loop_layers": (ctypes.c_int, [_VP, _VP, _I4, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _VP]),
    "mugiq_hip_displaced_loop_contraction_fused": (ctypes.c_int, [_VP, _SP, ctypes.POINTER(ctypes.c_double), ctypes.c_int,
                                                                  ctypes.POINTER(ctypes.c_void_p), _I4, ctypes.c_int,
                                                                  ctypes.c_int, ctypes.c_int, _I4, _VP, ctypes.c_int, _VP]),
    "mugiq_hip_prolongate_batched": (ctypes.c_int, [_SP, ctypes.POINTER(CoarseDesc), ctypes.c_int, ctypes.POINTER(TransferDesc), _VP]),
    "mugiq_hip_prolongate_contract_batched": (ctypes.c_int, [_VP, ctypes.c_int, ctypes.POINTER(CoarseDesc), ctypes.POINTER(ctypes.c_double),
                                                             ctypes.c_int, ctypes.POINTER(TransferDesc), _VP]),
    # driver (struct pointers are passed with ctypes.byref; see mugiq_amd/loop.py for the struct definitions)
    "mugiq_hip_loop_create": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), _VP, _SP, ctypes.POINTER(ctypes.c_double),
                                             ctypes.c_int, _VP, _VP]),
    "mugiq_hip_loop_create_coarse_levels": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), _VP, ctypes.POINTER(CoarseDesc),
                                                           ctypes.POINTER(ctypes.c_double), ctypes.c_int, ctypes.POINTER(TransferDesc),
                                                           ctypes.c_int, ctypes.c_int, _VP, _VP]),
    "mugiq_hip_prolongate_coarse_batched": (ctypes.c_int, [ctypes.POINTER(CoarseDesc), ctypes.POINTER(CoarseDesc), ctypes.c_int,
                                                           ctypes.POINTER(TransferDesc), _VP]),
    "mugiq_hip_loop_create_coarse": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), _VP, ctypes.POINTER(CoarseDesc),
                                                    ctypes.POINTER(ctypes.c_double), ctypes.c_int, ctypes.POINTER(TransferDesc),
                                                    ctypes.c_int, _VP, _VP]),
    "mugiq_hip_loop_compute": (ctypes.c_int, [_VP]),
    "mugiq_hip_loop_get_info": (ctypes.c_int, [_VP, _VP]),
    "mugiq_hip_loop_set_profiling": (ctypes.c_int, [_VP, ctypes.c_int]),
    "mugiq_hip_loop_get_phases": (ctypes.c_int, [_VP, _VP, ctypes.c_int]),
    "mugiq_hip_loop_get_entry": (ctypes.c_int, [_VP, ctypes.c_int, _I4]),
    "mugiq_hip_loop_entry_derived_from": (ctypes.c_int, [_VP, ctypes.c_int]),
    "mugiq_hip_loop_ultra_local_carrier": (ctypes.c_int, [_VP]),
    "mugiq_hip_loop_halos_packed_in_entry": (ctypes.c_int, [_VP]),
    "mugiq_hip_loop_data_pos_d": (_VP, [_VP]),
    "mugiq_hip_loop_data_pos_h": (_VP, [_VP]),
    "mugiq_hip_loop_data_mom_bcast_h": (_VP, [_VP]),
    "mugiq_hip_loop_write_hdf5": (ctypes.c_int, [_VP]),
    "mugiq_hip_write_loops_hdf5_mom": (ctypes.c_int, [ctypes.c_char_p, _VP, ctypes.c_int, ctypes.c_int, _I4, ctypes.c_int,
                                                      ctypes.POINTER(ctypes.c_char_p), _I4, _I4, ctypes.c_int, ctypes.c_int]),
    "mugiq_hip_loop_destroy": (ctypes.c_int, [_VP]),
    "mugiq_hip_extended_gauge_bytes": (ctypes.c_size_t, [_I4, _I4, ctypes.c_int]),
    "mugiq_hip_alloc_extended_gauge": (ctypes.c_int, [_GP, _I4, _I4, ctypes.c_int]),
    "mugiq_hip_free_extended_gauge": (ctypes.c_int, [_GP]),
    "mugiq_hip_create_extended_gauge": (ctypes.c_int, [_GP, ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, _VP, _VP]),
    "mugiq_hip_parse_displace_entry_string": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, _I4, _I4]),
    "mugiq_hip_parse_displacement": (ctypes.c_int, [ctypes.c_char_p, _I4, _I4]),
}

_lib = None


def load():
    """Load the library once; raise (never fall back) if it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s not found: build it with `make -C mugiq_amd/csrc` or "
                          "`python -c 'import __graft_entry__ as g; g.build()'` -- there is no CPU fallback" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status):
    if status != 0:
        msg = load().mugiq_hip_last_error()
        raise MugiqHipError("status %d: %s" % (status, msg.decode() if msg else "?"))


def int4(v):
    return (ctypes.c_int * 4)(*[int(x) for x in v])
