"""Python mirror of MuGiq's operator API (lib/contract_wrappers.cu; declared at include/loop_mugiq.h:280-311
and include/displace.h:109-111 of the reference) over the C ABI of libmugiq_hip.so.

Same names and argument meaning as the reference's templated wrappers; <Float, order> come from the field
descriptors instead of template parameters; errors raise MugiqHipError where the reference calls errorQuda.
All functions enqueue on torch's current stream and return without synchronising.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from .fields import SpinorField, GaugeField, desc_array, coarse_desc_array

N_GAMMA = 16
DispDir = {"x": 0, "y": 1, "z": 2, "t": 3}          # include/enum_mugiq.h:72-78
DispSignMinus, DispSignPlus = 0, 1                   # include/enum_mugiq.h:81-85
LOOP_FT_SIGN_MINUS, LOOP_FT_SIGN_PLUS = -1, 1        # include/enum_mugiq.h:28-33
DisplaceFlagArray = ["+x", "-x", "+y", "-y", "+z", "-z", "+t", "-t"]   # include/displace.h:21


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _prec_of(t):
    if t.dtype == torch.complex128:
        return 8
    if t.dtype == torch.complex64:
        return 4
    raise TypeError("loop buffers must be complex64 or complex128 tensors")


def copyGammaCoeffStructToSymbol(precision):
    _lib.check(_lib.load().mugiq_hip_copy_gamma_coeff_to_symbol(int(precision)))


def copyGammaMapStructToSymbol(precision):
    _lib.check(_lib.load().mugiq_hip_copy_gamma_map_to_symbol(int(precision)))


def gammaTables():
    """(row_value[16][4] complex, column_index[16][4], map_sign[16], map_index[16]) the kernels were built with."""
    rv = (ctypes.c_double * 128)()
    ci = (ctypes.c_int * 64)()
    ms = (ctypes.c_double * 16)()
    mi = (ctypes.c_int * 16)()
    _lib.check(_lib.load().mugiq_hip_get_gamma_tables(rv, ci, ms, mi))
    rv = np.array(rv).reshape(16, 4, 2)
    return rv[..., 0] + 1j * rv[..., 1], np.array(ci).reshape(16, 4), np.array(ms), np.array(mi)


def GammaName(m):
    s = _lib.load().mugiq_hip_gamma_name(int(m))
    if s is None:
        raise IndexError(m)
    return s.decode()


def performLoopContraction(loopData_d, eVecL, eVecR, sigma):
    """loopData[tid + V*iG] += (1/sigma) vL^dag(x) G(iG) vR(x)      lib/contract_wrappers.cu:88-115"""
    assert isinstance(eVecL, SpinorField) and isinstance(eVecR, SpinorField)
    assert _prec_of(loopData_d) == eVecL.precision and loopData_d.numel() >= N_GAMMA * 2 * eVecL.volumeCB
    dl, dr = eVecL.desc(), eVecR.desc()
    _lib.check(_lib.load().mugiq_hip_perform_loop_contraction(loopData_d.data_ptr(), ctypes.byref(dl), ctypes.byref(dr),
                                                               float(sigma), _stream()))


def performLoopContractionBatched(loopData_d, eVecsL, eVecsR, sigmas):
    """The eigenvector loop of Loop_Mugiq::computeCoarseLoop (lib/loop_mugiq.cpp:478-503) in one launch.
    A complex128 loop buffer over fp32 eigenvectors selects the mixed-precision mode (fp64 accumulation)."""
    n = len(eVecsL)
    assert len(eVecsR) == n and len(sigmas) == n and n >= 1
    assert loopData_d.numel() >= N_GAMMA * 2 * eVecsL[0].volumeCB
    L = desc_array(eVecsL)
    R = L if eVecsR is eVecsL else desc_array(eVecsR)
    sg = (ctypes.c_double * n)(*[float(s) for s in sigmas])
    _lib.check(_lib.load().mugiq_hip_perform_loop_contraction_batched_mixed(loopData_d.data_ptr(), _prec_of(loopData_d), L, R, sg, n,
                                                                             _stream()))


def performCovariantDisplacementVector(dst, src, gauge, dispDir, dispSign, commDim=(0, 0, 0, 0)):
    """dst(x) = U_d(x) src(x+d) | U_d^dag(x-d) src(x-d)            lib/contract_wrappers.cu:171-198.
    The halo exchange (exchangeGhostVec) must have filled src.ghost for partitioned dims."""
    assert isinstance(gauge, GaugeField)
    dd, ds, dg = dst.desc(), src.desc(), gauge.desc()
    _lib.check(_lib.load().mugiq_hip_perform_covariant_displacement_vector(
        ctypes.byref(dd), ctypes.byref(ds), ctypes.byref(dg), int(dispDir), int(dispSign), _lib.int4(commDim), _stream()))


def exchangeGhostVec(x, comm):
    """exchangeGhostVec(ColorSpinorField *x)                       lib/contract_wrappers.cu:166-169
    depth-1 ghost zones of every partitioned dimension, both directions, through `comm` (a GridComm; None = one process)."""
    if comm is None:
        return
    for d in range(4):
        if comm.comm_dim_partitioned(d):
            x.alloc_ghost(d, 0), x.alloc_ghost(d, 1)
    dx, c = x.desc(), comm.c_struct()
    _lib.check(_lib.load().mugiq_hip_exchange_ghost_vec(ctypes.byref(dx), ctypes.cast(ctypes.byref(c), ctypes.c_void_p), _stream()))
    torch.cuda.current_stream().synchronize()


def packFace(face_d, src, dim, high):
    """Send half of exchangeGhostVec: face of `src` in ghost-zone layout (see mugiq_hip_pack_face)."""
    ds = src.desc()
    assert face_d.numel() >= 2 * 12 * src.face_cb(dim) and face_d.dtype == src.data.dtype
    _lib.check(_lib.load().mugiq_hip_pack_face(face_d.data_ptr(), ctypes.byref(ds), int(dim), int(high), _stream()))


def createPhaseMatrixGPU(phaseMatrix_d, momMatrix_h, locV3, Nmom, FTSign, localL, totalL, commCoord=(0, 0, 0, 0)):
    """ph[v3 + locV3*im] = exp(i FTSign 2pi p.x/L)                 lib/contract_wrappers.cu:50-77"""
    mom = np.ascontiguousarray(np.asarray(momMatrix_h, dtype=np.int32).reshape(-1))
    assert mom.size == 3 * Nmom and phaseMatrix_d.numel() >= locV3 * Nmom
    _lib.check(_lib.load().mugiq_hip_create_phase_matrix(
        phaseMatrix_d.data_ptr(), mom.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), int(locV3), int(Nmom), int(FTSign),
        _lib.int4(localL), _lib.int4(totalL), _lib.int4(commCoord), _prec_of(phaseMatrix_d), _stream()))


def convertIdxOrder_mapGamma(dataPosMP_d, dataPos_d, nData, nLoop, nParity, volumeCB, localL):
    """even-odd [nData][V] -> [v3][nData][t] with the G -> g5 G map  lib/contract_wrappers.cu:133-156"""
    assert dataPosMP_d.dtype == dataPos_d.dtype and dataPosMP_d.numel() >= nData * nParity * volumeCB
    _lib.check(_lib.load().mugiq_hip_convert_idx_order_map_gamma(
        dataPosMP_d.data_ptr(), dataPos_d.data_ptr(), int(nData), int(nLoop), int(nParity), int(volumeCB),
        _lib.int4(localL), _prec_of(dataPos_d), _stream()))


def momentumProjection(dataMom_d, dataPosMP_d, phaseMatrix_d, locT, nData, locV3, Nmom):
    """dataMom[M x N] = dataPosMP[M x K] phase[K x N] (the Zgemm/Cgemm of lib/loop_mugiq.cpp:363-378)"""
    assert dataMom_d.numel() >= locT * nData * Nmom
    _lib.check(_lib.load().mugiq_hip_momentum_projection(
        dataMom_d.data_ptr(), dataPosMP_d.data_ptr(), phaseMatrix_d.data_ptr(), int(locT), int(nData), int(locV3),
        int(Nmom), _prec_of(dataPosMP_d), None, 0, _stream()))


def momentumProjectionSeparable(dataMom_d, dataPosMP_d, momMatrix, FTSign, localL, totalL, locT, nData, commCoord=(0, 0, 0, 0)):
    """dataMom[M x Nmom] = sum over the local spatial volume of dataPosMP[M x K] * phase, one direction at a time."""
    mom = np.ascontiguousarray(np.asarray(momMatrix, dtype=np.int32).reshape(-1))
    n_mom = mom.size // 3
    assert dataMom_d.numel() >= locT * nData * n_mom and dataMom_d.dtype == dataPosMP_d.dtype
    L = (ctypes.c_int * 4)(*[int(x) for x in localL])
    T = (ctypes.c_int * 4)(*[int(x) for x in totalL])
    cc = (ctypes.c_int * 4)(*[int(x) for x in commCoord])
    _lib.check(_lib.load().mugiq_hip_momentum_projection_separable(
        dataMom_d.data_ptr(), dataPosMP_d.data_ptr(), mom.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), n_mom, int(FTSign), L, T, cc,
        int(locT), int(nData), _prec_of(dataPosMP_d), None, 0, _stream()))


def convertAndProject(dataMom_d, dataPos_d, nData, nLoop, momMatrix, FTSign, localL, totalL, commCoord=(0, 0, 0, 0)):
    """reorder + gamma5 map + momentum projection from the even-odd buffer in one go (see mugiq_hip_convert_and_project)."""
    mom = np.ascontiguousarray(np.asarray(momMatrix, dtype=np.int32).reshape(-1))
    n_mom = mom.size // 3
    assert dataMom_d.numel() >= localL[3] * nData * n_mom and dataMom_d.dtype == dataPos_d.dtype
    _lib.check(_lib.load().mugiq_hip_convert_and_project(
        dataMom_d.data_ptr(), dataPos_d.data_ptr(), int(nData), int(nLoop), mom.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), n_mom,
        int(FTSign), _lib.int4(localL), _lib.int4(totalL), _lib.int4(commCoord), _prec_of(dataPos_d), None, 0, _stream()))


def packFaceLayers(faces_d, eVecs, dim, high, layers):
    """[nVec][layers] ghost zones in one buffer (see mugiq_hip_pack_face_layers)."""
    d = desc_array(eVecs)
    assert faces_d.numel() >= len(eVecs) * layers * 24 * eVecs[0].face_cb(dim)
    _lib.check(_lib.load().mugiq_hip_pack_face_layers(faces_d.data_ptr(), d, len(eVecs), int(dim), int(high), int(layers), _stream()))


def reflectDisplacedLoop(dstSlot_d, srcSlot_d, localL, dispDir, dstDispSign, length, commDim=(0, 0, 0, 0), ghostLayers_d=None):
    """dst slot = eta_G conj(src slot shifted by -+length along dispDir): the entry of the opposite sign (see mugiq_hip.h)."""
    V = int(np.prod(localL))
    assert dstSlot_d.numel() >= 16 * V and srcSlot_d.numel() >= 16 * V and dstSlot_d.dtype == srcSlot_d.dtype
    L = (ctypes.c_int * 4)(*[int(x) for x in localL])
    cd = (ctypes.c_int * 4)(*[int(x) for x in commDim])
    _lib.check(_lib.load().mugiq_hip_reflect_displaced_loop(
        dstSlot_d.data_ptr(), srcSlot_d.data_ptr(), ghostLayers_d.data_ptr() if ghostLayers_d is not None else None, L,
        int(dispDir), int(dstDispSign), int(length), cd, _prec_of(srcSlot_d), _stream()))


def packLoopLayers(layers_d, slot_d, localL, dim, high, layers):
    """The `layers` boundary layers of one loop slot, as reflectDisplacedLoop expects its ghost layers."""
    V = int(np.prod(localL))
    assert layers_d.numel() >= 32 * layers * (V // 2 // localL[dim]) and layers_d.dtype == slot_d.dtype
    L = (ctypes.c_int * 4)(*[int(x) for x in localL])
    _lib.check(_lib.load().mugiq_hip_pack_loop_layers(layers_d.data_ptr(), slot_d.data_ptr(), L, int(dim), int(high), int(layers),
                                                      _prec_of(slot_d), _stream()))


def displacedLoopContractionFused(loopData_d, eVecs, sigmas, pathLinkFields, kValues, dispDir, dispSign,
                                  commDim=(0, 0, 0, 0), ghostLayers_d=None, layers=0, ultraLocalSlot_d=None):
    """loop slot i += sum_n (1/sigma_n) v_n^dag G W_k v_n(x +- k mu), k = kValues[i] (see mugiq_hip.h).
    ultraLocalSlot_d: also ask for the ultra-local loop (displacement 0) in the same pass; returns whether the kernel carried it."""
    n, nk = len(eVecs), len(kValues)
    d = desc_array(eVecs)
    sg = (ctypes.c_double * n)(*[float(s) for s in sigmas])
    links = (ctypes.c_void_p * nk)(*[f.data.data_ptr() for f in pathLinkFields])
    kv = (ctypes.c_int * nk)(*[int(k) for k in kValues])
    if ultraLocalSlot_d is not None:
        carried = ctypes.c_int(0)
        _lib.check(_lib.load().mugiq_hip_displaced_loop_contraction_fused_carry(
            loopData_d.data_ptr(), _prec_of(loopData_d), d, sg, n, links, kv, nk, int(dispDir), int(dispSign), _lib.int4(commDim),
            ghostLayers_d.data_ptr() if ghostLayers_d is not None else None, int(layers), 0, ultraLocalSlot_d.data_ptr(),
            ctypes.byref(carried), _stream()))
        return bool(carried.value)
    _lib.check(_lib.load().mugiq_hip_displaced_loop_contraction_fused_mixed(
        loopData_d.data_ptr(), _prec_of(loopData_d), d, sg, n, links, kv, nk, int(dispDir), int(dispSign), _lib.int4(commDim),
        ghostLayers_d.data_ptr() if ghostLayers_d is not None else None, int(layers), _stream()))


def probeReadBandwidth(buf, nonTemporal=False):
    """Enqueue one streaming read of `buf` (a torch tensor); time it with events for the device's achievable GB/s."""
    _lib.check(_lib.load().mugiq_hip_probe_read_bandwidth(buf.data_ptr(), buf.numel() * buf.element_size(), int(nonTemporal), _stream()))


def prolongateEvecs(fineEvecs, coarseEvecs, transfer):
    """Loop_Mugiq::prolongateEvec (lib/loop_mugiq.cpp:277-319 = QUDA Transfer::P) for all eigenvectors in one launch."""
    assert len(fineEvecs) == len(coarseEvecs) >= 1
    t = transfer.desc()
    _lib.check(_lib.load().mugiq_hip_prolongate_batched(desc_array(fineEvecs), coarse_desc_array(coarseEvecs), len(fineEvecs),
                                                        ctypes.byref(t), _stream()))


def prolongateCoarseEvecs(finerEvecs, coarserEvecs, transfer):
    """One coarse -> coarse level, mg_env.transfer[lev-1]->P(tmpCSF[lev-1], tmpCSF[lev]) for all eigenvectors in one launch
    (lib/loop_mugiq.cpp:306-311)."""
    n = len(finerEvecs)
    assert len(coarserEvecs) == n and n >= 1
    t = transfer.desc()
    _lib.check(_lib.load().mugiq_hip_prolongate_coarse_batched(coarse_desc_array(finerEvecs), coarse_desc_array(coarserEvecs), n,
                                                               ctypes.byref(t), _stream()))


def prolongateContractBatched(loopData_d, coarseEvecs, sigmas, transfer):
    """loopData += sum_n (1/sigma_n) (P c_n)^dag G (P c_n) without writing the fine vectors (MG ultra-local loop)."""
    n = len(coarseEvecs)
    t = transfer.desc()
    sg = (ctypes.c_double * n)(*[float(s) for s in sigmas])
    _lib.check(_lib.load().mugiq_hip_prolongate_contract_batched(loopData_d.data_ptr(), _prec_of(loopData_d), coarse_desc_array(coarseEvecs),
                                                                 sg, n, ctypes.byref(t), _stream()))
