"""Python face of the C++ driver (csrc/loop_driver.cpp): MugiqLoopParam + Loop_Mugiq of the reference
(include/mugiq.h:28-47, include/loop_mugiq.h:123-134), same member names."""
import ctypes
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np
import torch

from . import _lib
from .comm import device_bytes
from .fields import GaugeField, desc_array, coarse_desc_array, transfer_desc_array

LOOP_CALC_TYPE_BLAS, LOOP_CALC_TYPE_OPT_KERNEL, LOOP_CALC_TYPE_BASIC_KERNEL = 0, 1, 2   # include/enum_mugiq.h:35-41


class _CLoopParam(ctypes.Structure):
    _fields_ = [("Nmom", ctypes.c_int), ("momMatrix", ctypes.POINTER(ctypes.c_int)), ("FTSign", ctypes.c_int),
                ("calcType", ctypes.c_int), ("writeMomSpaceHDF5", ctypes.c_int), ("writePosSpaceHDF5", ctypes.c_int),
                ("doMomProj", ctypes.c_int), ("doNonLocal", ctypes.c_int), ("nDispEntries", ctypes.c_int),
                ("disp_entry", ctypes.POINTER(ctypes.c_char_p)), ("disp_str", ctypes.POINTER(ctypes.c_char_p)),
                ("disp_start", ctypes.POINTER(ctypes.c_int)), ("disp_stop", ctypes.POINTER(ctypes.c_int)),
                ("fname_mom_h5", ctypes.c_char_p), ("fname_pos_h5", ctypes.c_char_p),
                ("gauge", ctypes.POINTER(_lib.GaugeDesc)), ("loopPrecision", ctypes.c_int)]


class _CLoopInfo(ctypes.Structure):
    _fields_ = [("nDispEntries", ctypes.c_int), ("nLoop", ctypes.c_int), ("nData", ctypes.c_int), ("Nmom", ctypes.c_int),
                ("precision", ctypes.c_int), ("field_order", ctypes.c_int), ("loopPrecision", ctypes.c_int),
                ("localL", ctypes.c_int * 4), ("totalL", ctypes.c_int * 4), ("locT", ctypes.c_int), ("totT", ctypes.c_int),
                ("locV4", ctypes.c_longlong), ("locV3", ctypes.c_longlong), ("totV3", ctypes.c_longlong),
                ("nElemPosLocPerLoop", ctypes.c_longlong), ("nElemMomLocPerLoop", ctypes.c_longlong),
                ("nElemMomTotPerLoop", ctypes.c_longlong), ("nElemPosLoc", ctypes.c_longlong),
                ("nElemMomLoc", ctypes.c_longlong), ("nElemMomTot", ctypes.c_longlong), ("nElemPhMat", ctypes.c_longlong)]


class _CLoopPhase(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int), ("entry", ctypes.c_int), ("ms", ctypes.c_double), ("bytes", ctypes.c_double)]


PHASE_NAMES = ["ultra_local", "entry_fused", "entry_reflected", "entry_stepwise", "momentum_projection", "halo_transfer",
               "entry_interior", "entry_boundary", "prolongation", "halo_prepare", "halo_wait", "momentum_copy",
               "momentum_reduce", "total_wall", "scratch_alloc", "momentum_reflect"]                           # MUGIQ_HIP_PHASE_* (include/mugiq_hip.h)


@dataclass
class MugiqLoopParam:
    """include/mugiq.h:28-47.  `gauge` is the border-extended device GaugeField (the reference passes host QDP
    links + QudaGaugeParam and lets Displace build it, lib/displace.cpp:104-134)."""
    Nmom: int = 0
    momMatrix: List[List[int]] = field(default_factory=list)
    FTSign: int = 1
    calcType: int = LOOP_CALC_TYPE_OPT_KERNEL
    writeMomSpaceHDF5: bool = False
    writePosSpaceHDF5: bool = False
    doMomProj: bool = False
    doNonLocal: bool = False
    disp_entry: List[str] = field(default_factory=list)
    disp_str: List[str] = field(default_factory=list)
    fname_mom_h5: str = ""
    fname_pos_h5: str = ""
    disp_start: List[int] = field(default_factory=list)
    disp_stop: List[int] = field(default_factory=list)
    gauge: Optional[GaugeField] = None
    loopPrecision: int = 0          # not in the reference: 8 over fp32 eigenvectors = mixed precision (configs[3])

    def set_displace_entry_string(self, s):
        """--displace-entry-string "+z:1,8;-x:3" (tests/loop.cpp:656-705)"""
        self.disp_entry, self.disp_str, self.disp_start, self.disp_stop = parseDisplaceEntryString(s)
        self.doNonLocal = True
        return self


def parseDisplaceEntryString(s, max_entries=64):
    lib = _lib.load()
    out = ctypes.create_string_buffer(4 * max_entries)
    a = (ctypes.c_int * max_entries)()
    b = (ctypes.c_int * max_entries)()
    n = lib.mugiq_hip_parse_displace_entry_string(s.encode(), max_entries, out, a, b)
    if n < 0:
        _lib.check(-n)
    strs = [out.raw[4 * i:4 * i + 4].split(b"\0")[0].decode() for i in range(n)]
    return s.split(";"), strs, list(a[:n]), list(b[:n])


def parseDisplacement(dstr):
    """Displace::setupDisplacement: "+x".."-t" -> (dir, sign)   lib/displace.cpp:206-223"""
    d, s = ctypes.c_int(), ctypes.c_int()
    _lib.check(_lib.load().mugiq_hip_parse_displacement(dstr.encode(), ctypes.byref(d), ctypes.byref(s)))
    return d.value, s.value


def read_momenta_file(path):
    """--momenta-filename: one "px py pz" integer triple per line (tests/loop.cpp:724-740)."""
    out = []
    for i, line in enumerate(open(path)):
        tok = line.split()
        if len(tok) < 3:
            raise ValueError("Incorrect file format in Line %d" % i)
        out.append([int(tok[0]), int(tok[1]), int(tok[2])])
    return out


def writeLoopsHDF5_Mom(filename, dataMom_bcast, momMatrix, disp_str, disp_start, disp_stop, locT, totT):
    """Stand-alone momentum-space writer (host only): the file tree of lib/loop_mugiq.cpp:529-656."""
    a = np.ascontiguousarray(dataMom_bcast)
    prec = 8 if a.dtype == np.complex128 else 4
    mom = np.ascontiguousarray(np.asarray(momMatrix, dtype=np.int32).reshape(-1))
    ne = len(disp_str)
    ds = (ctypes.c_char_p * max(ne, 1))(*[s.encode() for s in disp_str])
    st = (ctypes.c_int * max(ne, 1))(*[int(x) for x in disp_start])
    sp = (ctypes.c_int * max(ne, 1))(*[int(x) for x in disp_stop])
    _lib.check(_lib.load().mugiq_hip_write_loops_hdf5_mom(
        filename.encode(), a.ctypes.data_as(ctypes.c_void_p), prec, mom.size // 3, mom.ctypes.data_as(ctypes.POINTER(ctypes.c_int)),
        ne, ds, st, sp, int(locT), int(totT)))


def reflectMomentumSpace(dataMom_bcast, momMatrix, FTSign, totalL, nLoop, locT, totT, dstSlot, srcSlot, dispDir, dstDispSign, length):
    """mugiq_hip_reflect_momentum_space (host only): fill loop slot `dstSlot` of the gathered momentum-space array (flat complex
    numpy, layout of lib/loop_mugiq.cpp:415-424) from its opposite-sign source slot, in place."""
    a = dataMom_bcast
    assert a.flags["C_CONTIGUOUS"] and a.dtype in (np.complex128, np.complex64)
    mom = np.ascontiguousarray(np.asarray(momMatrix, dtype=np.int32).reshape(-1))
    _lib.check(_lib.load().mugiq_hip_reflect_momentum_space(
        a.ctypes.data_as(ctypes.c_void_p), 8 if a.dtype == np.complex128 else 4, mom.size // 3,
        mom.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), int(FTSign), _lib.int4(totalL), int(nLoop), int(locT), int(totT),
        int(dstSlot), int(srcSlot), int(dispDir), int(dstDispSign), int(length)))
    return a


class Loop_Mugiq:
    """Loop_Mugiq<Float, order>(loopParams, eigsolve): Float/order come from the eigenvector fields;
    `eVecs` / `eVals_sigma` are what the reference reads out of Eigsolve_Mugiq (lib/loop_mugiq.cpp:442,479)."""

    def __init__(self, loopParams, eVecs, eVals_sigma, comm=None, transfer=None):
        """`transfer` given: eVecs are CoarseField eigenvectors (eigsolve->computeCoarse) prolonged with it; a list
        [finest, level 1 -> 2, ...] for an MG hierarchy with several coarse levels (eVecs on the coarsest one)."""
        lib = _lib.load()
        self._keep = []
        self._params, self._transfer = loopParams, transfer
        p = _CLoopParam()
        n_mom = int(loopParams.Nmom) if loopParams.Nmom else len(loopParams.momMatrix)
        mom = np.ascontiguousarray(np.asarray(loopParams.momMatrix, dtype=np.int32).reshape(-1)) if n_mom else np.zeros(0, np.int32)
        self._keep.append(mom)
        p.Nmom = n_mom
        p.momMatrix = mom.ctypes.data_as(ctypes.POINTER(ctypes.c_int)) if n_mom else None
        p.FTSign = int(loopParams.FTSign)
        p.calcType = int(loopParams.calcType)
        p.writeMomSpaceHDF5 = int(bool(loopParams.writeMomSpaceHDF5))
        p.writePosSpaceHDF5 = int(bool(loopParams.writePosSpaceHDF5))
        p.doMomProj = int(bool(loopParams.doMomProj))
        p.doNonLocal = int(bool(loopParams.doNonLocal))
        ne = len(loopParams.disp_str)
        if not (ne == len(loopParams.disp_start) == len(loopParams.disp_stop)):
            raise _lib.MugiqHipError("Displacement string length not compatible with displacement limits length")
        p.nDispEntries = ne
        ent = (ctypes.c_char_p * max(ne, 1))(*[(loopParams.disp_entry[i] if i < len(loopParams.disp_entry) else "").encode() for i in range(ne)])
        dst = (ctypes.c_char_p * max(ne, 1))(*[s.encode() for s in loopParams.disp_str])
        a = (ctypes.c_int * max(ne, 1))(*[int(x) for x in loopParams.disp_start])
        b = (ctypes.c_int * max(ne, 1))(*[int(x) for x in loopParams.disp_stop])
        self._keep += [ent, dst, a, b]
        p.disp_entry, p.disp_str, p.disp_start, p.disp_stop = ent, dst, a, b
        p.fname_mom_h5 = loopParams.fname_mom_h5.encode()
        p.fname_pos_h5 = loopParams.fname_pos_h5.encode()
        if loopParams.gauge is not None:
            g = loopParams.gauge.desc()
            self._keep += [g, loopParams.gauge]
            p.gauge = ctypes.pointer(g)
        p.loopPrecision = int(loopParams.loopPrecision)
        self.eVecs = list(eVecs)
        sg = (ctypes.c_double * len(self.eVecs))(*[float(s) for s in eVals_sigma])
        self.comm = comm
        c = comm.c_struct() if comm is not None else None
        self._handle = ctypes.c_void_p()
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        if transfer is None:
            _lib.check(lib.mugiq_hip_loop_create(ctypes.byref(self._handle), ctypes.byref(p), desc_array(self.eVecs), sg,
                                                 len(self.eVecs), ctypes.byref(c) if c is not None else None, stream))
        elif isinstance(transfer, (list, tuple)):
            ts = transfer_desc_array(transfer)
            self._keep += [ts, list(transfer)]
            _lib.check(lib.mugiq_hip_loop_create_coarse_levels(ctypes.byref(self._handle), ctypes.byref(p), coarse_desc_array(self.eVecs),
                                                               sg, len(self.eVecs), ts, len(transfer), 2,
                                                               ctypes.byref(c) if c is not None else None, stream))
        else:
            t = transfer.desc()
            self._keep += [t, transfer]
            _lib.check(lib.mugiq_hip_loop_create_coarse(ctypes.byref(self._handle), ctypes.byref(p), coarse_desc_array(self.eVecs),
                                                        sg, len(self.eVecs), ctypes.byref(t), 2,
                                                        ctypes.byref(c) if c is not None else None, stream))
        info = _CLoopInfo()
        _lib.check(lib.mugiq_hip_loop_get_info(self._handle, ctypes.byref(info)))
        self.info = info
        for k in ("nDispEntries", "nLoop", "nData", "Nmom", "precision", "loopPrecision", "locT", "totT", "locV4", "locV3", "totV3",
                  "nElemPosLocPerLoop", "nElemMomLocPerLoop", "nElemMomTotPerLoop", "nElemPosLoc", "nElemMomLoc", "nElemMomTot"):
            setattr(self, k, getattr(info, k))
        self.localL, self.totalL = tuple(info.localL), tuple(info.totalL)
        self.device = self.eVecs[0].device

    def entry(self, idx):
        """(dispDir, dispSign, dispStart, dispStop, nLoopPerEntry, nLoopOffset) of displacement entry idx."""
        out = (ctypes.c_int * 6)()
        _lib.check(_lib.load().mugiq_hip_loop_get_entry(self._handle, idx, out))
        return tuple(out)

    def printLoopComputeParams(self, out=print):
        """Loop_Mugiq::printLoopComputeParams (lib/loop_mugiq.cpp:233-273): the same report, line by line; the lines that
        named CUDA libraries name what runs here."""
        p, coarse = self._params, self._transfer is not None
        out("******************************************")
        out("    Parameters of the Loop Computation")
        out("Precision is %s" % ("single" if self.precision == 4 else "double"))
        if self.loopPrecision != self.precision:
            out("Loop buffers and Fourier transform in double precision (mixed mode)")
        out("Will%s use Multigrid" % ("" if coarse else " NOT"))
        out("Working with %s operators/fields" % ("coarse" if coarse else "fine"))
        out("Will%s perform Momentum Projection (Fourier Transform)" % ("" if p.doMomProj else " NOT"))
        if p.doMomProj:
            out("Momentum Projection will be performed on GPU using the split-K HIP kernel")
            out("Number of momenta: %d" % self.Nmom)
            out("Fourier transform Exp. Sign: %d" % int(p.FTSign))
        nonlocal_ = self.nDispEntries > 0
        out("Will%s perform loop on non-local currents" % ("" if nonlocal_ else " NOT"))
        if nonlocal_:
            out("Will perform ultra-local loop, plus the following %d displacement entries:" % self.nDispEntries)
            for i in range(self.nDispEntries):
                d, sgn, a, b, n, off = self.entry(i)
                name = ("+" if sgn == 1 else "-") + "xyzt"[d]
                if a == b:
                    out("  %d: %s with length %d, #loops = %d, loop-offset = %d" % (i, name, a, n, off))
                else:
                    out("  %d: %s with lengths from %d to %d, #loops = %d, loop-offset = %d" % (i, name, a, b, n, off))
        out("Total number of Loop Traces to perform: %d" % self.nLoop)
        out("Local  lattice size (x,y,z,t): %d %d %d %d " % self.localL)
        out("Global lattice size (x,y,z,t): %d %d %d %d " % self.totalL)
        out("Global time extent: %d" % self.totT)
        out("Local  time extent: %d" % self.locT)
        out("Local  volume: %d" % self.locV4)
        out("Local  3d volume: %d" % self.locV3)
        out("Global 3d volume: %d" % self.totV3)
        out("******************************************")

    def derivedFrom(self, idx):
        """After computeCoarseLoop: the entry that entry `idx` was reflected from (opposite sign, same direction and
        lengths), or -1 if it was computed from the eigenvectors."""
        return int(_lib.load().mugiq_hip_loop_entry_derived_from(self._handle, int(idx)))

    def ultraLocalCarrier(self):
        """After computeCoarseLoop: the entry whose pass over the eigenvectors also produced the ultra-local loop, or -1."""
        return int(_lib.load().mugiq_hip_loop_ultra_local_carrier(self._handle))

    def halosPackedInEntry(self):
        """After computeCoarseLoop: how many posted halos had their face layers written by the first entry itself
        (mugiq_hip_loop_halos_packed_in_entry) instead of by pack kernels beside it."""
        return int(_lib.load().mugiq_hip_loop_halos_packed_in_entry(self._handle))

    def setProfiling(self, on=True):
        """Bracket every phase of the next computeCoarseLoop with HIP events (mugiq_hip_loop_set_profiling)."""
        _lib.check(_lib.load().mugiq_hip_loop_set_profiling(self._handle, int(bool(on))))
        return self

    def phases(self):
        """[{kind, entry, ms, bytes}] of the last computeCoarseLoop, in issue order (empty unless setProfiling was on)."""
        lib = _lib.load()
        n = lib.mugiq_hip_loop_get_phases(self._handle, None, 0)
        if n <= 0:
            return []
        buf = (_CLoopPhase * n)()
        lib.mugiq_hip_loop_get_phases(self._handle, ctypes.cast(buf, ctypes.c_void_p), n)
        return [{"kind": PHASE_NAMES[b.kind], "entry": b.entry, "ms": b.ms, "bytes": b.bytes} for b in buf]

    def computeCoarseLoop(self):
        """lib/loop_mugiq.cpp:439-525"""
        _lib.check(_lib.load().mugiq_hip_loop_compute(self._handle))

    @property
    def dataPos_d(self):
        """device view [nLoop*16*V] complex of the position-space loop buffer"""
        ptr = _lib.load().mugiq_hip_loop_data_pos_d(self._handle)
        nbytes = self.nElemPosLoc * 2 * self.loopPrecision
        return device_bytes(ptr, nbytes, self.device).view(torch.complex128 if self.loopPrecision == 8 else torch.complex64)

    @property
    def dataPos(self):
        """host copy of the position-space loop buffer (dataPos of the reference, copied on request: lib/loop_mugiq.cpp:512)"""
        ptr = _lib.load().mugiq_hip_loop_data_pos_h(self._handle)
        if not ptr:
            raise _lib.MugiqHipError("dataPos: %s" % (_lib.load().mugiq_hip_last_error() or b"?").decode())
        ct = ctypes.c_double if self.loopPrecision == 8 else ctypes.c_float
        a = np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ct)), shape=(2 * self.nElemPosLoc,))
        return a.view(np.complex128 if self.loopPrecision == 8 else np.complex64).copy()

    @property
    def dataMom_bcast(self):
        """host copy [time-rank][im][iL][ig][t_loc] of the momentum-projected loops (lib/loop_mugiq.cpp:415-424)"""
        ptr = _lib.load().mugiq_hip_loop_data_mom_bcast_h(self._handle)
        if not ptr:
            return None
        ct = ctypes.c_double if self.loopPrecision == 8 else ctypes.c_float
        a = np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ct)), shape=(2 * self.nElemMomTot,))
        return a.view(np.complex128 if self.loopPrecision == 8 else np.complex64).copy()

    def dataMom_global(self):
        """dataMom_bcast rearranged to [Nmom][nLoop][16][totT]"""
        b = self.dataMom_bcast
        nt = self.totT // self.locT
        return b.reshape(nt, self.Nmom, self.nLoop, 16, self.locT).transpose(1, 2, 3, 0, 4).reshape(self.Nmom, self.nLoop, 16, self.totT)

    def writeLoopsHDF5(self):
        """lib/loop_mugiq.cpp:668-693 (momentum-space file; rank 0 writes)"""
        _lib.check(_lib.load().mugiq_hip_loop_write_hdf5(self._handle))

    def close(self):
        if self._handle:
            _lib.load().mugiq_hip_loop_destroy(self._handle)
            self._handle = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
