"""Host-side mirror of MuGiq's `Displace<F, order>` class (include/displace.h:13-80, lib/displace.cpp) over the C ABI.

The reference keeps this class private to `Loop_Mugiq` (a friend); the driver behind `mugiq_hip_loop_compute` carries the
same state machine in C++ (csrc/loop_driver.cpp, BASIC plan).  This mirror exists so that the reference's own loop nest
(lib/loop_mugiq.cpp:455-509) can be written down call for call -- setupDisplacement / doVectorDisplacement /
performLoopContraction -- against libmugiq_hip.so; tests/test_gpu_driver.py does exactly that and compares with the
driver and the oracle.
"""
from .fields import SpinorField, GaugeField
from .loop import parseDisplacement
from .operators import (performCovariantDisplacementVector, exchangeGhostVec, DisplaceFlagArray, DispSignMinus, DispSignPlus)
from ._lib import MugiqHipError

DISPLACE_TYPE_COVARIANT = 0          # include/enum_mugiq.h:46
DisplaceTypeArray = ["Covariant"]    # include/displace.h:22
DisplaceDirArray = ["x", "y", "z", "t"]
DisplaceSignArray = ["-", "+"]
DispFlagNone = DispDirNone = DispSignNone = -0x7fffffff - 1   # MUGIQ_INVALID_ENUM (INT_MIN)


class Displace:
    """Displace(loopParams, csf, coarsePrec)                                     lib/displace.cpp:4-31

    loopParams.gauge: either a border-extended device `GaugeField` (already what createExtendedCudaGaugeField builds), or
    loopParams.gauge_qdp = the four host QDP link arrays of the local lattice (`MugiqLoopParam::gauge[4]`,
    include/mugiq.h:44), from which the extended field is built with borders 2 * commDimPartitioned(d) exactly like
    lib/displace.cpp:16,104-134.  `comm` (a GridComm) stands for QUDA's communicator; None = one process."""

    redundantComms = False            # include/displace.h:43

    def __init__(self, loopParams, csf, coarsePrec=None, comm=None, verbose=None):
        self.dispString = ""
        self.dispFlag, self.dispDir, self.dispSign = DispFlagNone, DispDirNone, DispSignNone
        self.comm = comm
        self.commDim = [int(bool(comm is not None and comm.comm_dim_partitioned(d))) for d in range(4)]
        self.exRng = [2 * int(self.redundantComms or c) for c in self.commDim]
        self._say = verbose or (lambda line: None)
        g = getattr(loopParams, "gauge", None)
        if isinstance(g, GaugeField):
            self.gaugeField = g
        else:
            qdp = getattr(loopParams, "gauge_qdp", None)
            if qdp is None:
                raise MugiqHipError("Displace: loopParams carries neither an extended GaugeField nor host QDP links")
            self.gaugeField = GaugeField(csf.X, self.exRng, csf.precision, device=csf.device).set_from_qdp_host(qdp, comm)
        self._say("Displace: Gauge field has%s extended Halo exchange" % ("" if any(self.gaugeField.R) else " NOT"))
        # auxDispVec: zero field with csf's geometry (csParam.create = QUDA_ZERO_FIELD_CREATE, precision = coarsePrec_)
        self.auxDispVec = SpinorField(csf.X, coarsePrec or csf.precision, csf.order, csf.stride - csf.volumeCB, csf.device)

    # ---- string -> flag -> (dir, sign)                                          lib/displace.cpp:137-223
    def WhichDisplaceFlag(self):
        if self.dispString not in DisplaceFlagArray:
            raise MugiqHipError("WhichDisplaceFlag: Cannot parse given displacement string = %s." % self.dispString)
        return DisplaceFlagArray.index(self.dispString)

    def WhichDisplaceDir(self):
        return self.dispFlag // 2

    def WhichDisplaceSign(self):
        return DispSignPlus if self.dispFlag % 2 == 0 else DispSignMinus

    def setupDisplacement(self, dStr):
        self.dispString = dStr
        self.dispFlag = self.WhichDisplaceFlag()
        self.dispDir, self.dispSign = self.WhichDisplaceDir(), self.WhichDisplaceSign()
        d, s = parseDisplacement(dStr)            # the library's own table must agree (mugiq_hip_parse_displacement)
        assert (d, s) == (self.dispDir, self.dispSign)
        self._say("setupDisplacement: Displacement(s) will take place in the %s%s direction"
                  % (DisplaceSignArray[self.dispSign], DisplaceDirArray[self.dispDir]))

    # ---- vectors                                                                lib/displace.cpp:40-67
    def resetAuxDispVec(self, fineEvec):
        self.auxDispVec.data.copy_(fineEvec.data)

    def swapAuxDispVec(self, displacedEvec):
        # `tmp` aliases displacedEvec in the reference, so after the first copy the second one is the identity
        displacedEvec.data.copy_(self.auxDispVec.data)

    def exchangeGhostVec(self, v):
        """exchangeGhostVec (lib/contract_wrappers.cu:166-169) through the library's twin mugiq_hip_exchange_ghost_vec"""
        exchangeGhostVec(v, self.comm)

    def doVectorDisplacement(self, dispType, displacedEvec, idisp):
        if dispType != DISPLACE_TYPE_COVARIANT:
            raise MugiqHipError("Unsupported Displacement type %d" % int(dispType))
        if self.dispDir == DispDirNone:
            raise MugiqHipError("doVectorDisplacement: Got invalid dispDir and/or dispSign.")
        self.auxDispVec.data.zero_()
        self.exchangeGhostVec(displacedEvec)
        performCovariantDisplacementVector(self.auxDispVec, displacedEvec, self.gaugeField, self.dispDir, self.dispSign, self.commDim)
        self.swapAuxDispVec(displacedEvec)
        self._say("doVectorDisplacement: Step-%02d of a Covariant displacement done" % idisp)
