"""mugiq_amd -- MI355X-native (HIP / gfx950) drop-in for the disconnected-loop hot path of ckallidonis/mugiq.

The product is libmugiq_hip.so (C ABI in include/mugiq_hip.h).  This package is the Python host-side mirror
of MuGiq's operator interface over that ABI; importing it loads the library and fails loudly if it is missing.
"""
from . import _lib

_lib.load()

from .fields import SpinorField, GaugeField, CoarseField, Transfer, FLOAT2, FLOAT4  # noqa: E402
from .operators import (  # noqa: E402
    copyGammaCoeffStructToSymbol, copyGammaMapStructToSymbol, gammaTables, GammaName,
    performLoopContraction, performLoopContractionBatched, performCovariantDisplacementVector, packFace, exchangeGhostVec,
    createPhaseMatrixGPU, convertIdxOrder_mapGamma, momentumProjection, momentumProjectionSeparable, convertAndProject, packFaceLayers, displacedLoopContractionFused, reflectDisplacedLoop, packLoopLayers, probeReadBandwidth, prolongateEvecs, prolongateCoarseEvecs, prolongateContractBatched,
    DispDir, DispSignMinus, DispSignPlus, LOOP_FT_SIGN_MINUS, LOOP_FT_SIGN_PLUS, DisplaceFlagArray,
)
from ._lib import MugiqHipError, LIB_PATH  # noqa: E402
from .loop import (  # noqa: E402
    MugiqLoopParam, Loop_Mugiq, parseDisplaceEntryString, parseDisplacement, read_momenta_file, writeLoopsHDF5_Mom, reflectMomentumSpace,
    LOOP_CALC_TYPE_BLAS, LOOP_CALC_TYPE_OPT_KERNEL, LOOP_CALC_TYPE_BASIC_KERNEL,
)
from .comm import GridComm, RcclComm  # noqa: E402
from .displace import Displace, DISPLACE_TYPE_COVARIANT  # noqa: E402

__all__ = [n for n in dir() if not n.startswith("_")]
