"""CPU-only checks of the drop-in boundary: the C-ABI library loads, exports every symbol include/*.h
declares, and its host-side logic (tables, validation, layout plumbing) agrees with the oracle.
No compute call is made here (there is no GPU in the build container)."""
import ctypes
import os
import re

import numpy as np
import pytest

from util import orc, ROOT


@pytest.fixture(scope="module")
def lib(hip):
    return hip._lib.load()


def _declared_symbols():
    names = set()
    inc = os.path.join(ROOT, "include")
    for fn in os.listdir(inc):
        if fn.endswith(".h"):
            src = open(os.path.join(inc, fn)).read()
            src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
            names |= set(re.findall(r"\b(mugiq_hip_[a-z0-9_]+)\s*\(", src))
    return names


def test_library_exports_every_declared_symbol(hip, lib):
    declared = _declared_symbols()
    assert len(declared) >= 14
    for name in declared:
        assert hasattr(lib, name), "include/*.h declares %s but the library does not export it" % name
    # and the Python binding table covers the same set
    assert set(hip._lib.SIGNATURES) == declared


def test_quda_adapter_uses_only_declared_entry_points_and_descriptor_members(hip):
    """include/mugiq_hip_quda_adapter.hpp cannot be compiled here (it needs QUDA's and MuGiq's headers); at least every C-ABI
    entry point it calls is declared + exported, every descriptor member it fills exists in the ctypes mirror of the structs,
    and it defines the reference's operator names and both computeLoop signatures."""
    src = open(os.path.join(ROOT, "include", "mugiq_hip_quda_adapter.hpp")).read()
    code = re.sub(r"//.*", "", src)
    called = set(re.findall(r"\b(mugiq_hip_[a-z0-9_]+)\s*\(", code)) - {"mugiq_hip_adapter"}
    assert called and called <= set(hip._lib.SIGNATURES), called - set(hip._lib.SIGNATURES)
    members = {"MugiqHipSpinorField": hip._lib.SpinorDesc, "MugiqHipGaugeField": hip._lib.GaugeDesc,
               "MugiqHipCoarseField": hip._lib.CoarseDesc, "MugiqHipTransfer": hip._lib.TransferDesc}
    for m in re.finditer(r"inline (MugiqHip\w+) describe\w*\([^)]*\) \{(.*?)\n\}", code, flags=re.S):
        fields = {f[0] for f in members[m.group(1)]._fields_}
        var = re.search(r"%s (\w+)\{\};" % m.group(1), m.group(2)).group(1)
        used = set(re.findall(r"\b%s\.(\w+)" % var, m.group(2)))
        assert used and used <= fields, (m.group(1), used - fields)
    for name in ("copyGammaCoeffStructToSymbol", "copyGammaMapStructToSymbol", "createPhaseMatrixGPU", "performLoopContraction",
                 "convertIdxOrder_mapGamma", "performCovariantDisplacementVector"):           # include/loop_mugiq.h:280-311, displace.h:109-111
        assert re.search(r"\bvoid %s\(" % name, code), name
    assert "void computeLoop(MugiqLoopParam loopParams, Eigsolve_Mugiq *eigsolve)" in code            # lib/interface_mugiq.cpp:158
    assert re.search(r"void computeLoop\(QudaMultigridParam mgParams, QudaEigParam QudaEigParams, MugiqLoopParam loopParams, MuGiqBool computeCoarse, MuGiqBool useMG\)", code)


def test_version_and_device_count(lib):
    assert lib.mugiq_hip_version() == 100
    assert lib.mugiq_hip_device_count() >= 0


def test_gamma_tables_match_reference_restatement(hip):
    rv, ci, ms, mi = hip.gammaTables()
    assert np.array_equal(rv, orc.GAMMA_ROW_VALUE)
    assert np.array_equal(ci, orc.GAMMA_COLUMN_INDEX)
    assert np.array_equal(ms, orc.gamma_map_sign())
    assert list(mi) == orc.INDEX_MAP_GAMMA
    assert [hip.GammaName(m) for m in range(16)] == orc.GAMMA_NAMES
    with pytest.raises(IndexError):
        hip.GammaName(16)


def test_gamma_symbol_copies_validate_precision(hip):
    hip.copyGammaCoeffStructToSymbol(8)
    hip.copyGammaMapStructToSymbol(4)
    with pytest.raises(hip.MugiqHipError):
        hip.copyGammaCoeffStructToSymbol(2)


def test_host_side_validation_without_gpu(hip, lib):
    """Preconditions the reference checks with errorQuda are rejected before any HIP call."""
    X = (ctypes.c_int * 4)(4, 4, 4, 4)
    st = lib.mugiq_hip_convert_idx_order_map_gamma(ctypes.c_void_p(16), ctypes.c_void_p(32), 17, 1, 2, 128, X, 8, None)
    assert st == 1 and b"nData = nLoop * NGamma" in lib.mugiq_hip_last_error()      # lib/contract_wrappers.cu:138
    d = hip._lib.SpinorDesc()
    d.data = 64
    d.precision, d.field_order, d.nParity, d.volumeCB, d.stride, d.parity_offset = 8, 2, 1, 128, 128, 12 * 128
    for i in range(4):
        d.X[i] = 4
    st = lib.mugiq_hip_perform_loop_contraction(ctypes.c_void_p(16), ctypes.byref(d), ctypes.byref(d), 1.0, None)
    assert st == 1 and b"Full Site Subset" in lib.mugiq_hip_last_error()             # lib/contract_wrappers.cu:100
    d.nParity = 2
    d.X[0] = 3
    st = lib.mugiq_hip_perform_loop_contraction(ctypes.c_void_p(16), ctypes.byref(d), ctypes.byref(d), 1.0, None)
    assert st == 1 and b"even" in lib.mugiq_hip_last_error()
    assert lib.mugiq_hip_momentum_projection_workspace(0, 16, 64, 3, 8) == 0
    assert lib.mugiq_hip_momentum_projection_workspace(32, 16, 32768, 123, 8) > 0


@pytest.mark.parametrize("order", [2, 4])
def test_python_layout_plumbing_matches_oracle_layout(hip, order):
    """fields.spinor_native_index is the host-side data-format glue of the product; pin it to the oracle's."""
    from mugiq_amd.fields import spinor_native_index
    rng = np.random.default_rng(1)
    for _ in range(200):
        p, x, s, c = rng.integers(0, 2), rng.integers(0, 500), rng.integers(0, 4), rng.integers(0, 3)
        assert spinor_native_index(order, p, x, s, c, 512, 12 * 512) == orc.spinor_native_index(order, p, x, s, c, 512, 12 * 512)


def test_missing_library_fails_loudly(hip, monkeypatch, tmp_path):
    """No CPU fallback: if libmugiq_hip.so is not there, loading raises instead of degrading."""
    from mugiq_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libmugiq_hip.so"))
    with pytest.raises(ImportError, match="no CPU fallback"):
        _lib.load()


def test_product_does_not_import_the_oracle():
    """Only tests/, smoke() and bench.py's cpu_baseline may touch oracle/."""
    pkg = os.path.join(ROOT, "mugiq_amd")
    for dirpath, _, files in os.walk(pkg):
        if "build" in dirpath:
            continue
        for fn in files:
            if fn.endswith((".py", ".cpp", ".hip", ".h", "Makefile")):
                src = open(os.path.join(dirpath, fn), errors="ignore").read()
                assert "import oracle" not in src and "from oracle" not in src and "mugiq_oracle" not in src, os.path.join(dirpath, fn)
