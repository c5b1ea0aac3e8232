"""The reference-side binding (include/mugiq_hip_quda_adapter.hpp: the six operator templates of the reference's
include/loop_mugiq.h:280-311 / include/displace.h:109-111, both computeLoop signatures of include/mugiq.h:79-81, the MPI
transport, the layout self-check) goes through a compiler: `-fsyntax-only` against declaration-only stand-ins for the QUDA, MPI
and MuGiq names it touches (tests/quda_stub/README.md).  This catches template, signature and C-ABI call errors; it proves
nothing about layouts -- that is what layoutSelfCheck() is for, under a real QUDA.  CPU test."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = os.path.join(ROOT, "tests", "quda_stub")
REF_INC = "/root/reference/include"


def _compiler():
    for c in ("/opt/rocm/lib/llvm/bin/clang++", shutil.which("g++"), shutil.which("clang++")):
        if c and os.path.exists(c):
            return c
    pytest.skip("no C++ compiler")


def _syntax_only(param_dir, extra=()):
    cmd = [_compiler(), "-std=c++17", "-fsyntax-only", "-x", "c++", "-D__HIP_PLATFORM_AMD__", "-Wall",
           "-I", os.path.join(ROOT, "include"), "-I", os.path.join(STUB, "mugiq_classes"), "-I", param_dir,
           "-I", os.path.join(STUB, "quda"), "-I", "/opt/rocm/include", *extra, os.path.join(STUB, "adapter_tu.cpp")]
    return subprocess.run(cmd, capture_output=True, text=True)


def test_adapter_compiles_against_the_stub():
    r = _syntax_only(os.path.join(STUB, "mugiq_params"))
    assert r.returncode == 0, r.stderr[-4000:]


def test_the_syntax_check_can_fail():
    """the check is alive: an undeclared C-ABI entry point in the same translation unit is an error"""
    r = _syntax_only(os.path.join(STUB, "mugiq_params"), extra=["-include", os.path.join(STUB, "break_on_purpose.h")])
    assert r.returncode != 0 and "mugiq_hip_no_such_entry_point" in r.stderr


@pytest.mark.skipif(not os.path.exists(os.path.join(REF_INC, "mugiq.h")), reason="the reference tree is not mounted here")
def test_adapter_compiles_against_the_reference_param_headers():
    """MugiqLoopParam, the enums and the declaration of computeLoop<Float> from the reference's OWN include/mugiq.h and
    include/enum_mugiq.h (read where they lie): convert() names real members, the definitions match the declaration."""
    r = _syntax_only(REF_INC)
    assert r.returncode == 0, r.stderr[-4000:]
    # and the reference's headers were the ones seen, not the stand-ins
    h = subprocess.run([_compiler(), "-std=c++17", "-fsyntax-only", "-x", "c++", "-H", "-D__HIP_PLATFORM_AMD__",
                        "-I", os.path.join(ROOT, "include"), "-I", os.path.join(STUB, "mugiq_classes"), "-I", REF_INC,
                        "-I", os.path.join(STUB, "quda"), "-I", "/opt/rocm/include", os.path.join(STUB, "adapter_tu.cpp")],
                       capture_output=True, text=True)
    assert REF_INC + "/mugiq.h" in h.stderr and REF_INC + "/enum_mugiq.h" in h.stderr
    assert "mugiq_params" not in h.stderr
