"""The C++ host-side mirror (include/mugiq_hip_operators.hpp) compiles against the C ABI (CPU check) and the C++
driver program tests/cpp/loop.cpp -- the computeLoop flow of the reference's tests/loop.cpp for configs[0] --
runs green on the GPU with the reference's own flag names."""
import os
import subprocess

import pytest

from util import ROOT

EXE = os.path.join(ROOT, "tests", "cpp", "loop_cpp_test")


def _build():
    cmd = ["/opt/rocm/bin/hipcc", "-std=c++17", "-O2", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cpp", "loop.cpp"), "-o", EXE,
           "-L", os.path.join(ROOT, "mugiq_amd"), "-lmugiq_hip", "-L", os.path.join(ROOT, "oracle"), "-lmugiq_oracle",
           "-Wl,-rpath," + os.path.join(ROOT, "mugiq_amd"), "-Wl,-rpath," + os.path.join(ROOT, "oracle")]
    subprocess.check_call(cmd)


def test_cpp_mirror_compiles_and_links(hip):
    if not os.path.exists(os.path.join(ROOT, "oracle", "libmugiq_oracle.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    _build()
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_cpp_loop_program_cfg1(hip, tmp_path):
    if not os.path.exists(EXE):
        _build()
    mom = tmp_path / "momenta.txt"
    mom.write_text("0 0 0\n1 0 0\n0 -1 1\n")
    h5 = tmp_path / "loop.h5"
    out = subprocess.run([EXE, "--dim", "8", "8", "8", "8", "--nev", "4", "--loop-ft-sign", "plus", "--loop-do-momproj", "yes",
                          "--momenta-filename", str(mom), "--loop-do-nonlocal", "yes", "--displace-entry-string", "+z:1,2;-t:1",
                          "--loop-calc-type", "opt", "--loop-write-mom-space", "yes", "--loop-mom-space-filename", str(h5), "--check"],
                         capture_output=True, text=True, timeout=300)
    print(out.stdout, out.stderr)
    assert out.returncode == 0 and "LOOP TEST PASSED" in out.stdout and "nLoop = 4" in out.stdout
    assert "reference loop nest through Displace" in out.stdout
    assert h5.exists() and h5.stat().st_size > 0
    bad = subprocess.run([EXE, "--loop-do-nonlocal", "yes", "--displace-entry-string", "+w:1"], capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and "Cannot parse given displacement string" in bad.stderr


@pytest.mark.gpu
def test_cpp_loop_program_partitioned_through_the_native_rccl_transport(hip, tmp_path):
    """The same program with QUDA's `--partition 12` (z and t): the C++ host creates the library's own RCCL transport
    (mugiq_hip_rccl_comm_create / _fill), the driver runs its partitioned path with the process as its own neighbour -- gauge
    borders and halos as ncclSend / ncclRecv -- and every slot still equals the reference loop nest run WITHOUT partitioning."""
    _build()
    mom = tmp_path / "momenta.txt"
    mom.write_text("0 0 0\n1 0 0\n0 -1 1\n")
    env = dict(os.environ, MUGIQ_HIP_SELF_HALO_COPY="1")      # the halos travel (default for a self-neighbour: packed in place)
    out = subprocess.run([EXE, "--dim", "8", "8", "8", "8", "--nev", "4", "--loop-ft-sign", "plus", "--loop-do-momproj", "yes",
                          "--momenta-filename", str(mom), "--loop-do-nonlocal", "yes", "--displace-entry-string", "+z:1,3;-t:1,2;+x:1",
                          "--loop-calc-type", "opt", "--partition", "12", "--check"], capture_output=True, text=True, timeout=300, env=env)
    print(out.stdout, out.stderr)
    assert out.returncode == 0 and "LOOP TEST PASSED" in out.stdout and "through the library's RCCL transport" in out.stdout
    assert "reference loop nest through Displace" in out.stdout
