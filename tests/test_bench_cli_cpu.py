"""bench.py's launcher logic where there is no GPU: `--gpus N` typed without a launcher must refuse cleanly before it starts
anything (RCCL needs one device per rank), and the plain call must say that it needs an MI355X -- never a fallback, never a
traceback.  (With GPUs the self-launch itself is exercised on the GPU box: profiles/r03_bench_gloo2_selflaunch.json.)"""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.skipif(torch.cuda.device_count() > 0, reason="a GPU is visible: bench.py would run")


def _run(*args):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, env=env, timeout=300)


def test_gpus_n_without_devices_refuses_before_launching():
    r = _run("--gpus", "2")
    assert r.returncode != 0 and "only 0 HIP device(s) visible" in r.stderr and "Traceback" not in r.stderr
    assert r.stdout.strip() == ""


def test_no_device_no_fallback():
    r = _run()
    assert r.returncode != 0 and "no HIP device is visible (there is no CPU fallback)" in r.stderr and "Traceback" not in r.stderr
