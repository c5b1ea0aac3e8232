"""bench.py's launcher logic where there is no GPU: `--gpus N` typed without a launcher must refuse cleanly before it starts
anything (RCCL needs one device per rank), and the plain call must say that it needs an MI355X -- never a fallback, never a
traceback.  (With GPUs the self-launch itself is exercised on the GPU box: profiles/r03_bench_gloo2_selflaunch.json.)"""
import json
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


# ---- the printed line (no GPU needed: the record is canned) ----------------------------------------------------------------
def _strict_loads(text):
    def refuse(tok):
        raise ValueError("non-finite constant %s in the bench line" % tok)
    return json.loads(text, parse_constant=refuse)


def _canned_record():
    """round 3's full bench record (20 KB on one line: the driver's 8 KB stdout tail cut it, BENCH_r03.json parsed = null) plus
    the keys this round adds and a non-finite number."""
    out = json.load(open(os.path.join(ROOT, "profiles", "r03_bench.json")))
    out["cpu_baseline"]["sample_short"] = "65536 sites x 200 eigenvectors of the bench fields, 23 passes in 12.1 s, oracle/mugiq_oracle.c + OpenMP"
    out["strong_scaling"].update({"global_lattice": [48, 48, 48, 96], "n_ev": 48, "grid": [1, 1, 2, 4], "seconds_1gpu": 0.168, "speedup": 5.9,
                                  "max_rel_diff_vs_1gpu": 3.1e-15, "halo_GBps_per_rank": float("nan")})
    out["partitioned"] = {"workload": "w" * 300, "local_lattice": [48, 48, 24, 24], "n_ev": 400, "seconds": 0.2, "sites_per_s_all_slots": 5.3e7,
                          "halo_bytes_sent_per_rank": 25480396800, "halo_GBps_per_rank": 61.0, "wait_ms_not_hidden": 3.0}
    out["also_measured"]["forced_partition_displaced_loops"]["max_rel_diff_forced_vs_unpartitioned"] = 2.2e-15
    out["also_measured"]["forced_partition_displaced_loops"]["parity_ok"] = True
    out["also_measured"]["error"] = "extra leg 'x' did not finish"
    out["parity_ok"] = True
    out["nccl_ranks"], out["process_grid"], out["backend"] = 8, [1, 1, 2, 4], "nccl"
    return out


def test_bench_line_is_short_strict_json():
    sys.path.insert(0, ROOT)
    import bench
    out = _canned_record()
    assert len(json.dumps(out)) > 15000                     # the record itself is what broke round 3
    text = bench.compact_line(out, "gpurun_out/bench_detail_n1.json")
    assert "\n" not in text and len(text) < bench.LINE_LIMIT <= 3000, len(text)
    line = _strict_loads(text)
    assert text.startswith('{"metric"')
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert line[k] == pytest.approx(out[k], rel=1e-5) if isinstance(out[k], float) else line[k] == out[k], k
    assert line["config"]["workload"] == out["config"]["workload"] and "model" not in line["config"]
    r = line["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-4) and "traffic" in r and r["kernel_ms"] > 0
    c = line["cpu_baseline"]
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(c) and c["kind"] == "port" and len(c["sample"]) < 200
    s = line["strong_scaling"]
    assert s["speedup"] == pytest.approx(5.9) and s["seconds_1gpu"] == pytest.approx(0.168) and s["halo_GBps_per_rank"] is None   # NaN -> null
    assert line["partitioned"]["n_ev"] == 400 and "workload" not in line["partitioned"]
    assert line["legs"]["forced_partition_displaced_loops"]["parity_ok"] is True
    assert line["detail_file"].endswith("bench_detail_n1.json") and line["nccl_ranks"] == 8


def test_bench_line_sheds_the_legs_rather_than_grow():
    sys.path.insert(0, ROOT)
    import bench
    out = _canned_record()
    for i in range(200):
        out["also_measured"]["leg_%d" % i] = {"seconds": 1.0, "roofline": {"bound": "hbm", "frac": 0.5, "kernel_ms": 1.0}}
    text = bench.compact_line(out, "d.json")
    line = _strict_loads(text)
    assert len(text) < bench.LINE_LIMIT and "legs" not in line and "roofline" in line and "cpu_baseline" in line


# ---- inputs that do not depend on the process grid (the strong-scaling leg compares N ranks with one) ------------------------
def test_hashed_inputs_are_functions_of_the_global_site():
    sys.path.insert(0, ROOT)
    import numpy as np
    import bench
    G = (4, 4, 8, 8)
    whole = bench.global_site_index(G, (1, 1, 1, 1), (0, 0, 0, 0), "cpu")
    assert sorted(whole.reshape(-1).tolist()) == list(range(int(np.prod(G))))
    # QUDA's checkerboard: (parity, x_cb) <-> full index 2 x_cb + ((parity + y + z + t) & 1) in the x-fastest lexicographic order
    i2 = 2 * np.arange(int(np.prod(G)) // 2)
    y, z, t = (i2 // G[0]) % G[1], (i2 // (G[0] * G[1])) % G[2], i2 // (G[0] * G[1] * G[2])
    for par in range(2):
        assert np.array_equal(whole[par].numpy(), i2 + ((par + y + z + t) & 1))
    u_whole = bench.random_su3_eo(G, "cpu", 7, whole)
    grid = (1, 1, 2, 2)
    X = tuple(G[d] // grid[d] for d in range(4))
    seen = []
    for cz in range(2):
        for ct in range(2):
            g = bench.global_site_index(X, grid, (0, 0, cz, ct), "cpu")
            seen += g.reshape(-1).tolist()
            # the local links equal the global field's at the same global sites
            u = bench.random_su3_eo(X, "cpu", 7, g)
            pos = {int(v): i for i, v in enumerate(whole.reshape(-1).tolist())}
            idx = [pos[int(v)] for v in g.reshape(-1).tolist()]
            assert np.array_equal(u.numpy(), u_whole.numpy()[:, idx])
            # local parity = global parity (even local extents)
            for par in range(2):
                gg = g[par].numpy()
                c = [gg % G[0], (gg // G[0]) % G[1], (gg // (G[0] * G[1])) % G[2], gg // (G[0] * G[1] * G[2])]
                assert np.all((c[0] + c[1] + c[2] + c[3]) % 2 == par)
    assert sorted(seen) == list(range(int(np.prod(G))))
    det = np.linalg.det(u_whole.numpy().reshape(-1, 3, 3))
    assert np.allclose(det, 1.0, atol=1e-12)
