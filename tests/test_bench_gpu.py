"""bench.py end to end: the JSON contract of the line on one GPU, and the N > 1 control flow (self-launch, the two
measurement passes, barriers, per-rank gathers) with two ranks sharing the one GPU of the test box over gloo
(PCOPS_BENCH_SHARED_GPU=1 -- the value of such a line means nothing and the line says so)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(extra, env=None):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--batch", "16",
           "--no-cpu-baseline", "--no-extras"] + extra
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, **(env or {})))
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-500:]
    return json.loads(lines[0])


def test_bench_line_contract():
    d = _bench([])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "kernels"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak" and d["dtype"] == "f32"
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "launches"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and 0 < r["frac"] < 1.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["launches"] == 3                      # the dominant kernel is bracketed inside the timed region: K launches
    assert abs(d["value"] - 16 * 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-6


def test_bench_two_ranks_control_flow():
    d = _bench(["--gpus", "2"], env={"PCOPS_BENCH_SHARED_GPU": "1"})
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and len(d["per_rank_clouds_per_s"]) == 2
    assert d["config"]["global_batch"] == 32 and d["config"]["parallelism"] == "dp2"
    assert d["config"]["shared_gpu_debug"] is True
    assert d["allreduce_ms_per_step"] > 0
    # the N = 1 point of the scaling curve from the SAME invocation (rank 0 alone, the other ranks parked)
    solo = d["single_gpu_same_invocation"]
    assert solo["n_gpus"] == 1 and solo["value"] > 0 and abs(solo["value"] - 16 * 3 / (solo["ms_per_step"] * 3e-3)) / solo["value"] < 1e-6


def test_bench_refuses_a_world_size_it_was_not_asked_for():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert p.returncode != 0 and "refusing" in (p.stderr + p.stdout)
