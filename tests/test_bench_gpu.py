"""GPU: bench.py's contract -- the one JSON line the driver parses -- on the single-GPU path, on the data-parallel path over the real
RCCL backend with the one rank a 1-GPU box can run, and through bench.py's own launcher (`--gpus 2` without torch.distributed.run
around it: two fresh ranks sharing cuda:0 over gloo, the debug mode of a 1-GPU box)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config")


def _run(args, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    return json.loads(lines[0])


def _check(out, n_gpus, steps, warmup, parallelism):
    for k in CONTRACT:
        assert k in out, k
    assert out["n_gpus"] == n_gpus and out["steps"] == steps and out["warmup"] == warmup
    assert out["unit"] == "frames/s" and out["higher_is_better"] is True and out["scaling"] == "weak" and out["vs_baseline"] is None
    assert out["dtype"] == "bf16" and out["data"] == "synthetic"
    assert out["config"]["parallelism"] == parallelism and out["config"]["global_batch"] == 32 * n_gpus and "workload" in out["config"]
    assert abs(out["value"] - 32 * n_gpus / (out["ms_per_step"] * 1e-3)) < 1e-6 * out["value"]
    assert out["roofline"]["bound"] == "mfma" and 0.05 < out["roofline"]["frac"] < 1.0
    assert out["final_loss"] == out["final_loss"]


def test_bench_single_gpu_line():
    out = _run(["--gpus", "1", "--steps", "20", "--warmup", "5", "--accuracy-steps", "0", "--no-cpu-baseline"])
    _check(out, 1, 20, 5, "single")
    assert 1.0 < out["ms_per_step"] < 50.0


def test_bench_rccl_path_with_one_rank():
    """The data-parallel path (process group, per-segment all-reduce over RCCL, 1/world in the optimizer) with the one rank this box has:
    same contract, and not slower than the plain path by more than the review's bound would tolerate by a wide margin (the tight
    comparison, <= 2 %, is profiles/r3_dist_overhead.json over 200 steps)."""
    seen = []
    for attempt in range(3):      # (two 40-step timings of two processes: one disturbed pair -- seen once in ~10 suite runs -- is measured again)
        single = _run(["--gpus", "1", "--steps", "40", "--warmup", "10", "--accuracy-steps", "0", "--no-cpu-baseline"])
        dp1 = _run(["--gpus", "1", "--force-dist", "--steps", "40", "--warmup", "10", "--accuracy-steps", "0", "--no-cpu-baseline"])
        _check(dp1, 1, 40, 10, "dp1")
        seen.append((dp1["ms_per_step"], single["ms_per_step"]))
        if dp1["ms_per_step"] < 1.10 * single["ms_per_step"]:
            break
    assert seen[-1][0] < 1.10 * seen[-1][1], seen


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it starts two fresh ranks itself (here: both on cuda:0 over gloo)."""
    out = _run(["--gpus", "2", "--same-device", "--dist-backend", "gloo", "--steps", "10", "--warmup", "3", "--accuracy-steps", "0", "--no-cpu-baseline"])
    _check(out, 2, 10, 3, "dp2")
