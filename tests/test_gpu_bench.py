"""-m gpu: the bench.py contract - one JSON line with the driver's keys, the three roofline objects timed with HIP events,
and (here with a reduced sample) the CPU-oracle baseline."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(900)
def test_bench_json_contract():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "2",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=800, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["metric"].startswith("train patches/sec") and d["unit"] == "patches/s" and d["n_gpus"] == 1
    assert d["steps"] == 3 and d["warmup"] == 2 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"] and d["config"]["global_batch"] == 32
    assert abs(d["value"] - 32 * 1e3 / d["ms_per_step"]) < 0.01 * d["value"] and d["value"] > 100
    for name in ("roofline", "roofline_core_unfused", "roofline_dominant"):
        ro = d[name]
        assert ro["bound"] in ("hbm", "mfma") and ro["unit"] in ("GB/s", "TFLOP/s")
        assert abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-3 and 0.05 < ro["frac"] < 1.0
        assert ro["launches"] > 0 and ro["avg_launch_us"] > 0
        assert ro["traffic"] is None or ro["traffic"] > 0
    assert "cpu_baseline" not in d


def test_bench_two_ranks_strong_scaling():
    """The driver's multi-GPU launch line on a ONE-GPU box: two ranks share the device and reduce over gloo (test-only
    switches DHZ_DIST_BACKEND / DHZ_SHARE_GPU; the real runs use RCCL, one rank per GPU).  --strong splits the global
    batch of 32 over the ranks (SURVEY 8d asks for both scalings); rank 0 alone prints the line."""
    env = dict(os.environ, DHZ_DIST_BACKEND="gloo", DHZ_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--strong",
           "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["global_batch"] == 32
    assert d["config"]["parallelism"] == "dp2" and d["value"] > 50
    assert abs(d["value"] - 32 * 1e3 / d["ms_per_step"]) < 0.01 * d["value"]


def test_bench_two_ranks_rccl():
    """bench.py --gpus 2 exactly as the driver launches it - one rank per GPU, RCCL ("nccl") over xGMI - whenever the box has
    two devices; on a one-GPU box the RCCL path cannot run and the exchange plan (bucket sizes, backward order, dead
    parameters, predicted ring time) is what tests/test_ddp_gloo.py pins instead."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one visible GPU: RCCL needs two devices (plan checked on the CPU in tests/test_ddp_gloo.py)")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("DHZ_DIST_BACKEND", None); env.pop("DHZ_SHARE_GPU", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29534", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2",
           "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["global_batch"] == 64


@pytest.mark.timeout(900)
def test_bench_bf16_contract_line():
    """bench.py --dtype bf16 (BASELINE config 4's recipe at a reduced size here: E = 64, 128 x 128, two patches): the contract line
    with dtype "bf16" and the bf16-GEMM roofline object against the dense bf16 matrix peak."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "2", "--embed_dim", "64",
                        "--ps", "128", "--batch", "2", "--dtype", "bf16", "--no-cpu-baseline"], capture_output=True, text=True,
                       timeout=800, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert d["dtype"] == "bf16" and d["metric"] == "train patches/sec (128x128, embed_dim=64)" and d["config"]["global_batch"] == 2
    ro = d["roofline"]
    assert "bf16" in ro["kernel"] and ro["peak"] == 2500.0 and ro["unit"] == "TFLOP/s" and 0 < ro["frac"] < 1 and ro["launches"] > 0


def test_bench_headline_is_the_default_six_term_arithmetic_with_the_fp32_pipe_beside_it():
    """bench.py: the headline runs the product's default arithmetic (six-term bf16 split: fp32 error class) and SAYS so
    (dtype f32 + config.arithmetic), and carries the same step on the fp32 matrix pipe as the `fp32_pipe` object, measured in the
    same process; DHZ_SPLIT_BF16=0 makes the fp32 pipe the headline (no second object)."""
    env = {k: v for k, v in os.environ.items() if k != "DHZ_SPLIT_BF16"}
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "2", "--batch", "4",
            "--no-cpu-baseline", "--no-kernel-timing"]
    r0 = subprocess.run(base, capture_output=True, text=True, timeout=800, cwd=ROOT, env=env)
    assert r0.returncode == 0, r0.stderr[-2000:]
    d0 = json.loads([ln for ln in r0.stdout.splitlines() if ln.startswith("{")][0])
    assert d0["dtype"] == "f32" and d0["config"]["split_terms"] == 6
    assert d0["config"]["arithmetic"] == "fp32 storage/accumulate; products 6xbf16 MFMA, dropped <= 2^-24"
    fp = d0["fp32_pipe"]
    assert fp["value"] > 0 and fp["ms_per_step"] > 0 and "fp32 matrix pipe" in fp["arithmetic"]
    r1 = subprocess.run(base, capture_output=True, text=True, timeout=800, cwd=ROOT, env=dict(env, DHZ_SPLIT_BF16="0"))
    assert r1.returncode == 0, r1.stderr[-2000:]
    d1 = json.loads([ln for ln in r1.stdout.splitlines() if ln.startswith("{")][0])
    assert d1["config"]["split_terms"] == 0 and "fp32_pipe" not in d1 and "fp32 matrix pipe" in d1["config"]["arithmetic"]
    # same seed, same batch, same steps: the two arithmetics agree to the run-to-run noise of the atomic summation order
    # amplified over five training steps
    assert abs(d0["config"]["loss_last_step"] - d1["config"]["loss_last_step"]) < 1e-2 * abs(d0["config"]["loss_last_step"])
    assert fp["loss_last_step"] > 0          # (measured after the headline's steps: further along the same training run)
    assert d0["config"]["workload"] == d1["config"]["workload"]
