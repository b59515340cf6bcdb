#!/usr/bin/env python3
"""bench.py - training patches/sec of the Uformer_ProbSparse path on MI355X (BASELINE.json metric).

A "step" is one full training step of BASELINE configs[1] on one per-GPU batch of synthetic haze pairs
that are already resident in HBM: forward (ProbSparse Uformer, E=32, ps=128, per-GPU bs=32, fp32),
clamp + Charbonnier + VGG19 contrastive loss, backward, bucketed RCCL gradient all-reduce (N>1) and the
AdamW update.  Nothing is skipped or cached inside the timed region.

  python bench.py --gpus 1 --steps 20 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline     - the fused window-attention kernel (dhz_fused_window_attn_fwd / _fwd6), timed live with HIP events on the stream
                 it is launched on; algorithmic work per DESIGN.md §9
  cpu_baseline - the CPU oracle's training step timed on this node's host cores on a bounded sample
                 (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "research-and-implementation-of-image-dehazing-algorithm-based-on-vision-transformer_amd")
for p in (PKG, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F32_PEAK_TF = 157.3       # fp32-input MFMA dense peak
MFMA_BF16_PEAK_TF = 2500.0     # bf16 MFMA dense peak (MI355X_MICROARCH.md: ~2.5 PF dense)


def _cpu_steps(P, params, opt, hazy, gt, vggW, w_cr, steps, warm=1):
    """median seconds per step over `steps` timed steps after `warm` untimed ones"""
    from oracle import uformer_oracle as O
    times = []
    for i in range(steps + warm):
        t0 = time.perf_counter()
        opt.zero_grad()
        loss, _ = O.train_step_loss(P, hazy, gt, w_char=1.0, w_cr=w_cr, vggW=vggW, training=True)
        loss.backward()
        opt.step()
        if i >= warm:
            times.append(time.perf_counter() - t0)
    times.sort()
    return times[len(times) // 2]


CPU_STEPS = 5            # >= 5 timed steps after 1 warm-up, at one thread and at every core


def cpu_baseline(bs=2):
    """CPU oracle (kind 'port') on this node's host cores, bounded sample (>= 5 timed steps after one warm-up
    step, at 1 thread and at the many-thread setting).  `value` is the HEADLINE's recipe - BASELINE configs[1]: E=32, ps=128, fp32,
    Charbonnier + VGG19 contrastive loss, AdamW - at 8 threads (the survey container's core count; the GPU hosts of this pool are
    shared 256-thread machines, where an every-core run measures the neighbours).  Side fields: the same recipe at 1 thread, the
    config-1 recipe (Charbonnier only: what the reference's own CPU path was timed with in the survey container, 1.98 patches/s at 8
    threads, 0.40 at 1 thread) at 8 and 1 threads, the every-core figure (2 timed steps) with the load average."""
    from oracle import uformer_oracle as O
    import My_model_1 as M1
    from dehaze_hip.train import synthetic_batch
    torch.manual_seed(1234)
    model = M1.Uformer(img_size=128, embed_dim=32, win_size=8, token_projection='linear', token_mlp='leff')
    P = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point) for k, v in model.state_dict().items()}
    params = [P[n] for n, _ in model.named_parameters()]
    opt = torch.optim.AdamW(params, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.02)
    gt, hazy = synthetic_batch(bs, 128, seed=99)
    cores = torch.get_num_threads()
    load0 = os.getloadavg() if hasattr(os, "getloadavg") else (None, None, None)
    vggW = O.seeded_vgg_weights()
    many = min(8, cores)
    t_cr, t_c1 = {}, {}
    try:
        for n in sorted({many, 1}, reverse=True):
            torch.set_num_threads(n)
            t_cr[n] = _cpu_steps(P, params, opt, hazy, gt, vggW, 1.0, CPU_STEPS)          # the headline's recipe
            t_c1[n] = _cpu_steps(P, params, opt, hazy, gt, None, 0.0, CPU_STEPS)          # config 1: Charbonnier only
        all_cores = None
        if cores > many:
            torch.set_num_threads(cores)
            all_cores = _cpu_steps(P, params, opt, hazy, gt, vggW, 1.0, 2)
    finally:
        torch.set_num_threads(cores)
    load1 = os.getloadavg() if hasattr(os, "getloadavg") else (None, None, None)
    rnd = lambda d: {str(n): round(bs / t, 4) for n, t in sorted(d.items())}
    return {"value": round(bs / t_cr[many], 4), "unit": "patches/s", "cores": many, "kind": "port",
            "sample": f"CPU oracle, BASELINE configs[1] recipe (E=32 ps=128 fp32, Charbonnier + VGG19 contrastive loss, AdamW) at bs={bs}, "
                      f"{many} threads, median of {CPU_STEPS} timed steps after 1 warm-up: {t_cr[many]:.3f} s/step "
                      f"(1 thread: {t_cr[1]:.3f} s/step); config-1 recipe (Charbonnier only), same protocol: "
                      + ", ".join(f"{t:.3f} s/step at {n} thread{'s' if n > 1 else ''}" for n, t in sorted(t_c1.items())),
            "patches_per_s_by_threads": rnd(t_cr), "value_1_thread": round(bs / t_cr[1], 4),
            "value_config1_charbonnier_only": round(bs / t_c1[many], 4), "patches_per_s_by_threads_config1": rnd(t_c1),
            "all_cores": None if all_cores is None else {"threads": cores, "patches_per_s": round(bs / all_cores, 4), "timed_steps": 2,
                                                          "note": "shared host: this figure measures the neighbours' load as much as the code"},
            "protocol": f"{CPU_STEPS} timed steps after 1 warm-up per (recipe, thread count); median",
            "host": {"os_cpu_count": os.cpu_count(), "torch_threads_default": cores,
                     "loadavg_1_5_15_before": [round(x, 2) if x is not None else None for x in load0],
                     "loadavg_1_5_15_after": [round(x, 2) if x is not None else None for x in load1]},
            "survey_container_reference": {"patches_per_s_8_threads": 1.98, "patches_per_s_1_thread": 0.40,
                                           "note": "the reference's own My_model_1.Uformer, config-1 recipe, BASELINE.md section 2"}}


def load_pmc_traffic(path, build_id):
    """HBM bytes per launch from the committed PMC passes (tools/pmc_summary.py: FETCH doubled per the gfx950 correction of
    MI355X_MICROARCH.md, WRITE exact; separate --pmc runs of this same command).  The file carries the dhz_build_id() of the
    library the passes ran with; a figure measured on other kernels than the ones loaded now is NOT reported:
    returns ({}, reason) then, (table, source) otherwise."""
    try:
        pmc = json.load(open(path))
    except Exception as e:
        return {}, f"no PMC table ({type(e).__name__}): traffic not reported"
    stamp = (pmc.get("_stamp") or {}).get("build_id")
    name = os.path.relpath(path, ROOT)
    if stamp is None or stamp != build_id:
        return {}, f"{name} was measured on library build {stamp}, the loaded one is {build_id}: traffic not reported"
    return {k: v for k, v in pmc.items() if not k.startswith("_")}, f"{name} (rocprofv3 --pmc passes of library build {build_id})"


def live_pmc_traffic(extra_args, timeout_s=240):
    """HBM bytes per launch measured IN THIS RUN: two child processes of this same script (2 + 2 steps, nothing but the headline) under
    `rocprofv3 --pmc FETCH_SIZE` and `rocprofv3 --pmc WRITE_SIZE` - separate passes, no trace domain beside the counters, the interpreter
    itself behind `--` - summarised as tools/pmc_summary.py does (FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md, KiB
    units, WRITE_SIZE exact).  Returns (table, source) or ({}, reason)."""
    import collections
    import csv
    import glob
    import re
    import shutil
    import signal
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return {}, "rocprofv3 not on this box"
    tmp = tempfile.mkdtemp(prefix="dhz_pmc_", dir="/tmp")
    agg = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__), "--steps", "2",
                   "--warmup", "2", "--no-cpu-baseline", "--no-kernel-timing", "--no-fp32-pipe", "--no-config4", "--no-config5", "--no-live-traffic"] + extra_args
            # own session: on a timeout the WHOLE group goes (rocprofv3 and the interpreter under it), nothing keeps running on the GPU
            proc = subprocess.Popen(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.PIPE,
                                    text=True, start_new_session=True)
            try:
                _, err = proc.communicate(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(proc.pid, signal.SIGKILL)
                except ProcessLookupError:
                    pass
                proc.wait()
                return {}, f"rocprofv3 --pmc {counter} pass exceeded {timeout_s} s (process group killed)"
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if proc.returncode != 0 or not files:
                return {}, f"rocprofv3 --pmc {counter} pass failed (rc {proc.returncode}): {(err or '')[-200:].strip()}"
            per = collections.defaultdict(list)
            for row in csv.DictReader(open(files[0])):
                if row["Counter_Name"] == counter:
                    per[re.sub(r"\(anonymous namespace\)::|^void ", "", row["Kernel_Name"]).split("(")[0]].append(float(row["Counter_Value"]))
            agg[counter] = per
    except Exception as e:                                   # a profiler hiccup must never cost the bench line
        return {}, f"live PMC passes failed ({type(e).__name__}: {e})"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    table = {}
    for k, vals in agg["FETCH_SIZE"].items():
        rd = 2.0 * sum(vals) / len(vals) * 1024.0
        wv = agg["WRITE_SIZE"].get(k) or [0.0]
        wr = sum(wv) / len(wv) * 1024.0
        table[k] = {"launches": len(vals), "read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr}
    return table, "measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, two child passes of this command (2 + 2 steps)"


def run_config4(dev, steps, warmup):
    """BASELINE configs[3] on the driver's clock: E=64, ps=256, per-GPU bs=8, bf16 activations + bf16 weight copies with fp32
    accumulation / master weights, Charbonnier + CR + AdamW.  Same step function, same timing discipline as the headline; the
    bf16 token-Linear GEMMs are timed with HIP events in a separate pass for the roofline object."""
    import warnings
    import My_model_1 as M1
    import My_CR
    from losses import CharbonnierLoss
    from dehaze_hip import ops
    from dehaze_hip.train import FlatAdamW, synthetic_batch, train_step
    E, ps, bs = 64, 256, 8
    torch.manual_seed(1234)
    model = M1.Uformer(img_size=ps, embed_dim=E, win_size=8, token_projection='linear', token_mlp='leff').to(dev)
    model.train()
    model.act_dtype = torch.bfloat16
    opt = FlatAdamW(model, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.02)
    opt.zero_grad()
    char = CharbonnierLoss()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        cr = My_CR.ContrastLoss(ablation=False).to(dev)
    target, input_ = synthetic_batch(bs, ps, seed=1234, device=dev)
    torch.manual_seed(4321)

    def step():
        return train_step(model, char, cr, opt, None, input_, target, 1.0, 1.0)

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss, _, _ = step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ops.KERNEL_TIMING = {"dhz_linear_bf16": [], "dhz_vgg_conv3x3_bf16": []}
    for _ in range(min(steps, 3)):
        step()
    torch.cuda.synchronize()
    timing, ops.KERNEL_TIMING = ops.KERNEL_TIMING, None
    out = {"workload": f"Uformer_ProbSparse train step E={E} ps={ps} per-GPU bs={bs} bf16 activations + bf16 weight copies, fp32 "
                       "accumulation / master weights, Charbonnier+CR(VGG19, seeded-random weights) + AdamW (BASELINE configs[3])",
           "value": round(bs * steps / el, 3), "unit": "patches/s", "ms_per_step": round(1e3 * el / steps, 3), "steps": steps,
           "warmup": warmup, "dtype": "bf16", "loss_last_step": round(float(loss), 6)}
    for key, name, kern in (("dhz_linear_bf16", "roofline", "gemm_bf16_pipe_kernel<2,NTS> / gemm_bf16_kernel<WM,WN,BTR> (dhz_linear_fwd_bf16 / dhz_linear_dgrad_bf16; backward-data mostly as the forward kernel on the bf16 copy of W^T)"),
                            ("dhz_vgg_conv3x3_bf16", "roofline_conv_bf16", "conv3_bf16_kernel<WM,WN> (dhz_vgg_conv3x3_bf16)")):
        ev = timing.get(key) or []
        if ev:
            ms = sum(e[0].elapsed_time(e[1]) for e in ev)
            flops = sum(e[2] for e in ev)
            tf = flops / (ms * 1e-3) / 1e12
            out[name] = {"kernel": kern, "bound": "mfma", "achieved": round(tf, 1), "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s",
                         "frac": round(tf / MFMA_BF16_PEAK_TF, 4), "traffic": None, "launches": len(ev),
                         "avg_launch_us": round(1e3 * ms / len(ev), 2), "alg_flops_per_launch": int(flops / len(ev))}
    return out


def run_config5(dev, forwards=10, warm=2):
    """BASELINE configs[4] on the driver's clock: whole-image restoration as the reference's test_long_GPU.py does it (:72-93) - a
    1200 x 1600 image wrap-padded to L = 1664, ONE eval forward on [1, 3, 1664, 1664] (43,264 windows per full-resolution block), crop,
    clamp.  s/image over `forwards` timed forwards after `warm`; the window-attention kernel of the north star is timed with HIP events
    in a separate pass (its inference regime: no training saves)."""
    import My_model_1 as M1
    import test_long_GPU as TL
    from dehaze_hip import ops
    from dehaze_hip.train import synthetic_batch
    torch.manual_seed(1234)
    model = M1.Uformer(img_size=128, embed_dim=32, win_size=8, token_projection='linear', token_mlp='leff').to(dev).eval()
    _, hazy = synthetic_batch(1, (1200, 1600), seed=900, device=dev)
    L = TL.padded_size(1200, 1600, 128)
    with torch.no_grad():
        for _ in range(warm):
            out = TL.restore_image(model, hazy, 128)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(forwards):
            out = TL.restore_image(model, hazy, 128)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        peak = torch.cuda.max_memory_allocated(dev)
        ops.KERNEL_TIMING = {"dhz_fused_window_attn_fwd": []}
        for _ in range(3):
            TL.restore_image(model, hazy, 128)
        torch.cuda.synchronize()
        ev, ops.KERNEL_TIMING = ops.KERNEL_TIMING["dhz_fused_window_attn_fwd"], None
    res = {"workload": f"Uformer_ProbSparse whole-image eval forward E=32, 1200x1600 image wrap-padded to [1,3,{L},{L}], fp32 "
                       "(BASELINE configs[4]: test_long_GPU.py pad + one forward + crop + clamp), random-init weights",
           "value": round(el / forwards, 5), "unit": "s/image", "higher_is_better": False, "images_per_s": round(forwards / el, 3),
           "forwards": forwards, "warmup": warm, "padded_side": L, "windows_per_full_res_block": (L // 8) ** 2,
           "peak_hbm_gb": round(peak / 2 ** 30, 2), "output_finite": bool(torch.isfinite(out).all().item()),
           "psnr_ssim": "not a bench quantity (random-init weights, synthetic image); parity of this path: tests/test_gpu_data_eval.py"}
    if ev:
        ms = sum(a.elapsed_time(b) for a, b, _, _ in ev)
        flops = sum(n * 2 * 64 * (4 * c * c + 75 * c) for _, _, n, c in ev)
        tf = flops / (ms * 1e-3) / 1e12
        res["roofline"] = {"kernel": "fused_window_attn_fwd_kernel<C,SAVE=0> (dhz_fused_window_attn_fwd, inference: no training saves), C in {32,64,128}",
                           "bound": "mfma", "achieved": round(tf, 2), "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                           "frac": round(tf / MFMA_F32_PEAK_TF, 4), "traffic": None, "launches": len(ev),
                           "avg_launch_us": round(1e3 * ms / len(ev), 2), "alg_flops_per_launch": flops // len(ev)}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch (weak scaling)")
    ap.add_argument("--strong", action="store_true", help="strong scaling (SURVEY 8d): GLOBAL batch = --batch, split over the ranks")
    ap.add_argument("--embed_dim", type=int, default=32)
    ap.add_argument("--ps", type=int, default=128)
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="storage type of the token tensors: f32 (BASELINE configs 2/3) or bf16 (config 4: --embed_dim 64 --ps 256 --batch 8)")
    ap.add_argument("--no-cr", action="store_true", help="Charbonnier only (NOT the headline config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--bucket-mb", type=float, default=25.0, help="gradient all-reduce bucket size (N > 1)")
    ap.add_argument("--no-config4", action="store_true",
                    help="skip the BASELINE configs[3] measurement (E=64 ps=256 bs=8 bf16) that the default configs[1] run appends as the "
                         "`config4` object")
    ap.add_argument("--no-config5", action="store_true",
                    help="skip the BASELINE configs[4] measurement (whole-image eval forward on [1,3,1664,1664]) appended as the `config5` object")
    ap.add_argument("--no-overlap", action="store_true",
                    help="N > 1 diagnostics: launch every gradient bucket's all-reduce AFTER backward instead of from the hooks (separates "
                         "'RCCL starved by the persistent compute grids' from 'RCCL slow')")
    ap.add_argument("--reserve-cus", type=int, default=None,
                    help="CUs the persistent compute grids leave free (dhz_set_reserved_cus).  N > 1: for RCCL's kernels beside the backward "
                         "pass (default: DHZ_COMM_RESERVE_CUS or 0).  N = 1: applied as given - measures what a reservation costs the step")
    ap.add_argument("--side-stream", type=int, choices=[0, 1], default=int(os.environ.get("DHZ_SIDE_STREAM", "0")),
                    help="1: the no-gradient VGG19 passes of the contrastive loss (target, hazy input) run on a second HIP stream beside the "
                         "model's forward (train_step(side=SideStream)); same work, same results")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not measure `roofline.traffic` with two rocprofv3 --pmc child passes; use the stamped profiles/pmc_traffic.json")
    ap.add_argument("--no-fp32-pipe", action="store_true",
                    help="skip the second measurement of the same step on the fp32 matrix pipe (the `fp32_pipe` object)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback for the product path)")
    # DHZ_DIST_BACKEND=gloo + DHZ_SHARE_GPU=1: test-only way to run N ranks on a box with ONE device (tests/test_gpu_ddp.py
    # does the same for the step itself); the driver's multi-GPU runs use RCCL ("nccl"), one rank per GPU
    backend = os.environ.get("DHZ_DIST_BACKEND", "nccl")
    if os.environ.get("DHZ_SHARE_GPU"):
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if args.strong:
        assert args.batch % world == 0, f"--strong: global batch {args.batch} not divisible by {world} ranks"
        args.batch //= world

    import My_model_1 as M1
    import My_CR
    from losses import CharbonnierLoss
    from dehaze_hip import ops, _lib
    from dehaze_hip.train import FlatAdamW, GradReducer, synthetic_batch, train_step

    torch.manual_seed(1234)                       # identical replicas on every rank
    model = M1.Uformer(img_size=args.ps, embed_dim=args.embed_dim, win_size=8, token_projection='linear',
                       token_mlp='leff').to(dev)
    model.train()
    if args.dtype == "bf16":
        if args.embed_dim % 64:
            raise SystemExit(f"bench.py --dtype bf16: the bf16 kernels tile channels in 64s; --embed_dim {args.embed_dim} is not a "
                             "multiple of 64 (BASELINE config 4 is --embed_dim 64 --ps 256 --batch 8)")
        model.act_dtype = torch.bfloat16          # bf16 activations / weight copies, fp32 accumulation and master weights
    opt = FlatAdamW(model, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.02)
    opt.zero_grad()
    reducer = GradReducer(opt, bucket_mb=args.bucket_mb, overlap=not args.no_overlap, reserve_cus=args.reserve_cus) if world > 1 else None
    if world == 1 and args.reserve_cus:
        _lib.call("dhz_set_reserved_cus", int(args.reserve_cus))
    char = CharbonnierLoss()
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        cr = None if args.no_cr else My_CR.ContrastLoss(ablation=False).to(dev)
    target, input_ = synthetic_batch(args.batch, args.ps, seed=1234 + rank, device=dev)
    torch.manual_seed(4321 + rank)                # per-rank sampling / DropPath streams

    side = None
    if args.side_stream and not args.no_cr:
        from dehaze_hip.train import SideStream
        side = SideStream(dev)

    def step():
        return train_step(model, char, cr, opt, reducer, input_, target, 1.0, 0.0 if args.no_cr else 1.0, side=side)

    # The headline runs the product's DEFAULT arithmetic (dehaze_hip.ops.SPLIT_BF16 = 6 unless the environment overrides it): fp32
    # storage and accumulation, the products of the GEMM-shaped kernels as six bf16 MFMA passes over operands cut into three bf16
    # pieces (dropped terms <= 2^-24 relative).  The same step on the fp32 matrix pipe is timed after it as `fp32_pipe`.
    headline_terms = ops.SPLIT_BF16 if args.dtype == "f32" else 0

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, _, _ = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = tmax.item()

    # per-kernel HIP-event timing for the roofline objects: a SEPARATE pass of the same step after the timed region (the
    # product path records no events; ~180 event records per step would be work the timed steps do not do)
    timing = None
    if not args.no_kernel_timing and rank == 0:
        ops.KERNEL_TIMING = {"dhz_ps_attn_fwd": [], "dhz_fused_window_attn_fwd": [], "dhz_winograd_conv3x3": [],
                             "dhz_linear_bf16": [], "dhz_vgg_conv3x3_bf16": [], "dhz_linear_split6": [], "dhz_linear_wgrad_split": []}
    if not args.no_kernel_timing:
        for _ in range(min(args.steps, 5)):
            step()                                # every rank runs it (the step holds a collective)
        torch.cuda.synchronize()
    timing = ops.KERNEL_TIMING
    ops.KERNEL_TIMING = None
    exchange_diag = None
    if reducer is not None:
        # N > 1 self-diagnosis (the 8-GPU run is the driver's, not the builder's): (a) how long the compute stream sits in
        # reducer.wait() per step - the exchange time NOT hidden behind the backward pass; (b) the same step with every bucket
        # launched after backward (no overlap at all).  (a) near 0 and (b) - headline near the ring time: the overlap works;
        # (a) near the ring time: RCCL's kernels did not get CUs beside the persistent grids (try --reserve-cus 8).
        n_diag = min(args.steps, 5)
        reducer.timing = []
        for _ in range(n_diag):
            step()
        torch.cuda.synchronize()
        exposed = [a.elapsed_time(b) for a, b in reducer.timing]
        reducer.timing = None
        was = reducer.overlap
        reducer.overlap = False
        step()
        dist.barrier()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for _ in range(n_diag):
            step()
        torch.cuda.synchronize()
        dist.barrier()
        el2 = time.perf_counter() - t2
        reducer.overlap = was
        tm2 = torch.tensor([el2, sum(exposed) / max(len(exposed), 1)], device=dev, dtype=torch.float64)
        dist.all_reduce(tm2, op=dist.ReduceOp.MAX)
        exchange_diag = {"exposed_wait_ms_per_step": round(tm2[1].item(), 3), "no_overlap_ms_per_step": round(1e3 * tm2[0].item() / n_diag, 3),
                         "steps": n_diag, "how": "HIP events around the joins of GradReducer.wait() on the compute stream (max over ranks); "
                                                 "then the same step with every bucket's all-reduce launched after backward"}
    fp32_pipe = None
    if headline_terms and not args.no_fp32_pipe:
        # the same model / optimizer state / batch with every product on the fp32 matrix pipe (v_mfma_f32_16x16x4_f32)
        ops.SPLIT_BF16 = 0
        for _ in range(min(args.warmup, 5)):
            step()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            loss_s, _, _ = step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        el = time.perf_counter() - t1
        ops.SPLIT_BF16 = headline_terms
        if world > 1:
            tm = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            el = tm.item()
        fp32_pipe = {"value": round(args.batch * world * args.steps / el, 3), "unit": "patches/s",
                     "ms_per_step": round(1e3 * el / args.steps, 3), "steps": args.steps,
                     "arithmetic": "every product on the fp32 matrix pipe (v_mfma_f32_16x16x4_f32); same process, model, optimizer "
                                   "state and batch as the headline, measured after it",
                     "loss_last_step": round(float(loss_s), 6)}
    extras = {}
    if rank == 0:
        # every TIMED measurement first (the other single-GPU BASELINE configurations, the CPU baseline); the counter passes of
        # `roofline.traffic` - two profiler child processes on the same GPU - run after the last of them
        plan = reducer.plan() if reducer is not None else None
        reserve = reducer.reserve_cus if reducer is not None else None
        if world == 1 and not args.no_config4 and (args.dtype, args.embed_dim, args.ps, args.batch) == ("f32", 32, 128, 32) \
                and not args.no_cr:
            # on the same clock: free the config-2 state first
            model = opt = cr = reducer = None
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            extras["config4"] = run_config4(dev, args.steps, min(args.warmup, 5))
            if not args.no_config5:
                gc.collect()
                torch.cuda.empty_cache()
                torch.cuda.reset_peak_memory_stats(dev)
                extras["config5"] = run_config5(dev)
        if world == 1 and not args.no_cpu_baseline:
            extras["cpu_baseline"] = cpu_baseline()
        total = args.batch * world * args.steps
        out = {
            "metric": f"train patches/sec ({args.ps}x{args.ps}, embed_dim={args.embed_dim})", "value": round(total / elapsed, 3),
            "unit": "patches/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "strong" if args.strong else "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"Uformer_ProbSparse train step E={args.embed_dim} ps={args.ps} per-GPU bs={args.batch} "
                                   f"{'fp32' if args.dtype == 'f32' else 'bf16 activations + bf16 weight copies, fp32 accumulation / master weights,'} "
                                   f"{'Charbonnier' if args.no_cr else 'Charbonnier+CR(VGG19, seeded-random weights)'} + AdamW "
                                   + ("(BASELINE configs[1])" if (args.dtype, args.embed_dim, args.ps, args.batch) == ("f32", 32, 128, 32)
                                      else "(BASELINE configs[3])" if (args.dtype, args.embed_dim, args.ps) == ("bf16", 64, 256) else "(not a BASELINE config)"),
                       "global_batch": args.batch * world, "parallelism": f"dp{world}",
                       **({"reserve_cus": int(args.reserve_cus), "grid_cus": _lib.load().dhz_grid_cus()} if world == 1 and args.reserve_cus else {}),
                       "loss_last_step": round(float(loss), 6)},
        }
        if args.dtype == "f32":
            out["config"]["arithmetic"] = (
                "fp32 storage/accumulate; products 6xbf16 MFMA, dropped <= 2^-24" if headline_terms == 6 else
                "fp32 storage/accumulate; products on the fp32 matrix pipe" if headline_terms == 0 else
                "EXPERIMENT (not a product setting): fp32 storage/accumulate; products 3xbf16 MFMA (~16 mantissa bits per product)")
            out["config"]["split_terms"] = headline_terms
        if plan is not None:
            # the exchange of one step as performed (bucket byte ranges in launch order, single-ring xGMI time): makes a
            # scaling run diagnosable from its JSON line alone
            out["exchange"] = {"backend": backend, "bucket_mb": args.bucket_mb, "overlap_with_backward": not args.no_overlap, "reserve_cus": reserve,
                               "grid_cus": _lib.load().dhz_grid_cus(), "payload_bytes": plan["payload_bytes"],
                               "n_buckets": len(plan["buckets"]), "bucket_bytes": [b["bytes"] for b in plan["buckets"]],
                               "ring_time_ms_single_link": round(plan["ring_time_ms"], 3),
                               "measured": "ring time is the plan's figure, not a measurement",
                               **(exchange_diag or {})}
        build_id = _lib.load().dhz_build_id().decode()
        pmc, traffic_source = load_pmc_traffic(os.path.join(ROOT, "profiles", "pmc_traffic.json"), build_id)
        if world == 1 and timing and not args.no_live_traffic and args.dtype == "f32":
            extra = ["--batch", str(args.batch), "--embed_dim", str(args.embed_dim), "--ps", str(args.ps)] + (["--no-cr"] if args.no_cr else [])
            live, why = live_pmc_traffic(extra)
            if live:
                pmc, traffic_source = live, why
            else:
                traffic_source = f"{traffic_source} (live passes: {why})"
        out["library_build_id"] = build_id
        if timing and timing.get("dhz_linear_bf16"):
            # config 4: the token-Linear GEMMs on v_mfma_f32_16x16x32_bf16 (forward + backward-data launches), against the dense
            # bf16 matrix peak; most of their shapes are HBM-bound at bf16 MFMA rates, so the HBM view is given beside it
            ev = timing["dhz_linear_bf16"]
            ms = sum(e[0].elapsed_time(e[1]) for e in ev)
            flops = sum(e[2] for e in ev)
            tf = flops / (ms * 1e-3) / 1e12
            out["roofline"] = {"kernel": "gemm_bf16_pipe_kernel<2,NTS> / gemm_bf16_kernel<WM,WN,BTR> (dhz_linear_fwd_bf16 / dhz_linear_dgrad_bf16; backward-data mostly as the forward kernel on the bf16 copy of W^T)", "bound": "mfma",
                               "achieved": round(tf, 1), "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s",
                               "frac": round(tf / MFMA_BF16_PEAK_TF, 4), "traffic": None, "launches": len(ev),
                               "avg_launch_us": round(1e3 * ms / len(ev), 2), "alg_flops_per_launch": int(flops / len(ev))}
        if timing and timing.get("dhz_vgg_conv3x3_bf16"):
            # config 4: the VGG19 convolutions of the contrastive loss as implicit GEMMs on the bf16 matrix pipe (direct-convolution
            # FLOPs, nothing skipped); LDS bandwidth caps this tiling near 0.5 of the dense peak (docs/history/DESIGN_rounds_1-5.md section 7)
            ev = timing["dhz_vgg_conv3x3_bf16"]
            ms = sum(e[0].elapsed_time(e[1]) for e in ev)
            flops = sum(e[2] for e in ev)
            tf = flops / (ms * 1e-3) / 1e12
            out["roofline_conv_bf16"] = {"kernel": "conv3_bf16_kernel<WM,WN> (dhz_vgg_conv3x3_bf16)", "bound": "mfma",
                                         "achieved": round(tf, 1), "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s",
                                         "frac": round(tf / MFMA_BF16_PEAK_TF, 4), "traffic": None, "launches": len(ev),
                                         "avg_launch_us": round(1e3 * ms / len(ev), 2), "alg_flops_per_launch": int(flops / len(ev))}
        if timing and timing.get("dhz_fused_window_attn_fwd"):
            # the window-attention kernel of the north star: LN + QKV + ProbSparse core + out-proj + residual, fused.
            # algorithmic FLOPs per window = 2*64*(4C^2 + 75C) (four CxC projections + 3x(25x64x32) core)
            ev = timing["dhz_fused_window_attn_fwd"]
            ms = sum(a.elapsed_time(b) for a, b, _, _ in ev)
            flops = sum(n * 2 * 64 * (4 * c * c + 75 * c) for _, _, n, c in ev)
            tf = flops / (ms * 1e-3) / 1e12
            # launch-weighted mean over the template instances (C = 32 / 64 / 128) of the committed PMC passes
            inst = [v for k, v in pmc.items() if k.startswith("fused_window_attn_fwd_kernel")]
            traffic = (sum(v["hbm_bytes_per_launch"] * v["launches"] for v in inst) / sum(v["launches"] for v in inst)) if inst else None
            # the C = 64 launches run their four weight products (Q, K, V, out-projection) as six bf16 MFMA passes each (fused.ATTN_FUSED_P6):
            # their ISSUED bf16 FLOPs against the dense bf16 peak, over the WHOLE launch time (the phases are not timed apart)
            from dehaze_hip import fused as _fused
            p6 = [(a, b, n, c) for a, b, n, c in ev if _fused.ATTN_FUSED_P6 and c in _fused.ATTN_FUSED_P6_C]
            p6_ms = sum(a.elapsed_time(b) for a, b, _, _ in p6)
            p6_issued = sum(6 * n * 2 * 64 * 4 * c * c for _, _, n, c in p6)
            out["roofline"] = {"kernel": "fused_window_attn_fwd_kernel<C,SAVE,NW,P6> (dhz_fused_window_attn_fwd / _fwd6), C in {32,64,128}; "
                                         "C = 64: weight products six-term on the bf16 pipe (P6), attention core on the fp32 pipe",
                               "bound": "mfma", "achieved": round(tf, 2), "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                               "frac": round(tf / MFMA_F32_PEAK_TF, 4),
                               "frac_counts": "ALGORITHMIC FLOPs 2*64*(4C^2 + 75C) per window against the fp32 matrix peak (the contract's number)",
                               "traffic": round(traffic) if traffic else None, "traffic_source": traffic_source,
                               "launches": len(ev), "avg_launch_us": round(1e3 * ms / len(ev), 2),
                               "alg_flops_per_launch": flops // len(ev)}
            if p6:
                out["roofline"]["p6_launches"] = {
                    "launches": len(p6), "avg_launch_us": round(1e3 * p6_ms / len(p6), 2),
                    "alg_frac_of_fp32_peak": round(sum(n * 2 * 64 * (4 * c * c + 75 * c) for _, _, n, c in p6) / (p6_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TF, 4),
                    "projection_issued_bf16_tflops_over_launch_time": round(p6_issued / (p6_ms * 1e-3) / 1e12, 1),
                    "projection_issued_bf16_frac_of_2500": round(p6_issued / (p6_ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TF, 4)}
        if timing and timing.get("dhz_ps_attn_fwd"):
            # the stand-alone ProbSparse core (stages the fused kernel does not cover): HBM-bound, 32 KiB / window-head
            ev = timing["dhz_ps_attn_fwd"]
            ms = sum(e[0].elapsed_time(e[1]) for e in ev)
            bytes_alg = sum(n for _, _, n in ev)          # Q, K, V in + context out per window-head: 4 * 64 * d * element size
            gbs = bytes_alg / (ms * 1e-3) / 1e9
            traffic = next((v["hbm_bytes_per_launch"] for k, v in pmc.items() if k.startswith("ps_attn_fwd_kernel<32")), None) \
                if args.dtype == "f32" and args.embed_dim == 32 else None
            entry = {"kernel": "ps_attn_fwd_kernel<d,T> (dhz_ps_attn_fwd)", "bound": "hbm", "achieved": round(gbs, 1),
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                     "traffic": round(traffic) if traffic else None, "traffic_source": traffic_source, "launches": len(ev),
                     "avg_launch_us": round(1e3 * ms / len(ev), 2), "alg_bytes_per_launch": bytes_alg // len(ev)}
            out["roofline" if "roofline" not in out else "roofline_core_unfused"] = entry
        if timing and timing.get("dhz_winograd_conv3x3"):
            # the kernel with the largest share of the step (VGG19 convolutions of the contrastive loss): Winograd F(2x2,3x3),
            # so the matrix pipe executes direct-conv FLOPs / 2.25; frac is the ISSUED MFMA rate against the fp32 peak
            ev = timing["dhz_winograd_conv3x3"]
            ms = sum(e[0].elapsed_time(e[1]) for e in ev)
            direct = sum(e[2] for e in ev)
            issued = sum(e[3] for e in ev)                           # direct / 2.25 on the F(2x2,3x3) launches, / 4 on the F(4x4,3x3) ones
            tf_direct = direct / (ms * 1e-3) / 1e12
            tf_issued = issued / (ms * 1e-3) / 1e12
            inst = [v for k, v in pmc.items() if k.startswith("winograd_conv3x3_kernel") or k.startswith("winograd43_conv3x3_kernel")]
            traffic = (sum(v["hbm_bytes_per_launch"] * v["launches"] for v in inst) / sum(v["launches"] for v in inst)) if inst else None
            out["roofline_dominant"] = {"kernel": "winograd43_conv3x3_kernel<FWD> (dhz_winograd43_conv3x3 / _pool: F(4x4,3x3); the no-gradient passes and the backward-data products on grids of whole rounds) + "
                                                  "winograd_conv3x3_kernel<FWD,Q8> (dhz_winograd_conv3x3: F(2x2,3x3); the differentiated forward pass, half-round grids, 8 x 8 maps): "
                                                  "the VGG19 convolutions, largest share of the step",
                                        "bound": "mfma", "achieved": round(tf_issued, 2), "peak": MFMA_F32_PEAK_TF,
                                        "unit": "TFLOP/s", "frac": round(tf_issued / MFMA_F32_PEAK_TF, 4),
                                        "counts": "ISSUED transform-domain matrix FLOPs (direct-convolution FLOPs / 4 on the F(4x4) launches, "
                                                  "/ 2.25 on the F(2x2) ones)",
                                        "direct_conv_equivalent_tflops": round(tf_direct, 1),
                                        "traffic": round(traffic) if traffic else None, "traffic_source": traffic_source,
                                        "launches": len(ev), "avg_launch_us": round(1e3 * ms / len(ev), 2),
                                        "alg_flops_per_launch": int(issued / len(ev))}
        for key, name, kern in (("dhz_linear_split6", "roofline_split6_gemm",
                                 "split6_wide_kernel / split6_gemm_kernel / gemm_split_kernel (dhz_linear_fwd_split6, dhz_linear_dgrad_split6, "
                                 "dhz_linear_fwd_split, dhz_linear_dgrad_split): six-term forward / backward-data token-Linear GEMMs"),
                                ("dhz_linear_wgrad_split", "roofline_split6_wgrad",
                                 "wgrad_split6_kernel (dhz_linear_wgrad_split): six-term weight gradients")):
            ev = (timing or {}).get(key) or []
            if ev:
                # condition (v) of the round-3 ruling: a split kernel is priced on the bf16 FLOPs it ISSUES (six MFMA products per
                # multiply-add) against the dense bf16 matrix peak - never against the fp32 pipe's 157.3
                ms = sum(e[0].elapsed_time(e[1]) for e in ev)
                flops = sum(e[2] for e in ev)
                tf = flops / (ms * 1e-3) / 1e12
                out[name] = {"kernel": kern, "bound": "mfma", "achieved": round(tf, 1), "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s",
                             "frac": round(tf / MFMA_BF16_PEAK_TF, 4), "flops_counted": "issued bf16 MFMA FLOPs = 6 x algorithmic",
                             "algorithmic_tflops": round(tf / 6.0, 1), "traffic": None, "launches": len(ev),
                             "avg_launch_us": round(1e3 * ms / len(ev), 2), "issued_flops_per_launch": int(flops / len(ev))}
        if fp32_pipe is not None:
            out["fp32_pipe"] = fp32_pipe
        out.update(extras)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
