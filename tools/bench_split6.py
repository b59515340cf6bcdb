"""six-term split GEMMs: csrc/split6_gemm.hip (pre-split weight planes, round 4) against csrc/linear_split.hip (round 3, both
operands split in the kernel) and the fp32-pipe kernels on the token-Linear shapes of the config-2 step.  us per call + max
error against float64 (first call of each)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..",
                                "research-and-implementation-of-image-dehazing-algorithm-based-on-vision-transformer_amd"))
import torch
from dehaze_hip import _lib
dev = torch.device("cuda:0")
s = torch.cuda.current_stream().cuda_stream
stages = [(524288, 32), (131072, 64), (32768, 128), (8192, 256), (2048, 512), (8192, 512), (32768, 256), (131072, 128), (524288, 64)]
only = [int(a) for a in sys.argv[1:] if a.isdigit()]
check = "check" in sys.argv
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
def planes(w):
    n = w.numel()
    out = [torch.empty(n, dtype=torch.bfloat16, device=dev) for _ in range(3)]
    _lib.call("dhz_split3_planes", w.data_ptr(), n, out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), s)
    return out
tot = [0.0] * 6
print(f"{'T':>7} {'K':>5} {'N':>5} | fwd: fp32  old6  new6   err32    err6 | dgrad: fp32  old6  new6   err32    err6")
for T, C in stages:
    if only and C not in only: continue
    for K, N in [(C, 3 * C), (C, C), (C, 4 * C), (4 * C, C)]:
        x = torch.randn(T, K, device=dev); w = torch.randn(N, K, device=dev) / K ** 0.5; b = torch.randn(N, device=dev)
        dy = torch.randn(T, N, device=dev)
        y = torch.empty(T, N, device=dev); dx = torch.empty(T, K, device=dev)
        ph, pm, pl = planes(w)
        r = []
        r.append(timeit(lambda: _lib.call("dhz_linear_fwd", x.data_ptr(), K, w.data_ptr(), b.data_ptr(), y.data_ptr(), N, T, N, K, s)))
        e32 = e6 = float('nan')
        if check:
            ref = (x[:4096].double() @ w.double().t() + b.double()); e32 = (y[:4096].double() - ref).abs().max().item()
        ok = K % 64 == 0 and N % 64 == 0
        r.append(timeit(lambda: _lib.call("dhz_linear_fwd_split", x.data_ptr(), K, w.data_ptr(), b.data_ptr(), y.data_ptr(), N, T, N, K, 6, s)) if ok else float('nan'))
        r.append(timeit(lambda: _lib.call("dhz_linear_fwd_split6", x.data_ptr(), K, ph.data_ptr(), pm.data_ptr(), pl.data_ptr(), b.data_ptr(), y.data_ptr(), N, T, N, K, s)))
        if check: e6 = (y[:4096].double() - ref).abs().max().item()
        r += [e32, e6]
        r.append(timeit(lambda: _lib.call("dhz_linear_dgrad", dy.data_ptr(), N, w.data_ptr(), dx.data_ptr(), K, T, N, K, s)))
        if check:
            ref = dy[:4096].double() @ w.double(); e32 = (dx[:4096].double() - ref).abs().max().item()
        r.append(timeit(lambda: _lib.call("dhz_linear_dgrad_split", dy.data_ptr(), N, w.data_ptr(), dx.data_ptr(), K, T, N, K, 6, s)) if ok else float('nan'))
        okd = K % 64 == 0
        r.append(timeit(lambda: _lib.call("dhz_linear_dgrad_split6", dy.data_ptr(), N, ph.data_ptr(), pm.data_ptr(), pl.data_ptr(), dx.data_ptr(), K, T, N, K, s)) if okd else float('nan'))
        if check and okd: e6 = (dx[:4096].double() - ref).abs().max().item()
        r += [e32, e6]
        for i, j in enumerate((0, 1, 2, 5, 6, 7)):
            tot[i] += r[j] if r[j] == r[j] else r[0 if i < 3 else 5]
        print(f"{T:7d} {K:5d} {N:5d} | {r[0]:9.1f} {r[1]:5.1f} {r[2]:5.1f} {r[3]:.1e} {r[4]:.1e} | {r[5]:11.1f} {r[6]:5.1f} {r[7]:5.1f} {r[8]:.1e} {r[9]:.1e}", flush=True)
print("sum (nan -> fp32): fwd fp32 %.0f old6 %.0f new6 %.0f | dgrad fp32 %.0f old6 %.0f new6 %.0f us" % tuple(tot))
