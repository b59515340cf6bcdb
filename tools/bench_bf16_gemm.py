"""bf16 token-Linear forward GEMM (dhz_linear_fwd_bf16) on the config-4 step's shapes: us per call, TFLOP/s, and (with 'check') the error
against float64 on the same bf16 operands.  DHZ_BF16_PIPE=0 in a second process gives the round-2 kernel's times (csrc/linear_bf16.hip);
the default routes the shapes with a 256 x 128 tile per CU to csrc/gemm_bf16_pipe.hip.  'dgrad' times the backward-data entry point
(transposed reads of W) against the forward kernel on a bf16 copy of W^T."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "research-and-implementation-of-image-dehazing-algorithm-based-on-vision-transformer_amd"))
import torch
from dehaze_hip import _lib
dev = torch.device("cuda:0"); s = torch.cuda.current_stream().cuda_stream
check = "check" in sys.argv; dgrad = "dgrad" in sys.argv
shapes = []
for T, C in [(524288, 64), (131072, 128), (32768, 256), (8192, 512), (2048, 1024), (8192, 1024), (32768, 512), (131072, 256), (524288, 128)]:
    shapes += [(T, C, 3 * C), (T, C, C), (T, C, 4 * C), (T, 4 * C, C)]          # (T, K, N)
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
tot = 0.0; totd = [0.0, 0.0]
for T, K, N in shapes:
    x = torch.randn(T, K, device=dev).to(torch.bfloat16); w = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    b = torch.randn(N, device=dev); y = torch.empty(T, N, device=dev, dtype=torch.bfloat16)
    f = lambda: _lib.call("dhz_linear_fwd_bf16", x.data_ptr(), K, w.data_ptr(), b.data_ptr(), y.data_ptr(), N, T, N, K, s)
    err = ""
    if check:
        f(); torch.cuda.synchronize()
        rows = torch.randint(0, T, (2048,), device=dev)
        ref = x[rows].double() @ w.double().t() + b.double()
        e = (y[rows].double() - ref).abs().max().item(); sc = ref.abs().max().item()
        err = f"  max err {e:.3e} (|y| max {sc:.2f})"
        # ragged tail / last rows
        ref2 = x[-300:].double() @ w.double().t() + b.double()
        err += f" tail {(y[-300:].double() - ref2).abs().max().item():.3e}"
    t = timeit(f); tot += t
    line = f"T={T:7d} K={K:5d} N={N:5d} | fwd {t:7.1f} us {2 * T * N * K / t / 1e6:7.0f} TF"
    if dgrad:
        dy = torch.randn(T, N, device=dev).to(torch.bfloat16); dx = torch.empty(T, K, device=dev, dtype=torch.bfloat16)
        wt = w.t().contiguous()
        fd = lambda: _lib.call("dhz_linear_dgrad_bf16", dy.data_ptr(), N, w.data_ptr(), dx.data_ptr(), K, T, N, K, s)
        ft = lambda: _lib.call("dhz_linear_fwd_bf16", dy.data_ptr(), N, wt.data_ptr(), None, dx.data_ptr(), K, T, K, N, s)
        a, c = timeit(fd), timeit(ft); totd[0] += a; totd[1] += c
        line += f" | dgrad (transposed reads) {a:7.1f} us, forward kernel on W^T {c:7.1f} us"
        if check:
            fd(); d1 = dx.clone(); ft(); torch.cuda.synchronize()
            line += f"  max |diff| {(d1.float() - dx.float()).abs().max().item():.2e}"
    print(line + err, flush=True)
print(f"sum: fwd {tot:.0f} us" + (f", dgrad {totd[0]:.0f} us, forward kernel on W^T {totd[1]:.0f} us" if dgrad else ""))
