"""The residual epilogue of the six-term token-Linear GEMM (K4 in the GEMM: ops.gemm_fwd_res) against the plain GEMM followed by the separate
reverse_residual pass, on the out-projection / linear2 shapes of the step: us per call."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "research-and-implementation-of-image-dehazing-algorithm-based-on-vision-transformer_amd"))
import torch
from dehaze_hip import ops, _lib
dev = torch.device("cuda:0")
def timeit(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (B, H, K, N, win) in [(32, 64, 128, 128, 1), (32, 64, 512, 128, 0), (32, 32, 256, 256, 1), (32, 32, 1024, 256, 0), (32, 16, 512, 512, 1), (32, 16, 2048, 512, 0), (32, 128, 256, 64, 0), (32, 16, 256, 256, 1)]:
    T = B * H * H
    x = torch.randn(T, K, device=dev); w = torch.randn(N, K, device=dev) / K ** 0.5; b = torch.randn(N, device=dev)
    res = torch.randn(T, N, device=dev); sc = torch.ones(B, device=dev)
    t_plain = timeit(lambda: ops.gemm_fwd(x, w, b))
    t_epi = timeit(lambda: ops.gemm_fwd_res(x, w, b, res, sc, B, H, H, 4 if win else 0, bool(win)))
    y = ops.gemm_fwd(x, w, b); out = torch.empty_like(res)
    t_rr = timeit(lambda: _lib.call("dhz_reverse_residual_fwd_dt", y.data_ptr(), res.data_ptr(), sc.data_ptr(), out.data_ptr(), B, H, H, N, 4 if win else 0, win, 0, torch.cuda.current_stream().cuda_stream))
    print(f"T {T:7d} K {K:5d} N {N:4d} win {win}: plain GEMM {t_plain:6.1f} us, with epilogue {t_epi:6.1f} us (+{t_epi - t_plain:5.1f}), separate pass {t_rr:6.1f} us")
