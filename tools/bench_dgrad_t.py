"""six-term backward-data: the transposed-read kernel on w's planes against the forward kernel on the planes of w^T (us per call)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "research-and-implementation-of-image-dehazing-algorithm-based-on-vision-transformer_amd"))
import torch
from dehaze_hip import _lib
dev = torch.device("cuda:0"); s = torch.cuda.current_stream().cuda_stream
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
def planes(w):
    pl = torch.empty((3, w.numel()), dtype=torch.bfloat16, device=dev)
    _lib.call("dhz_split3_planes", w.data_ptr(), w.numel(), pl[0].data_ptr(), pl[1].data_ptr(), pl[2].data_ptr(), s)
    return pl
tot = [0.0, 0.0]
for T, C in [(131072, 64), (32768, 128), (8192, 256), (8192, 512), (32768, 256), (131072, 128), (524288, 64)]:
    for K, N in [(C, 3 * C), (C, C), (C, 4 * C), (4 * C, C)]:
        w = torch.randn(N, K, device=dev) / K ** 0.5; dy = torch.randn(T, N, device=dev); dx = torch.empty(T, K, device=dev)
        p = planes(w); pt = planes(w.t().contiguous())
        a = timeit(lambda: _lib.call("dhz_linear_dgrad_split6", dy.data_ptr(), N, p[0].data_ptr(), p[1].data_ptr(), p[2].data_ptr(), dx.data_ptr(), K, T, N, K, s))
        r1 = dx.clone()
        b = timeit(lambda: _lib.call("dhz_linear_fwd_split6", dy.data_ptr(), N, pt[0].data_ptr(), pt[1].data_ptr(), pt[2].data_ptr(), None, dx.data_ptr(), K, T, K, N, s))
        tot[0] += a; tot[1] += b
        print(f"T={T:7d} K={K:5d} N={N:5d} | transposed reads {a:7.1f} | forward kernel on w^T planes {b:7.1f} | max diff {(dx - r1).abs().max().item():.1e}", flush=True)
print("sum", tot)
