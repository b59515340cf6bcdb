"""Micro-benchmark: dhz_linear_wgrad vs the library TN GEMM on the model's real shapes (bs=32)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "research-and-implementation-of-image-dehazing-algorithm-based-on-vision-transformer_amd"))
import torch
from dehaze_hip import _lib

dev = torch.device("cuda:0")
nolib = "nolib" in sys.argv[1:]          # variant runs (tools/variants.sh): only this library's kernel
shapes = []
for T, C in [(524288, 32), (131072, 64), (32768, 128), (8192, 256), (2048, 512), (8192, 512), (32768, 256), (131072, 128), (524288, 64)]:
    shapes += [(T, 3 * C, C), (T, C, C), (T, 4 * C, C), (T, C, 4 * C)]
s = torch.cuda.current_stream().cuda_stream
for T, N, K in shapes:
    dy = torch.randn(T, N, device=dev); x = torch.randn(T, K, device=dev)
    dw = torch.zeros(N, K, device=dev); db = torch.zeros(N, device=dev)
    def mine():
        _lib.call("dhz_linear_wgrad", dy.data_ptr(), N, x.data_ptr(), K, T, N, K, dw.data_ptr(), db.data_ptr(), s)
    def lib():
        return dy.t() @ x, dy.sum(0)
    res = []
    for f in ((mine, mine) if nolib else (mine, lib)):
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 10 * 1e3)
    gb = T * (N + K) * 4 / 1e9; tf = 2 * T * N * K / 1e12
    tot = (globals().get("tot", 0.0)) + res[0]
    print(f"T={T:7d} N={N:5d} K={K:5d}  mine {res[0]:8.1f} us ({gb/res[0]*1e6:7.0f} GB/s, {tf/res[0]*1e6:6.1f} TF)   lib {res[1]:8.1f} us  x{res[1]/res[0]:.2f}")
print(f"sum mine {tot:.0f} us")
