"""split-bf16 experiment: the fp32 GEMM kernels against csrc/linear_split.hip on the step's K >= 128 shapes (us per call)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..",
                                "research-and-implementation-of-image-dehazing-algorithm-based-on-vision-transformer_amd"))
import torch
from dehaze_hip import _lib
dev = torch.device("cuda:0")
s = torch.cuda.current_stream().cuda_stream
# (T, K, N) of forward GEMMs y = x W^T in the config-2 step with K >= 128, and of backward-data dx = dy W with N >= 128
shapes = [(524288, 128, 32), (131072, 256, 64), (131072, 128, 384), (131072, 128, 128), (131072, 128, 512), (131072, 512, 128),
          (32768, 256, 768), (32768, 256, 1024), (32768, 1024, 256), (8192, 512, 1536), (8192, 512, 2048), (8192, 2048, 512),
          (524288, 64, 256), (524288, 256, 64)]
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
tot = [0.0, 0.0, 0.0, 0.0]
print(f"{'T':>7} {'K':>5} {'N':>5} | fwd fp32  split3 split6 | dgrad fp32  split3 split6   (us)")
for T, K, N in shapes:
    x = torch.randn(T, K, device=dev); w = torch.randn(N, K, device=dev) / K ** 0.5; b = torch.randn(N, device=dev)
    y = torch.empty(T, N, device=dev); dx = torch.empty(T, K, device=dev)
    r = []
    ok_f = K >= 128 and K % 64 == 0 and N % 64 == 0
    ok_d = N >= 128 and N % 64 == 0 and K % 64 == 0
    r.append(timeit(lambda: _lib.call("dhz_linear_fwd", x.data_ptr(), K, w.data_ptr(), b.data_ptr(), y.data_ptr(), N, T, N, K, s)))
    for terms in (3, 6):
        r.append(timeit(lambda: _lib.call("dhz_linear_fwd_split", x.data_ptr(), K, w.data_ptr(), b.data_ptr(), y.data_ptr(), N, T, N, K, terms, s)) if ok_f else float('nan'))
    r.append(timeit(lambda: _lib.call("dhz_linear_dgrad", y.data_ptr(), N, w.data_ptr(), dx.data_ptr(), K, T, N, K, s)))
    for terms in (3, 6):
        r.append(timeit(lambda: _lib.call("dhz_linear_dgrad_split", y.data_ptr(), N, w.data_ptr(), dx.data_ptr(), K, T, N, K, terms, s)) if ok_d else float('nan'))
    print(f"{T:7d} {K:5d} {N:5d} | {r[0]:8.1f} {r[1]:6.1f} {r[2]:6.1f} | {r[3]:10.1f} {r[4]:6.1f} {r[5]:6.1f}")
