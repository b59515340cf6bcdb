"""A few representative token-Linear shapes, forward and backward-data, us per call (min over interleaved rounds):
   python tools/bench_gemm_few.py      (DHZ_LIB_PATH selects the library build)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "research-and-implementation-of-image-dehazing-algorithm-based-on-vision-transformer_amd"), ROOT]
import torch
from dehaze_hip import ops
dev = torch.device("cuda:0")
shapes = [(8192, 512, 2048), (8192, 2048, 512), (32768, 256, 1024), (131072, 128, 512), (131072, 512, 128), (524288, 64, 256),
          (524288, 256, 64), (524288, 64, 192), (2048, 512, 2048), (524288, 32, 128)]
fs, labels = [], []
for T, K, N in shapes:
    x = torch.randn(T, K, device=dev); W = torch.randn(N, K, device=dev) * 0.1; b = torch.randn(N, device=dev); dy = torch.randn(T, N, device=dev)
    fs.append(lambda x=x, W=W, b=b: ops.gemm_fwd(x, W, b)); labels.append(f"fwd   {T:7d} {K:5d} {N:5d}")
    fs.append(lambda dy=dy, W=W: ops.gemm_dgrad(dy, W)); labels.append(f"dgrad {T:7d} {K:5d} {N:5d}")
best = [1e9] * len(fs)
for f in fs: f()
torch.cuda.synchronize()
for _ in range(4):
    for i, f in enumerate(fs):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): f()
        e1.record(); torch.cuda.synchronize()
        best[i] = min(best[i], e0.elapsed_time(e1) / 5 * 1e3)
print(" ".join(f"{b:6.1f}" for b in best), f"| sum {sum(best):.0f}")
