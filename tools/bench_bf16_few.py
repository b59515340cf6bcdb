"""a few bf16 forward GEMM shapes, us per call (variant builds: tools/variants.sh SRC=gemm_bf16_pipe)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "research-and-implementation-of-image-dehazing-algorithm-based-on-vision-transformer_amd"))
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
out = []
for T, K, N in [(131072, 256, 1024), (131072, 1024, 256), (32768, 512, 2048), (524288, 128, 512), (8192, 4096, 1024)]:
    x = torch.randn(T, K, device=dev).to(torch.bfloat16); w = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    b = torch.randn(N, device=dev); y = torch.empty(T, N, device=dev, dtype=torch.bfloat16)
    f = lambda: _lib.call("dhz_linear_fwd_bf16", x.data_ptr(), K, w.data_ptr(), b.data_ptr(), y.data_ptr(), N, T, N, K, s)
    out.append(f"{T}x{K}->{N}: {timeit(f):6.1f}")
print(" | ".join(out))
