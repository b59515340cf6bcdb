"""csrc/split6_gemm.hip on a few representative shapes (us per call): forward | backward-data.  For phase-ablation builds
(tools/variants.sh SRC=split6_gemm VARIANTS="-DDHZ_S6_ABL=0 -DDHZ_S6_ABL=1 ...")."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..",
                                "research-and-implementation-of-image-dehazing-algorithm-based-on-vision-transformer_amd"))
import torch
from dehaze_hip import _lib
dev = torch.device("cuda:0")
s = torch.cuda.current_stream().cuda_stream
shapes = [(32768, 256, 1024), (131072, 128, 384), (8192, 2048, 512), (131072, 512, 128), (524288, 64, 256), (524288, 256, 64), (131072, 64, 64)]
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
out = []
for T, K, N in shapes:
    x = torch.randn(T, K, device=dev); w = torch.randn(N, K, device=dev) / K ** 0.5; b = torch.randn(N, device=dev)
    dy = torch.randn(T, N, device=dev); y = torch.empty(T, N, device=dev); dx = torch.empty(T, K, device=dev)
    pl = [torch.empty(N * K, dtype=torch.bfloat16, device=dev) for _ in range(3)]
    _lib.call("dhz_split3_planes", w.data_ptr(), N * K, pl[0].data_ptr(), pl[1].data_ptr(), pl[2].data_ptr(), s)
    f = timeit(lambda: _lib.call("dhz_linear_fwd_split6", x.data_ptr(), K, pl[0].data_ptr(), pl[1].data_ptr(), pl[2].data_ptr(), b.data_ptr(), y.data_ptr(), N, T, N, K, s))
    d = timeit(lambda: _lib.call("dhz_linear_dgrad_split6", dy.data_ptr(), N, pl[0].data_ptr(), pl[1].data_ptr(), pl[2].data_ptr(), dx.data_ptr(), K, T, N, K, s))
    out.append(f"{f:6.1f}|{d:6.1f}")
print("  ".join(out))
