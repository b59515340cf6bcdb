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
for T, K, N in [(32768, 256, 1024), (131072, 128, 384), (8192, 512, 2048)]:
    w = torch.randn(N, K, device=dev) / K ** 0.5; b = torch.randn(N, device=dev)
    pl = [torch.empty(N * K, dtype=torch.bfloat16, device=dev) for _ in range(3)]
    _lib.call("dhz_split3_planes", w.data_ptr(), N * K, pl[0].data_ptr(), pl[1].data_ptr(), pl[2].data_ptr(), s)
    out = []
    for pad in (0, 8, 32, 64, 96):
        for ypad in (0, 32):
            big = torch.randn(T, K + pad, device=dev); x = big[:, :K]
            ybig = torch.empty(T, N + ypad, device=dev)
            t = timeit(lambda: _lib.call("dhz_linear_fwd_split6", x.data_ptr(), K + pad, pl[0].data_ptr(), pl[1].data_ptr(), pl[2].data_ptr(), b.data_ptr(), ybig.data_ptr(), N + ypad, T, N, K, s))
            t32 = timeit(lambda: _lib.call("dhz_linear_fwd", x.data_ptr(), K + pad, w.data_ptr(), b.data_ptr(), ybig.data_ptr(), N + ypad, T, N, K, s))
            out.append(f"lda+{pad} ldy+{ypad}: {t:.1f} (fp32 {t32:.1f})")
    print(T, K, N, " | ".join(out))
