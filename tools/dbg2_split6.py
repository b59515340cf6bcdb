import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "research-and-implementation-of-image-dehazing-algorithm-based-on-vision-transformer_amd"))
import torch
from dehaze_hip import _lib
dev = torch.device("cuda:0"); s = torch.cuda.current_stream().cuda_stream
out = []
for T, K, N in [(1024, 64, 128), (4096, 128, 128), (131072, 64, 64), (32768, 256, 1024)]:
    g = torch.Generator().manual_seed(T + K + N)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev); dy = torch.randn(T, N, generator=g).to(dev)
    pl = [torch.empty(N * K, dtype=torch.bfloat16, device=dev) for _ in range(3)]
    _lib.call("dhz_split3_planes", w.data_ptr(), N * K, pl[0].data_ptr(), pl[1].data_ptr(), pl[2].data_ptr(), s)
    dx = torch.zeros(T, K, device=dev)
    ref = dy.double() @ w.double()
    for rep in range(2):
        _lib.call("dhz_linear_dgrad_split6", dy.data_ptr(), N, pl[0].data_ptr(), pl[1].data_ptr(), pl[2].data_ptr(), dx.data_ptr(), K, T, N, K, s)
        torch.cuda.synchronize()
        e = (dx.double() - ref).abs()
        bad = (e > 1e-3)
        out.append("%.1e(%d bad; rows %s cols %s)" % (e.max().item(), bad.sum().item(), sorted(set((bad.nonzero()[:, 0] % 128).tolist()))[:6], sorted(set((bad.nonzero()[:, 1]).tolist()))[:8]))
print(" | ".join(out))
