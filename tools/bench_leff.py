"""Micro-benchmark of the LeFF depthwise stage kernels at the model's shapes (bs=32)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "research-and-implementation-of-image-dehazing-algorithm-based-on-vision-transformer_amd"))
import torch
from dehaze_hip import ops
dev = torch.device("cuda:0")
tot_f = tot_b = 0
for res, C in [(128, 32), (64, 64), (32, 128), (16, 256), (8, 512), (16, 512), (32, 256), (64, 128), (128, 64)]:
    Ch = 4 * C
    u = torch.randn(32, res * res, Ch, device=dev, requires_grad=True)
    w = torch.randn(Ch, 1, 3, 3, device=dev, requires_grad=True); b = torch.randn(Ch, device=dev, requires_grad=True)
    z = ops.leff_dwconv(u, w, b, res, res); go = torch.randn_like(z)
    def fwd(): return ops.leff_dwconv(u, w, b, res, res)
    def bwd(): z.backward(go, retain_graph=True)
    out = []
    for f in (fwd, bwd):
        for _ in range(2): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): f()
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / 5 * 1e3)
    mb = u.numel() * 4 / 1e6
    tot_f += out[0]; tot_b += out[1]
    print(f"res {res:4d} Ch {Ch:5d} ({mb:6.1f} MB/tensor)  fwd {out[0]:7.1f} us ({3*mb/out[0]*1e3:5.0f} GB/s)   bwd {out[1]:7.1f} us ({4*mb/out[1]*1e3:5.0f} GB/s)")
print(f"sum fwd {tot_f:.0f} us  bwd {tot_b:.0f} us  (x2 blocks per stage per step)")
