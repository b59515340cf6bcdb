"""Does running the LeFF chain on batch chunks (intermediates resident in the 256 MB Infinity Cache) pay?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "research-and-implementation-of-image-dehazing-algorithm-based-on-vision-transformer_amd"))
import torch
from dehaze_hip import ops
dev = torch.device("cuda:0")
def timeit(f, n=5):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for res, C in [(128, 32), (64, 64), (32, 128), (128, 64), (64, 128)]:
    B = 32
    x = torch.randn(B, res * res, C, device=dev, requires_grad=True)
    w1 = torch.randn(4 * C, C, device=dev) * 0.1; b1 = torch.zeros(4 * C, device=dev)
    wd = torch.randn(4 * C, 1, 3, 3, device=dev) * 0.1; bd = torch.zeros(4 * C, device=dev)
    w2 = torch.randn(C, 4 * C, device=dev) * 0.1; b2 = torch.zeros(C, device=dev)
    for p_ in (w1, b1, wd, bd, w2, b2): p_.requires_grad_()
    go = torch.randn(B, res * res, C, device=dev)
    def chain(xc):
        Bc, L, _ = xc.shape
        u = ops.linear_tokens(xc.reshape(Bc * L, C), w1, b1).view(Bc, L, 4 * C)
        z = ops.leff_dwconv(u, wd, bd, res, res)
        return ops.linear_tokens(z.view(Bc * L, 4 * C), w2, b2).view(Bc, L, C)
    for nchunk in (1, 2, 4, 8):
        def fwd():
            with torch.no_grad():
                for xc in x.chunk(nchunk): chain(xc)
        def fb():
            for xc, g in zip(x.detach().chunk(nchunk), go.chunk(nchunk)):
                xc = xc.requires_grad_()
                chain(xc).backward(g)
        print(f"res {res} C {C} chunks {nchunk}: fwd {timeit(fwd):8.1f} us   fwd+bwd {timeit(fb):8.1f} us")
