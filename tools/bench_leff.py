"""Fused LeFF kernels against the kernel chain on the config-2 stage shapes (bs = 32): forward, forward+backward (us)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "research-and-implementation-of-image-dehazing-algorithm-based-on-vision-transformer_amd"), ROOT]
import torch
import My_model_1 as M1
from dehaze_hip import fused
dev = torch.device("cuda:0")


def bench(f, reps=5):
    f(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            f()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best


print(f"{'C':>4} {'res':>4} | fwd chain  fused | fwd+bwd chain  fused   (us; MFMA floor fwd at 157 TF)")
for C, res in [(32, 128), (64, 64), (128, 32), (128, 64), (64, 128)]:
    B = 32
    norm = torch.nn.LayerNorm(C).to(dev)
    mlp = M1.LeFF(C, 4 * C).to(dev)
    x = torch.randn(B, res * res, C, device=dev, requires_grad=True)
    g = torch.randn(B, res * res, C, device=dev)
    sc = torch.ones(B, device=dev)
    r = []
    fused.LEFF_FUSED_C = (32, 64, 128); fused.LEFF_FUSED_C64_MAX_T = 1 << 30      # (the tool measures the fused kernel wherever it exists)
    for on in (False, True):
        fused.LEFF_FUSED = on
        def fwd():
            with torch.no_grad():
                fused.leff_branch(x, norm, mlp, sc, res, res)
        def fb():
            y = fused.leff_branch(x, norm, mlp, sc, res, res)
            y.backward(g)
            x.grad = None
        def ftrain():
            fused.leff_branch(x, norm, mlp, sc, res, res)
        r += [bench(fwd), bench(fb), bench(ftrain)]
    fused.LEFF_FUSED = True
    floor = 2 * B * res * res * 8 * C * C / 157.3e12 * 1e6
    print(f"{C:4d} {res:4d} | {r[0]:9.1f} {r[3]:6.1f} | {r[1]:13.1f} {r[4]:6.1f}   ({floor:.0f}) | fwd with saves: chain {r[2]:.1f} fused {r[5]:.1f}")
