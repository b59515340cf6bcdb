"""The fused LeFF forward with its two weight products six-term on the bf16 matrix pipe (fused.LEFF_FUSED_P6, csrc/leff_fused.hip L6) against
the fp32-pipe form of the same kernel and against the kernel chain: forward with the training saves, inference (us, bs 32), output difference."""
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


for C, res in [(32, 128), (64, 64), (64, 128)]:
    B = 32
    torch.manual_seed(C + res)
    norm = torch.nn.LayerNorm(C).to(dev)
    mlp = M1.LeFF(C, 4 * C).to(dev)
    x = torch.randn(B, res * res, C, device=dev, requires_grad=True)
    sc = torch.ones(B, device=dev)
    fused.LEFF_FUSED_C = (32, 64); fused.LEFF_FUSED_C64_MAX_T = 1 << 30
    out = {}
    for name, on, p6 in (("chain", False, False), ("fused fp32", True, False), ("fused six-term", True, True)):
        fused.LEFF_FUSED, fused.LEFF_FUSED_P6 = on, p6
        def infer():
            with torch.no_grad():
                return fused.leff_branch(x, norm, mlp, sc, res, res)
        def train():
            return fused.leff_branch(x, norm, mlp, sc, res, res)
        out[name] = (bench(train), bench(infer), infer().clone())
    fused.LEFF_FUSED = fused.LEFF_FUSED_P6 = True
    ref = out["chain"][2]
    print(f"C {C:3d} res {res:3d}: " + "; ".join(f"{k}: train {v[0]:6.1f} infer {v[1]:6.1f} us" for k, v in out.items())
          + f"; max |six-term - fp32| = {(out['fused six-term'][2] - out['fused fp32'][2]).abs().max().item():.2e}, "
            f"max |six-term - chain| = {(out['fused six-term'][2] - ref).abs().max().item():.2e} (max |out| {ref.abs().max().item():.2f})")
