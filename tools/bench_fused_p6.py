"""The fused window-attention forward at C = 64 with the QKV product of every head on the bf16 matrix pipe (six-term planes by LDS-DMA
into the dead Q / K / V / S tiles: fused.ATTN_FUSED_P6) against the fp32-pipe projections: time per launch (bs 32; training with the saves
of the backward kernel chain, inference) and the difference of the outputs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "research-and-implementation-of-image-dehazing-algorithm-based-on-vision-transformer_amd"))
import torch
import My_model_1 as M1
from dehaze_hip import fused, ops
dev = torch.device("cuda:0")


def timeit(f, n=20):
    for _ in range(3):
        f()
    ops.KERNEL_TIMING = {"dhz_fused_window_attn_fwd": []}
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    ev, ops.KERNEL_TIMING = ops.KERNEL_TIMING["dhz_fused_window_attn_fwd"], None
    return 1e3 * sum(a.elapsed_time(b) for a, b, _, _ in ev) / len(ev)


for res, C, heads in [(64, 64, 2), (128, 64, 2), (32, 128, 4), (128, 32, 1)]:
    for shift in (0, 4):
        torch.manual_seed(res + shift)
        blk = M1.LeWinTransformerBlock(dim=C, input_resolution=(res, res), num_heads=heads, win_size=8, shift_size=shift, token_mlp='leff',
                                       drop_path=0.).to(dev)
        x = torch.randn(32, res * res, C, device=dev)
        idx = torch.randint(64, (64, 25)).to(torch.uint8).to(dev)
        mask = blk._shift_mask(res, res, dev) if shift else None
        tab = blk.attn.relative_position_bias_table

        def train():
            xx = x.detach().requires_grad_()
            return fused.fused_attn_branch(xx, blk.norm1, blk.attn.ProbSpare, tab, idx, mask, None, res, res, shift, heads)

        def infer():
            with torch.no_grad():
                return fused.fused_attn_branch(x, blk.norm1, blk.attn.ProbSpare, tab, idx, mask, None, res, res, shift, heads)
        out = {}
        for p6 in (False, True):
            fused.ATTN_FUSED_P6 = p6
            out[p6] = (timeit(train), timeit(infer), infer().clone())
        d = (out[True][2] - out[False][2]).abs().max().item()
        nwin = 32 * (res // 8) ** 2
        fl = nwin * 2 * 64 * (4 * C * C + 75 * C)
        print(f"res {res:4d} C {C} shift {shift}: training {out[False][0]:7.1f} -> {out[True][0]:7.1f} us ({fl / out[True][0] / 1e6:5.1f} TF = "
              f"{fl / out[True][0] / 1e6 / 157.3:.3f}), inference {out[False][1]:7.1f} -> {out[True][1]:7.1f} us ({fl / out[True][1] / 1e6 / 157.3:.3f}); "
              f"max |out6 - out32| = {d:.2e} (max |out| {out[False][2].abs().max().item():.2f})")
