"""Attention branch of a LeWin block, forward + backward, us per call (bs 32): fused forward + backward kernel CHAIN (the forward
saves xn / QKV / context / statistics) against fused forward + FUSED backward (csrc/fused_attn_bwd.hip: only the ranks are saved),
and the forward alone in both save modes.      python tools/bench_fused_bwd.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "research-and-implementation-of-image-dehazing-algorithm-based-on-vision-transformer_amd"))
import torch
import My_model_1 as M1
from dehaze_hip import fused
dev = torch.device("cuda:0")


def timeit(f, n=10, rounds=3):
    best = 1e9
    for _ in range(3): f()
    torch.cuda.synchronize()
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


print(f"{'res':>4} {'C':>4} {'shift':>5} | fwd(all saves) fwd(ranks only) | fwd+bwd chain  fwd+bwd fused |  bwd chain  bwd fused")
for res, C, heads in [(128, 32, 1), (64, 32, 1)]:
    for shift in (0, 4):
        blk = M1.LeWinTransformerBlock(dim=C, input_resolution=(res, res), num_heads=heads, win_size=8, shift_size=shift, token_mlp='leff', drop_path=0.).to(dev)
        x = torch.randn(32, res * res, C, device=dev)
        g = torch.randn(32, res * res, C, device=dev)
        idx = torch.randint(64, (64, 25)).to(torch.uint8).to(dev)
        mask = blk._shift_mask(res, res, dev) if shift else None
        sc = torch.ones(32, device=dev)
        def fwd():
            xx = x.detach().requires_grad_()
            return xx, fused.fused_attn_branch(xx, blk.norm1, blk.attn.ProbSpare, blk.attn.relative_position_bias_table, idx, mask, sc, res, res, shift, heads)
        def fb():
            xx, y = fwd()
            y.backward(g)
        r = {}
        for mode, cfg in (("chain", ()), ("fused", (32,))):
            fused.ATTN_FUSED_BWD_C = cfg
            for p in blk.parameters(): p.grad = None
            r[mode] = (timeit(lambda: fwd()), timeit(fb))
        fused.ATTN_FUSED_BWD_C = (32,)
        print(f"{res:4d} {C:4d} {shift:5d} | {r['chain'][0]:14.1f} {r['fused'][0]:15.1f} | {r['chain'][1]:13.1f} {r['fused'][1]:14.1f} | "
              f"{r['chain'][1] - r['chain'][0]:10.1f} {r['fused'][1] - r['fused'][0]:10.1f}")
