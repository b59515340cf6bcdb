"""Fused attention-branch forward kernel vs the unfused chain, per stage shape (bs=32), training and inference."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "research-and-implementation-of-image-dehazing-algorithm-based-on-vision-transformer_amd"))
import torch
import My_model_1 as M1
from dehaze_hip import fused
dev = torch.device("cuda:0")
def timeit(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for res, C, heads in [(128, 32, 1), (64, 64, 2), (32, 128, 4), (64, 128, 4), (128, 64, 2)]:
    for shift in (0, 4):
        blk = M1.LeWinTransformerBlock(dim=C, input_resolution=(res, res), num_heads=heads, win_size=8, shift_size=shift, token_mlp='leff', drop_path=0.).to(dev)
        x = torch.randn(32, res * res, C, device=dev)
        idx = torch.randint(64, (64, 25)).to(torch.uint8).to(dev)
        mask = blk._shift_mask(res, res, dev) if shift else None
        def fwd_fused_train():
            xx = x.detach().requires_grad_()
            return fused.fused_attn_branch(xx, blk.norm1, blk.attn.ProbSpare, blk.attn.relative_position_bias_table, idx, mask, None, res, res, shift, heads)
        def fwd_fused_eval():
            with torch.no_grad():
                return fused.fused_attn_branch(x, blk.norm1, blk.attn.ProbSpare, blk.attn.relative_position_bias_table, idx, mask, None, res, res, shift, heads)
        def fwd_unfused():
            from dehaze_hip import ops
            xx = x.detach().requires_grad_()
            xw = ops.ln_partition(xx, blk.norm1.weight, blk.norm1.bias, res, res, shift)
            aw = blk.attn(xw.view(-1, 64, C), mask=mask, idx=idx)
            return ops.reverse_residual(aw.reshape(-1, C), xx, None, res, res, shift)
        nwin = 32 * (res // 8) ** 2
        fl = nwin * 2 * 64 * (4 * C * C + 75 * C)
        tt, te, tu = timeit(fwd_fused_train), timeit(fwd_fused_eval), timeit(fwd_unfused)
        print(f"res {res:4d} C {C:4d} shift {shift}: fused train {tt:7.1f} us ({fl/tt/1e6:5.1f} TF)  fused eval {te:7.1f} us ({fl/te/1e6:5.1f} TF)  unfused chain {tu:7.1f} us")
