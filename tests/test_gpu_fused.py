"""-m gpu: the fused attention-branch kernel (dhz_fused_window_attn_fwd) against the unfused kernel chain and
against the reference goldens, forward and backward, for every supported width."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
T = torch.from_numpy


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def _run_block(blk, x, idx, gout, fused_on):
    from dehaze_hip import fused
    fused.ENABLED = fused_on
    try:
        for p in blk.parameters():
            p.grad = None
        xx = x.clone().requires_grad_()
        blk._staged_idx = idx
        y = blk(xx)
        (y * gout).sum().backward()
        grads = {n: (p.grad.clone() if p.grad is not None else None) for n, p in blk.named_parameters()}
        return y.detach(), xx.grad.clone(), grads
    finally:
        fused.ENABLED = True


@pytest.mark.parametrize("C,heads,res,shift,B", [(32, 1, 16, 0, 2), (32, 1, 32, 4, 2), (64, 2, 16, 4, 3), (64, 2, 32, 0, 1),
                                               (128, 4, 16, 4, 2), (128, 4, 24, 0, 1)])
def test_fused_equals_unfused_chain(dev, C, heads, res, shift, B):
    import My_model_1 as M1
    torch.manual_seed(C + res + shift)
    blk = M1.LeWinTransformerBlock(dim=C, input_resolution=(res, res), num_heads=heads, win_size=8, shift_size=shift,
                                   token_mlp='leff', drop_path=0.1).to(dev)
    with torch.no_grad():
        for p in blk.parameters():
            if p.ndim == 1:
                p.add_(0.1 * torch.randn_like(p))
    blk.train()
    x = torch.randn(B, res * res, C, device=dev)
    gout = torch.randn(B, res * res, C, device=dev)
    idx = torch.randint(64, (64, 25)).to(torch.uint8).to(dev)
    torch.manual_seed(1); torch.cuda.manual_seed(1)
    y0, dx0, g0 = _run_block(blk, x, idx, gout, False)
    torch.manual_seed(1); torch.cuda.manual_seed(1)
    y1, dx1, g1 = _run_block(blk, x, idx, gout, True)
    assert torch.allclose(y1, y0, atol=2e-5, rtol=1e-4), (y1 - y0).abs().max()
    assert torch.allclose(dx1, dx0, atol=5e-5, rtol=1e-3), (dx1 - dx0).abs().max()
    for n in g0:
        if g0[n] is None:
            assert g1[n] is None, n
        else:
            err = (g1[n] - g0[n]).abs().max().item()
            assert err <= 2e-4 + 2e-3 * g0[n].abs().max().item(), (n, err)
    # inference mode (no saves)
    blk.eval()
    with torch.no_grad():
        blk._staged_idx = idx
        ye = blk(x)
        from dehaze_hip import fused
        fused.ENABLED = False
        blk._staged_idx = idx
        yu = blk(x)
        fused.ENABLED = True
    assert torch.allclose(ye, yu, atol=2e-5, rtol=1e-4)


@pytest.mark.parametrize("C,heads,mode", [(32, 1, "pairs"), (32, 1, "all"), (64, 2, "pairs"), (128, 4, "all")])
def test_fused_rank_ties(dev, C, heads, mode):
    """Windows with bit-equal sparsity measures: duplicated token rows + one shared sample row make M[i] == M[i'] exactly
    (pairs of horizontally adjacent tokens, or all 64 tokens of every window), so the top-25 set is decided by the index
    tie rule (lower index first, ATT:122 / the oracle).  The fused kernel counts larger measures with lane masks and only
    falls back to the per-lane tie rule when the counts reveal a tie - this is the case that takes the fallback."""
    import My_model_1 as M1
    torch.manual_seed(C)
    res, B = 16, 2
    blk = M1.LeWinTransformerBlock(dim=C, input_resolution=(res, res), num_heads=heads, win_size=8, shift_size=0,
                                   token_mlp='leff', drop_path=0.).to(dev).eval()
    if mode == "pairs":
        base = torch.randn(B, res, res // 2, C, device=dev)
        x = base.repeat_interleave(2, dim=2).reshape(B, res * res, C)
    else:
        x = torch.randn(B, 1, C, device=dev).expand(B, res * res, C).contiguous()
    idx = torch.randint(64, (1, 25)).expand(64, 25).contiguous().to(torch.uint8).to(dev)
    from dehaze_hip import fused
    with torch.no_grad():
        blk._staged_idx = idx
        yf = blk(x)
        fused.ENABLED = False
        try:
            blk._staged_idx = idx
            yu = blk(x)
        finally:
            fused.ENABLED = True
    assert torch.isfinite(yf).all()
    assert torch.allclose(yf, yu, atol=2e-5, rtol=1e-4), (yf - yu).abs().max()


@pytest.mark.parametrize("name,heads,shift,C", [("block_m1_c32_shift0", 1, 0, 32), ("block_m1_c32_shift4", 1, 4, 32),
                                                ("block_m1_c64_shift4", 2, 4, 64)])
def test_fused_block_vs_reference_golden(golden, dev, name, heads, shift, C):
    import My_model_1 as M1
    from dehaze_hip import fused
    assert fused.ENABLED
    g = golden(name)
    blk = M1.LeWinTransformerBlock(dim=C, input_resolution=(16, 16), num_heads=heads, win_size=8, shift_size=shift,
                                   token_mlp='leff', drop_path=0.)
    blk.load_state_dict({k[3:]: T(g[k]) for k in g.files if k.startswith("sd/")})
    blk.to(dev)
    x = T(g["x"]).to(dev).requires_grad_()
    blk._staged_idx = T(g["idx"].astype(np.uint8)).to(dev)
    y = blk(x)
    assert torch.allclose(y.cpu(), T(g["y"]), atol=3e-5, rtol=1e-4), (y.cpu() - T(g["y"])).abs().max()
    (y * T(g["gout"]).to(dev)).sum().backward()
    assert torch.allclose(x.grad.cpu(), T(g["dx"]), atol=5e-5, rtol=1e-3)
    for n, p in blk.named_parameters():
        ref = g["g/" + n]
        if ref.size:
            err = (p.grad.cpu() - T(ref)).abs().max().item()
            assert err <= 2e-4 + 2e-3 * np.abs(ref).max(), (n, err)


@pytest.mark.parametrize("name,use_fused", [("block_m1_c128_shift4", True), ("block_m1_c128_shift4", False),
                                            ("block_m1_c256_shift4", False), ("block_m1_c512_shift0", False),
                                            ("block_m1_c16_shift4", True), ("block_m1_c32h2_shift4", True),      # head_dim 16 (Uformer16)
                                            ("block_m1_c64_ffn_shift4", True), ("block_m1_c64_ffn_shift4", False)])   # token_mlp ffn (Mlp)
def test_block_wide_vs_reference_golden(golden, dev, name, use_fused):
    """C = 128 (four heads, shifted windows, 16 x 16): the widest instance of the fused window-attention forward - and the kernel chain;
    C = 256 (eight heads) and C = 512 (sixteen heads, the bottleneck's single 8 x 8 window): the kernel chain every block of those widths
    runs - against the REFERENCE's numbers (tests/golden/block_m1_c*.npz), not only against each other or through the whole model"""
    import My_model_1 as M1
    from dehaze_hip import fused
    from test_oracle_golden import _wide_block_inputs, check_c128_grads
    g = golden(name)
    blk, x, gout = _wide_block_inputs(g, M1, name)
    blk.to(dev)
    x = x.to(dev).requires_grad_()
    blk._staged_idx = T(g["idx"].astype(np.uint8)).to(dev)
    old = fused.ENABLED
    fused.ENABLED = use_fused
    try:
        y = blk(x)
        assert torch.allclose(y.cpu(), T(g["y"]), atol=5e-5, rtol=1e-4), (y.cpu() - T(g["y"])).abs().max()
        (y * gout.to(dev)).sum().backward()
    finally:
        fused.ENABLED = old
    assert torch.allclose(x.grad.cpu(), T(g["dx"]), atol=1e-4, rtol=1e-3)
    check_c128_grads(g, {n: p.grad for n, p in blk.named_parameters()}, 2e-3)
