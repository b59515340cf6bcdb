"""-m gpu: the fused LeFF forward kernel (dhz_leff_fused_fwd, csrc/leff_fused.hip) and the backward kernel chain against an fp64 restatement
of M1:873 + M1:496-534 (norm2 -> linear1 -> GELU -> depthwise 3x3 -> GELU -> linear2 -> DropPath scale -> residual) and
against the unfused kernel chain, forward and backward, for every supported width; non-square maps, tiles on the image
border, several tiles per image, with and without a DropPath vector."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _reference(x, norm, mlp, scale, H, W):
    """fp64 on the CPU, autograd for the gradients."""
    B, L, C = x.shape
    xn = F.layer_norm(x, (C,), norm.weight, norm.bias, 1e-5)
    u = F.gelu(F.linear(xn, mlp.linear1[0].weight, mlp.linear1[0].bias))
    m = u.view(B, H, W, -1).permute(0, 3, 1, 2)
    t = F.gelu(F.conv2d(m, mlp.dwconv[0].weight, mlp.dwconv[0].bias, padding=1, groups=m.shape[1]))
    z = t.permute(0, 2, 3, 1).reshape(B, L, -1)
    y = F.linear(z, mlp.linear2[0].weight, mlp.linear2[0].bias)
    return x + (y if scale is None else scale.view(B, 1, 1) * y)


@pytest.mark.parametrize("C,H,W,B,drop", [(32, 16, 16, 2, True), (32, 8, 48, 1, False), (64, 16, 16, 3, True),
                                          (64, 24, 32, 1, True), (128, 16, 16, 2, True), (128, 8, 32, 1, False)])
def test_leff_fused_vs_fp64_and_chain(C, H, W, B, drop):
    import copy
    import My_model_1 as M1
    from dehaze_hip import fused
    dev = torch.device("cuda:0")
    torch.manual_seed(C + H + W)
    norm = torch.nn.LayerNorm(C)
    mlp = M1.LeFF(C, 4 * C)
    with torch.no_grad():
        for p in list(norm.parameters()) + list(mlp.parameters()):
            p.add_(0.1 * torch.randn_like(p))
    x = torch.randn(B, H * W, C)
    gout = torch.randn(B, H * W, C)
    scale = (torch.rand(B) > 0.3).float() / 0.7 if drop else None
    if drop:
        scale[0] = 1.0 / 0.7
    # fp64 reference
    n64, m64 = copy.deepcopy(norm).double(), copy.deepcopy(mlp).double()
    x64 = x.double().requires_grad_()
    y64 = _reference(x64, n64, m64, None if scale is None else scale.double(), H, W)
    (y64 * gout.double()).sum().backward()
    ref = {"y": y64.detach(), "dx": x64.grad}
    for n_, p in list(n64.named_parameters(prefix="norm")) + list(m64.named_parameters(prefix="mlp")):
        ref[n_] = p.grad

    def run(fused_on, p6_c=None):
        saved = (fused.LEFF_FUSED, fused.LEFF_FUSED_C, fused.LEFF_FUSED_P6_C)
        fused.LEFF_FUSED, fused.LEFF_FUSED_C = fused_on, (32, 64, 128)
        if p6_c is not None:
            fused.LEFF_FUSED_P6_C = p6_c
        try:
            nd, md = copy.deepcopy(norm).to(dev), copy.deepcopy(mlp).to(dev)
            xd = x.to(dev).requires_grad_()
            y = fused.leff_branch(xd, nd, md, None if scale is None else scale.to(dev), H, W)
            (y * gout.to(dev)).sum().backward()
            out = {"y": y.detach().cpu().double(), "dx": xd.grad.cpu().double()}
            for n_, p in list(nd.named_parameters(prefix="norm")) + list(md.named_parameters(prefix="mlp")):
                out[n_] = p.grad.cpu().double()
            with torch.no_grad():                       # inference mode: no saves
                out["y_eval"] = fused.leff_branch(x.to(dev), nd, md, None, H, W).cpu().double()
            return out
        finally:
            fused.LEFF_FUSED, fused.LEFF_FUSED_C, fused.LEFF_FUSED_P6_C = saved

    got, chain = run(True), run(False)                  # fused forward + chain backward (the shipped default), and the pure chain
    if C in (32, 64):
        # the other arithmetic of the fused forward at this width: six-term products on the bf16 pipe (round 6, dhz_leff_fused_fwd6; the
        # default at C = 64) / the fp32 pipe - same tolerances against fp64
        other = run(True, p6_c=(32, 64) if C not in fused.LEFF_FUSED_P6_C else ())
        for k, r in ref.items():
            tol = 3e-5 + 3e-5 * r.abs().max().item()
            if k not in ("y", "dx"):
                tol *= (B * H * W) ** 0.5
            assert (other[k] - r).abs().max().item() < tol, ("other arithmetic", k, (other[k] - r).abs().max().item(), tol)
        assert (other["y_eval"] - chain["y_eval"]).abs().max().item() < 3e-5
    for k, r in ref.items():
        tol = 3e-5 + 3e-5 * r.abs().max().item()
        if k not in ("y", "dx"):
            tol *= (B * H * W) ** 0.5                   # sums over all tokens
        assert (got[k] - r).abs().max().item() < tol, (k, (got[k] - r).abs().max().item(), tol)
        assert (chain[k] - r).abs().max().item() < tol, ("chain", k)
    assert (got["y_eval"] - chain["y_eval"]).abs().max().item() < 3e-5


def test_leff_fused_c_abi_argument_checks():
    from dehaze_hip import _lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    a = torch.zeros(64, device=dev)
    p = a.data_ptr()
    s = torch.cuda.current_stream().cuda_stream
    assert lib.dhz_leff_fused_fwd(p, p, p, p, p, p, p, p, p, None, p, None, None, None, None, None, 1, 8, 16, 48, s) == -22
    assert b"supported" in lib.dhz_last_error()
    assert lib.dhz_leff_fused_fwd(p, p, p, p, p, p, p, p, p, None, p, None, None, None, None, None, 1, 12, 16, 32, s) == -22
    assert b"tile" in lib.dhz_last_error()
    assert lib.dhz_leff_fused_fwd(p, p, p, p, p, p, p, p, p, None, p, p, None, None, None, None, 1, 8, 16, 32, s) == -22
