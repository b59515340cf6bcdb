"""-m gpu: the 4x4 / stride-2 / pad-1 down-sampling convolution on the token layout (csrc/conv_gemm.hip: implicit GEMMs, forward,
backward-data per parity class, gathered weight gradient) against fp64 torch.nn.functional.conv2d and its autograd, through
the raw C-ABI and through the Downsample module."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 16, 16, 32, 64), (1, 32, 16, 64, 128), (3, 8, 8, 128, 256), (2, 64, 64, 32, 64),
                                            (1, 8, 16, 256, 512), (1, 26, 18, 32, 64)])
def test_conv4s2_c_abi(B, H, W, Cin, Cout):
    from dehaze_hip import _lib
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(B + H + W + Cin)
    x = torch.randn(B, H * W, Cin, generator=g)
    w = torch.randn(Cout, Cin, 4, 4, generator=g) * 0.05
    b = torch.randn(Cout, generator=g) * 0.1
    xr = x.double().view(B, H, W, Cin).permute(0, 3, 1, 2).requires_grad_()
    wr, br = w.double().requires_grad_(), b.double().requires_grad_()
    yr = F.conv2d(xr, wr, br, stride=2, padding=1)
    gy = torch.randn(yr.shape, generator=g, dtype=torch.float64)
    (yr * gy).sum().backward()
    s = torch.cuda.current_stream().cuda_stream
    xd = x.to(dev)
    wp = w.permute(0, 2, 3, 1).reshape(Cout, 16 * Cin).contiguous().to(dev)
    y = torch.empty(B, (H // 2) * (W // 2), Cout, device=dev)
    _lib.call("dhz_conv4s2_fwd", xd.data_ptr(), wp.data_ptr(), b.to(dev).data_ptr(), y.data_ptr(), B, H, W, Cin, Cout, s)
    yref = yr.detach().permute(0, 2, 3, 1).reshape(B, -1, Cout)
    tol = 3e-6 * (16 * Cin) ** 0.5 + 1e-5
    assert (y.cpu().double() - yref).abs().max() < tol * max(1.0, yref.abs().max().item())
    dy = gy.permute(0, 2, 3, 1).reshape(B, -1, Cout).float().contiguous().to(dev)
    wq = w.permute(2, 3, 0, 1).contiguous().to(dev)
    dx = torch.full((B, H * W, Cin), 9.0, device=dev)
    _lib.call("dhz_conv4s2_dgrad", dy.data_ptr(), wq.data_ptr(), dx.data_ptr(), B, H, W, Cin, Cout, s)
    dxref = xr.grad.permute(0, 2, 3, 1).reshape(B, H * W, Cin)
    assert (dx.cpu().double() - dxref).abs().max() < (3e-6 * (4 * Cout) ** 0.5 + 1e-5) * max(1.0, dxref.abs().max().item())
    Ho, Wo = H // 2, W // 2
    lib = _lib.load()
    if (Ho & (Ho - 1)) == 0 and (Wo & (Wo - 1)) == 0 and (B * Ho * Wo) % 32 == 0:
        dwp = torch.full((Cout, 16 * Cin), 0.5, device=dev)
        db = torch.full((Cout,), -1.0, device=dev)
        _lib.call("dhz_conv4s2_wgrad", dy.data_ptr(), xd.data_ptr(), dwp.data_ptr(), db.data_ptr(), B, H, W, Cin, Cout, s)
        dwref = wr.grad.permute(0, 2, 3, 1).reshape(Cout, 16 * Cin)
        T = B * Ho * Wo
        assert (dwp.cpu().double() - 0.5 - dwref).abs().max() < 3e-6 * T ** 0.5 * max(1.0, dwref.abs().max().item() / T ** 0.5) + 1e-4
        assert (db.cpu().double() + 1.0 - br.grad).abs().max() < 3e-5 * T ** 0.5 + 1e-4
    else:
        dwp = torch.zeros(Cout, 16 * Cin, device=dev)
        assert lib.dhz_conv4s2_wgrad(dy.data_ptr(), xd.data_ptr(), dwp.data_ptr(), None, B, H, W, Cin, Cout, s) == -22
    assert lib.dhz_conv4s2_fwd(xd.data_ptr(), wp.data_ptr(), None, y.data_ptr(), B, H + 1, W, Cin, Cout, s) == -22


def test_downsample_module_matches_library_conv():
    """Downsample on the token layout: the implicit-GEMM path (default for fp32 tokens) against the library convolution it
    replaces, forward and every gradient, with the weight gradient accumulated in place into an existing .grad."""
    import My_model_1 as M1
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    ds = M1.Downsample(64, 128).to(dev)
    x = torch.randn(2, 32 * 32, 64, device=dev)
    gout = torch.randn(2, 16 * 16, 128, device=dev)
    xa = x.clone().requires_grad_()
    ya = ds(xa)
    ya.backward(gout)
    ga = (ds.conv[0].weight.grad.clone(), ds.conv[0].bias.grad.clone(), xa.grad.clone())
    ds.zero_grad(set_to_none=True)
    xb = x.clone().requires_grad_()
    m = xb.view(2, 32, 32, 64).permute(0, 3, 1, 2)
    yb = ds.conv(m).permute(0, 2, 3, 1).reshape(2, 256, 128)
    yb.backward(gout)
    assert torch.allclose(ya, yb, atol=2e-4, rtol=1e-4), (ya - yb).abs().max()
    assert torch.allclose(ga[2], xb.grad, atol=2e-4, rtol=1e-3)
    assert torch.allclose(ga[0], ds.conv[0].weight.grad, atol=2e-3, rtol=1e-3), (ga[0] - ds.conv[0].weight.grad).abs().max()
    assert torch.allclose(ga[1], ds.conv[0].bias.grad, atol=2e-3, rtol=1e-3)
    # a second backward accumulates on top
    ya2 = ds(x.clone().requires_grad_())
    ya2.backward(gout)
    assert torch.allclose(ds.conv[0].weight.grad, 2 * ga[0], atol=4e-3, rtol=1e-3)


@pytest.mark.parametrize("B,H,W,E", [(2, 32, 32, 32), (1, 48, 40, 64), (3, 16, 16, 32), (1, 128, 128, 32)])
def test_input_proj_vs_torch(B, H, W, E):
    """InputProj (conv3x3 3 -> E + LeakyReLU into tokens, csrc/input_proj.hip) against fp64 conv2d + leaky_relu and autograd,
    through the module (weight / bias gradients accumulated in place) - maps that are not multiples of the 16 x 16 tile included."""
    import My_model_1 as M1
    dev = torch.device("cuda:0")
    torch.manual_seed(B + H + E)
    ip = M1.InputProj(in_channel=3, out_channel=E, kernel_size=3, stride=1, act_layer=torch.nn.LeakyReLU).to(dev)
    img = torch.rand(B, 3, H, W, device=dev)
    gout = torch.randn(B, H * W, E, device=dev)
    y = ip(img)
    y.backward(gout)
    w64, b64 = ip.proj[0].weight.detach().double().cpu().requires_grad_(), ip.proj[0].bias.detach().double().cpu().requires_grad_()
    yr = F.leaky_relu(F.conv2d(img.double().cpu(), w64, b64, padding=1), 0.01).permute(0, 2, 3, 1).reshape(B, H * W, E)
    (yr * gout.double().cpu()).sum().backward()
    assert (y.detach().cpu().double() - yr.detach()).abs().max() < 2e-5
    T = B * H * W
    assert (ip.proj[0].weight.grad.cpu().double() - w64.grad).abs().max() < 3e-6 * T ** 0.5 * max(1.0, w64.grad.abs().max().item() / T ** 0.5) + 1e-4
    assert (ip.proj[0].bias.grad.cpu().double() - b64.grad).abs().max() < 3e-5 * T ** 0.5 + 1e-4
