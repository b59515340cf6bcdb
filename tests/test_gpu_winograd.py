"""-m gpu: the Winograd F(2x2,3x3) MFMA convolution (dhz_winograd_conv3x3) vs torch conv2d: forward (+bias, +ReLU)
and the backward-data pass (rotated/transposed filters + ReLU mask)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _blocked(x):
    from dehaze_hip import _lib
    B, C, H, W = x.shape
    out = torch.empty(B, C // 8, H, W, 8, device=x.device)
    _lib.call("dhz_layout_blocked8", x.contiguous().data_ptr(), out.data_ptr(), B, C, H * W, 1, torch.cuda.current_stream().cuda_stream)
    return out


def _plain(xb, C):
    from dehaze_hip import _lib
    B, CG, H, W, _ = xb.shape
    out = torch.empty(B, C, H, W, device=xb.device)
    _lib.call("dhz_layout_blocked8", xb.data_ptr(), out.data_ptr(), B, C, H * W, 0, torch.cuda.current_stream().cuda_stream)
    return out


@pytest.mark.parametrize("B,C,K,H", [(2, 64, 64, 32), (1, 64, 128, 16), (2, 128, 128, 16), (1, 256, 512, 16), (3, 8, 32, 48)])
def test_winograd_forward_and_dgrad(B, C, K, H):
    from dehaze_hip import _lib
    dev = torch.device("cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(C + K + H)
    x = torch.randn(B, C, H, H, generator=g).to(dev)
    w = (torch.randn(K, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5).to(dev)
    b = (0.1 * torch.randn(K, generator=g)).to(dev)
    up = torch.empty(16 * K * C, device=dev)
    _lib.call("dhz_winograd_prepack", w.data_ptr(), up.data_ptr(), K, C, 0, s)
    xb = _blocked(x)
    yb = torch.empty(B, K // 8, H, H, 8, device=dev)
    _lib.call("dhz_winograd_conv3x3", xb.data_ptr(), None, up.data_ptr(), b.data_ptr(), yb.data_ptr(), B, H, H, C, K, 1, s)
    y = _plain(yb, K)
    ref = F.relu(F.conv2d(x.double(), w.double(), b.double(), padding=1)).float()
    assert torch.allclose(y, ref, atol=2e-5, rtol=1e-4), (y - ref).abs().max()
    assert torch.equal(_plain(xb, C), x)
    # backward-data: dx = conv_transpose(dy * (y > 0), w) == winograd with transposed_rot filters and the ReLU mask
    dy = torch.randn(B, K, H, H, generator=g).to(dev)
    if C % 32 == 0:
        upt = torch.empty(16 * K * C, device=dev)
        _lib.call("dhz_winograd_prepack", w.data_ptr(), upt.data_ptr(), C, K, 1, s)      # Kout = C (of fwd), Cin = K
        dxb = torch.empty(B, C // 8, H, H, 8, device=dev)
        _lib.call("dhz_winograd_conv3x3", _blocked(dy).data_ptr(), yb.data_ptr(), upt.data_ptr(), None, dxb.data_ptr(), B, H, H, K, C, 0, s)
        dx = _plain(dxb, C)
        refdx = F.conv_transpose2d((dy * (ref > 0)).double(), w.double(), padding=1).float()
        assert torch.allclose(dx, refdx, atol=5e-5, rtol=1e-4), (dx - refdx).abs().max()
