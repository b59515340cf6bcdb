"""GPU side of tools/wino_error_table.py: the product's Winograd kernel on the same seeded shapes, error against the fp64 direct
convolution (calibrates the CPU emulation's accumulation-order assumption against the MFMA accumulation)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "research-and-implementation-of-image-dehazing-algorithm-based-on-vision-transformer_amd"))
import torch
import torch.nn.functional as F
from dehaze_hip import _lib

dev = torch.device("cuda:0")
s = torch.cuda.current_stream().cuda_stream
ENTRY = sys.argv[1] if len(sys.argv) > 1 else "dhz_winograd_conv3x3"
PREP = sys.argv[2] if len(sys.argv) > 2 else "dhz_winograd_prepack"


def blocked(x):
    B, C, H, W = x.shape
    out = torch.empty(B, C // 8, H, W, 8, device=x.device)
    _lib.call("dhz_layout_blocked8", x.contiguous().data_ptr(), out.data_ptr(), B, C, H * W, 1, None, 0, s)
    return out


def plain(xb, C):
    B, CG, H, W, _ = xb.shape
    out = torch.empty(B, C, H, W, device=xb.device)
    _lib.call("dhz_layout_blocked8", xb.data_ptr(), out.data_ptr(), B, C, H * W, 0, None, 0, s)
    return out


for (B, C, K, H) in [(2, 64, 64, 32), (1, 64, 128, 16), (2, 128, 128, 16), (1, 256, 512, 16), (1, 512, 512, 16), (4, 512, 512, 16), (2, 256, 256, 32)]:
    g = torch.Generator().manual_seed(C + K + H)
    x = torch.randn(B, C, H, H, generator=g).to(dev)
    w = (torch.randn(K, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5).to(dev)
    up = torch.empty(36 * K * C, device=dev)
    _lib.call(PREP, w.data_ptr(), up.data_ptr(), K, C, 0, s)
    yb = torch.empty(B, K // 8, H, H, 8, device=dev)
    _lib.call(ENTRY, blocked(x).data_ptr(), up.data_ptr(), None, 0, None, None, yb.data_ptr(), B, H, H, C, K, s)
    y = plain(yb, K).double()
    ref = F.conv2d(x.double(), w.double(), padding=1)
    yd = F.conv2d(x, w, padding=1).double()
    err = (y - ref).abs()
    tol = 2e-5 + 1e-4 * ref.abs()
    print(f"{ENTRY} C={C:4d} K={K:4d} {H:3d}x{H:<3d} B={B}  max {err.max().item():.2e}  rms {err.pow(2).mean().sqrt().item():.2e}  "
          f"(library direct fp32: max {(yd - ref).abs().max().item():.2e} rms {(yd - ref).pow(2).mean().sqrt().item():.2e})  err/tol {(err / tol).max().item():.2f}")
