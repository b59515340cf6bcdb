"""dhz_winograd_conv3x3 vs MIOpen (torch conv2d) on the VGG19 layer shapes, batch 64 (forward) ."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "research-and-implementation-of-image-dehazing-algorithm-based-on-vision-transformer_amd"))
import torch, torch.nn.functional as F
from dehaze_hip import _lib
dev = torch.device("cuda:0"); s = torch.cuda.current_stream().cuda_stream
def timeit(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
B = int(os.environ.get("B", 64))
F43 = os.environ.get("F43")          # F43=1: the F(4x4,3x3) kernel (dhz_winograd43_*) instead of F(2x2,3x3)
PRE, ENT, NPOS = ("dhz_winograd43_prepack", "dhz_winograd43_conv3x3", 36) if F43 else ("dhz_winograd_prepack", "dhz_winograd_conv3x3", 16)
for C, K, H in [(64, 64, 128), (64, 128, 64), (128, 128, 64), (128, 256, 32), (256, 256, 32), (256, 512, 16), (512, 512, 16)]:
    x = torch.randn(B, C, H, H, device=dev); w = torch.randn(K, C, 3, 3, device=dev) * 0.05; b = torch.randn(K, device=dev)
    xb = torch.empty(B, C // 8, H, H, 8, device=dev); yb = torch.empty(B, K // 8, H, H, 8, device=dev)
    up = torch.empty(NPOS * K * C, device=dev)
    _lib.call(PRE, w.data_ptr(), up.data_ptr(), K, C, 0, s)
    t_m = timeit(lambda: _lib.call(ENT, xb.data_ptr(), up.data_ptr(), b.data_ptr(), 1, None, None, yb.data_ptr(), B, H, H, C, K, s))
    t_l = timeit(lambda: F.relu(F.conv2d(x, w, b, padding=1))) if not os.environ.get("NO_LIB") else float("nan")
    fl = 2.0 * B * H * H * C * K * 9
    print(f"C {C:4d} K {K:4d} H {H:4d}: wino-mfma {t_m:8.1f} us ({fl/t_m/1e6:6.1f} TF-equiv)   miopen+relu {t_l:8.1f} us ({fl/t_l/1e6:6.1f} TF-equiv)   x{t_l/t_m:.2f}")
