"""dhz_linear_wgrad_bf16 on config-4 shapes, us per call (CMD of tools/variants.sh SRC=linear_bf16 VARIANTS="-DBF_ABL=..")"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..",
                                "research-and-implementation-of-image-dehazing-algorithm-based-on-vision-transformer_amd"))
import torch
from dehaze_hip import _lib
dev = torch.device("cuda:0"); s = torch.cuda.current_stream().cuda_stream
tot = 0.0
for T, N, K in [(524288, 256, 64), (524288, 64, 256), (524288, 192, 64), (131072, 512, 128), (131072, 128, 512), (32768, 1024, 256),
                (8192, 2048, 512), (8192, 4096, 1024), (2048, 4096, 1024)]:
    dy = torch.randn(T, N, device=dev).bfloat16(); x = torch.randn(T, K, device=dev).bfloat16()
    dw = torch.zeros(N, K, device=dev); db = torch.zeros(N, device=dev)
    pw = (ctypes.c_void_p * 1)(dw.data_ptr()); pb = (ctypes.c_void_p * 1)(db.data_ptr())
    f = lambda: _lib.call("dhz_linear_wgrad_bf16", dy.data_ptr(), N, x.data_ptr(), K, T, 1, N, K, ctypes.cast(pw, ctypes.c_void_p),
                          ctypes.cast(pb, ctypes.c_void_p), s)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    tot += us
    print(f"T={T:7d} N={N:5d} K={K:5d}  {us:7.1f} us  {T * (N + K) * 2 / us / 1e6:5.2f} TB/s  {2 * T * N * K / us / 1e6:6.1f} TF")
print(f"sum {tot:.0f} us")
