"""bf16 weight gradients (dhz_linear_wgrad_bf16) on the config-4 step's shapes: us per call and TFLOP/s; 'check' compares dW / db with an
fp32 torch reference on the same bf16 operands.  DHZ_BF16_WGRAD_PIPE=0 in a second process: the round-2 kernel alone."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "research-and-implementation-of-image-dehazing-algorithm-based-on-vision-transformer_amd"))
import torch
from dehaze_hip import _lib
dev = torch.device("cuda:0"); s = torch.cuda.current_stream().cuda_stream
check = "check" in sys.argv
shapes = []
for T, C in [(524288, 64), (131072, 128), (32768, 256), (8192, 512), (2048, 1024), (8192, 1024), (32768, 512), (131072, 256), (524288, 128)]:
    shapes += [(T, 3 * C, C, 3), (T, C, C, 1), (T, 4 * C, C, 1), (T, C, 4 * C, 1)]          # (T, N, K, nmat)
tot = 0.0
for T, N, K, nmat in shapes:
    dy = (torch.randn(T, N, device=dev) * 0.5).bfloat16(); x = torch.randn(T, K, device=dev).bfloat16()
    nper = N // nmat
    dws = [torch.zeros(nper, K, device=dev) for _ in range(nmat)]; dbs = [torch.zeros(nper, device=dev) for _ in range(nmat)]
    pw = (ctypes.c_void_p * nmat)(*[t.data_ptr() for t in dws]); pb = (ctypes.c_void_p * nmat)(*[t.data_ptr() for t in dbs])
    f = lambda: _lib.call("dhz_linear_wgrad_bf16", dy.data_ptr(), N, x.data_ptr(), K, T, nmat, nper, K, ctypes.cast(pw, ctypes.c_void_p),
                          ctypes.cast(pb, ctypes.c_void_p), s)
    err = ""
    if check:
        f(); torch.cuda.synchronize()
        ref = dy.float().t() @ x.float(); refb = dy.float().sum(0)
        got = torch.cat(dws, 0); gotb = torch.cat(dbs, 0)
        e = (got - ref).abs().max().item() / ref.abs().max().item(); eb = (gotb - refb).abs().max().item() / refb.abs().max().item()
        err = f"  rel err dW {e:.1e} db {eb:.1e}"
        assert e < 2e-4 and eb < 2e-4, (T, N, K, e, eb)
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    tot += us
    print(f"T={T:7d} N={N:5d} K={K:5d} | {us:7.1f} us {2 * T * N * K / us / 1e6:6.0f} TF{err}", flush=True)
print(f"sum {tot:.0f} us")
