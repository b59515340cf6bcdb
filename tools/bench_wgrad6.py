"""six-term weight gradients: the pipelined kernel (csrc/linear_split.hip::wgrad_split6_kernel, round 4) against the round-3 kernel
(DHZ_WGRAD6_OLD=1 in a second process) and the fp32 pipe, on the step's shapes.  us per call; 'check' adds the error against fp64."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "research-and-implementation-of-image-dehazing-algorithm-based-on-vision-transformer_amd"))
import torch
from dehaze_hip import _lib
dev = torch.device("cuda:0"); s = torch.cuda.current_stream().cuda_stream
check = "check" in sys.argv
only = [int(a) for a in sys.argv[1:] if a.isdigit()]
shapes = []
for T, C in [(131072, 64), (32768, 128), (8192, 256), (2048, 512), (8192, 512), (32768, 256), (131072, 128), (524288, 64)]:
    if only and C not in only: continue
    shapes += [(T, 3 * C, C, 3), (T, C, C, 1), (T, 4 * C, C, 1), (T, C, 4 * C, 1)]
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
tot = [0.0, 0.0]
for T, N, K, nmat in shapes:
    dy = torch.randn(T, N, device=dev); x = torch.randn(T, K, device=dev)
    nper = N // nmat
    dws = [torch.zeros(nper, K, device=dev) for _ in range(nmat)]; dbs = [torch.zeros(nper, device=dev) for _ in range(nmat)]
    pw = (ctypes.c_void_p * nmat)(*[t.data_ptr() for t in dws]); pb = (ctypes.c_void_p * nmat)(*[t.data_ptr() for t in dbs])
    f6 = lambda: _lib.call("dhz_linear_wgrad_split", dy.data_ptr(), N, x.data_ptr(), K, T, nmat, nper, K, ctypes.cast(pw, ctypes.c_void_p),
                           ctypes.cast(pb, ctypes.c_void_p), None, 0, 6, s)
    dw32 = torch.zeros(N, K, device=dev); db32 = torch.zeros(N, device=dev)
    f32 = lambda: _lib.call("dhz_linear_wgrad", dy.data_ptr(), N, x.data_ptr(), K, T, N, K, dw32.data_ptr(), db32.data_ptr(), s)
    err = ""
    if check:
        for t_ in dws: t_.zero_()
        f6(); torch.cuda.synchronize()
        ref = dy.double().t() @ x.double()
        e6 = (torch.cat(dws, 0).double() - ref).abs().max().item()
        dw32.zero_(); f32(); torch.cuda.synchronize()
        e32 = (dw32.double() - ref).abs().max().item()
        err = f"  err6 {e6:.2e} err32 {e32:.2e}"
    a, b = timeit(f6), timeit(f32)
    tot[0] += a; tot[1] += b
    print(f"T={T:7d} N={N:5d} K={K:5d} | six-term {a:7.1f} us ({12 * T * N * K / a / 1e9:6.0f} TF bf16 issued) | fp32 pipe {b:7.1f} us{err}", flush=True)
print(f"sum: six-term {tot[0]:.0f} us, fp32 pipe {tot[1]:.0f} us")
