"""dhz_ln_partition_bwd / fwd bandwidth on the model's shapes (bs=32)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "research-and-implementation-of-image-dehazing-algorithm-based-on-vision-transformer_amd"))
import torch
from dehaze_hip import _lib
dev = torch.device("cuda:0"); s = torch.cuda.current_stream().cuda_stream
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
B = 32
for res, C in [(128, 32), (64, 64), (32, 128), (16, 256), (8, 512), (16, 512), (32, 256), (64, 128), (128, 64)]:
    T = B * res * res
    x = torch.randn(T, C, device=dev); dy = torch.randn(T, C, device=dev); dres = torch.randn(T, C, device=dev)
    gamma = torch.randn(C, device=dev); stats = torch.rand(T, 2, device=dev) + 0.5
    dx = torch.empty_like(x); dg = torch.zeros(C, device=dev); db = torch.zeros(C, device=dev)
    for part in (1, 0):
        t = timeit(lambda: _lib.call("dhz_ln_partition_bwd", dy.data_ptr(), x.data_ptr(), gamma.data_ptr(), stats.data_ptr(), dres.data_ptr(),
                                     dx.data_ptr(), dg.data_ptr(), db.data_ptr(), B, res, res, C, 4 if part else 0, part, s))
        gb = T * C * 4 * 4 / 1e9
        print(f"res {res:4d} C {C:4d} partition {part}: bwd {t:7.1f} us  {gb / t * 1e6:7.0f} GB/s")
