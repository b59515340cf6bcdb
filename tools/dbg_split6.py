import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "research-and-implementation-of-image-dehazing-algorithm-based-on-vision-transformer_amd"))
import torch
from dehaze_hip import _lib
dev = torch.device("cuda:0"); s = torch.cuda.current_stream().cuda_stream
shapes = [(128, 32, 32), (128, 64, 64), (256, 64, 64), (1024, 64, 128), (4096, 128, 128), (131072, 64, 64), (131072, 64, 192), (777, 128, 512), (5000, 192, 64), (64, 1024, 256), (32768, 256, 1024)]
for T, K, N in shapes:
    g = torch.Generator().manual_seed(T + K + N)
    x = torch.randn(T, K, generator=g).to(dev); w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev); b = torch.randn(N, generator=g).to(dev)
    dy = torch.randn(T, N, generator=g).to(dev)
    pl = [torch.empty(N * K, dtype=torch.bfloat16, device=dev) for _ in range(3)]
    _lib.call("dhz_split3_planes", w.data_ptr(), N * K, pl[0].data_ptr(), pl[1].data_ptr(), pl[2].data_ptr(), s)
    torch.cuda.synchronize()
    rec = (pl[0].float() + pl[1].float() + pl[2].float()).view(N, K)
    assert torch.equal(rec, w), "planes"
    y = torch.zeros(T, N, device=dev)
    print(T, K, N, "fwd", end=" ", flush=True)
    for rep in range(3):
        _lib.call("dhz_linear_fwd_split6", x.data_ptr(), K, pl[0].data_ptr(), pl[1].data_ptr(), pl[2].data_ptr(), b.data_ptr(), y.data_ptr(), N, T, N, K, s)
        torch.cuda.synchronize()
        ref = x.double() @ w.double().t() + b.double()
        print("%.1e" % (y.double() - ref).abs().max().item(), end=" ", flush=True)
    if K % 64 == 0:
        dx = torch.zeros(T, K, device=dev)
        print("dgrad", end=" ", flush=True)
        for rep in range(3):
            _lib.call("dhz_linear_dgrad_split6", dy.data_ptr(), N, pl[0].data_ptr(), pl[1].data_ptr(), pl[2].data_ptr(), dx.data_ptr(), K, T, N, K, s)
            torch.cuda.synchronize()
            ref = dy.double() @ w.double()
            print("%.1e" % (dx.double() - ref).abs().max().item(), end=" ", flush=True)
    print(flush=True)
