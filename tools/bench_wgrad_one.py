"""One-shape loop for PMC profiling of dhz_linear_wgrad."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "research-and-implementation-of-image-dehazing-algorithm-based-on-vision-transformer_amd"))
import torch
from dehaze_hip import _lib
dev = torch.device("cuda:0")
T, N, K = [int(v) for v in sys.argv[1:4]]
dy = torch.randn(T, N, device=dev); x = torch.randn(T, K, device=dev)
dw = torch.zeros(N, K, device=dev); db = torch.zeros(N, device=dev)
s = torch.cuda.current_stream().cuda_stream
for _ in range(10):
    _lib.call("dhz_linear_wgrad", dy.data_ptr(), N, x.data_ptr(), K, T, N, K, dw.data_ptr(), db.data_ptr(), s)
torch.cuda.synchronize()
