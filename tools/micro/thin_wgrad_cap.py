"""thin-convolution weight gradient, us per call at config 2's shape (CMD of tools/variants.sh SRC=thin_conv VARIANTS="-DTW_ABL=..")"""
import sys, os
sys.path.insert(0, "research-and-implementation-of-image-dehazing-algorithm-based-on-vision-transformer_amd")
import torch
from dehaze_hip import _lib
dev = torch.device("cuda:0"); s = torch.cuda.current_stream().cuda_stream
B, H, W, C = 32, 128, 128, 64
x = torch.randn(B, H * W, C, device=dev); dy = torch.randn(B, 3, H, W, device=dev)
dw = torch.zeros(3, C, 3, 3, device=dev); db = torch.zeros(3, device=dev)
def run(): _lib.call("dhz_thin_conv3x3_wgrad_dt", dy.data_ptr(), x.data_ptr(), dw.data_ptr(), db.data_ptr(), B, H, W, C, 0, s)
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
print("us", e0.elapsed_time(e1) * 1e3 / 20)
