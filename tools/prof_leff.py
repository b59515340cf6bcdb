"""Runs the fused LeFF forward + backward a few times on one stage shape (for rocprofv3 --pmc / --kernel-trace)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "research-and-implementation-of-image-dehazing-algorithm-based-on-vision-transformer_amd"), ROOT]
import torch
import My_model_1 as M1
from dehaze_hip import fused
C, res, B = int(sys.argv[1]), int(sys.argv[2]), 32
dev = torch.device("cuda:0")
norm = torch.nn.LayerNorm(C).to(dev)
mlp = M1.LeFF(C, 4 * C).to(dev)
x = torch.randn(B, res * res, C, device=dev, requires_grad=True)
g = torch.randn(B, res * res, C, device=dev)
sc = torch.ones(B, device=dev)
for _ in range(4):
    y = fused.leff_branch(x, norm, mlp, sc, res, res)
    y.backward(g)
    x.grad = None
torch.cuda.synchronize()
