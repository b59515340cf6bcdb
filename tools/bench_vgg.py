"""Micro-benchmark of the contrastive-loss VGG19 passes (MIOpen) in NCHW vs channels_last, bs=32."""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "research-and-implementation-of-image-dehazing-algorithm-based-on-vision-transformer_amd"))
import torch
if os.environ.get("CUDNN_BENCH"): torch.backends.cudnn.benchmark = True
warnings.simplefilter("ignore")
import My_CR
dev = torch.device("cuda:0")
cl = My_CR.ContrastLoss().to(dev)
a0 = torch.rand(32, 3, 128, 128, device=dev); p = torch.rand_like(a0); n = torch.rand_like(a0)
for fmt in ("nchw", "channels_last"):
    if fmt == "channels_last":
        cl = cl.to(memory_format=torch.channels_last)
    def run():
        a = a0.clone().requires_grad_()
        aa, pp, nn_ = (a, p, n) if fmt == "nchw" else (a.contiguous(memory_format=torch.channels_last), p.contiguous(memory_format=torch.channels_last), n.contiguous(memory_format=torch.channels_last))
        loss, _, _ = cl(aa, pp, nn_)
        loss.backward()
    for _ in range(3): run()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): run()
    torch.cuda.synchronize(); print(fmt, (time.perf_counter() - t) / 10 * 1e3, "ms per CR fwd+bwd (bs 32)")
