"""Host (CPU) time of one training step against its device time: after a device sync, the wall time of N step() calls WITHOUT
synchronisation is the time the host needs to enqueue them (as long as the stream's queue does not fill up); the device time follows
from the sync at the end.  host << device means the step is GPU-bound."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "research-and-implementation-of-image-dehazing-algorithm-based-on-vision-transformer_amd"), ROOT]
import torch, warnings
import My_model_1 as M1, My_CR
from losses import CharbonnierLoss
from dehaze_hip.train import FlatAdamW, synthetic_batch, train_step
dev = torch.device("cuda:0")
torch.manual_seed(1234)
model = M1.Uformer(img_size=128, embed_dim=32, win_size=8, token_projection='linear', token_mlp='leff').to(dev).train()
opt = FlatAdamW(model, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.02); opt.zero_grad()
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    cr = My_CR.ContrastLoss(ablation=False).to(dev)
char = CharbonnierLoss()
target, input_ = synthetic_batch(32, 128, seed=1234, device=dev)
step = lambda: train_step(model, char, cr, opt, None, input_, target, 1.0, 1.0)
for _ in range(6):
    step()
torch.cuda.synchronize()
for n in (1, 2, 4):
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{n} step(s): host enqueue {1e3 * (t1 - t0) / n:.2f} ms/step, until the device is done {1e3 * (t2 - t0) / n:.2f} ms/step")
