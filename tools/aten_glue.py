"""Which torch (ATen) kernels still run inside the config-2 training step, and from where?  One profiled step under
torch.profiler with Python stacks; prints every ATen op that launched a device kernel, grouped by (op, shapes, innermost repo frame)."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "research-and-implementation-of-image-dehazing-algorithm-based-on-vision-transformer_amd"), ROOT]
import torch
from torch.profiler import profile, ProfilerActivity
import My_model_1 as M1, My_CR
from losses import CharbonnierLoss
from dehaze_hip.train import FlatAdamW, synthetic_batch, train_step
dev = torch.device("cuda:0")
torch.manual_seed(1234)
model = M1.Uformer(img_size=128, embed_dim=32, win_size=8, token_projection='linear', token_mlp='leff').to(dev).train()
opt = FlatAdamW(model, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.02); opt.zero_grad()
import warnings
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    cr = My_CR.ContrastLoss(ablation=False).to(dev)
char = CharbonnierLoss()
target, input_ = synthetic_batch(32, 128, seed=1234, device=dev)
step = lambda: train_step(model, char, cr, opt, None, input_, target, 1.0, 1.0)
for _ in range(4):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if ev.device_type != torch.autograd.DeviceType.CPU or not ev.name.startswith("aten::"):
        continue
    kt = sum(k.duration for k in ev.kernels)
    if not ev.kernels:
        continue
    frame = next((s for s in ev.stack if "/repo/" in s and "tools/" not in s), ev.stack[0] if ev.stack else "?")
    frame = frame.split("/repo/")[-1]
    agg[(ev.name, str(ev.input_shapes)[:70], frame[-80:])][0] += len(ev.kernels)
    agg[(ev.name, str(ev.input_shapes)[:70], frame[-80:])][1] += kt
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for _, v in rows); n = sum(v[0] for _, v in rows)
print(f"ATen ops with device kernels in one step: {n} launches, {tot/1e3:.3f} ms")
for (name, shp, fr), (c, t) in rows:
    print(f"{t:9.1f} us {c:4d}x  {name:28s} {shp:70s} {fr}")
