"""Which torch-side copies / elementwise ops run in a training step (torch.profiler, shapes + python stack)."""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "research-and-implementation-of-image-dehazing-algorithm-based-on-vision-transformer_amd"))
import torch
from torch.profiler import profile, ProfilerActivity
import My_model_1 as M1, My_CR
from losses import CharbonnierLoss
from dehaze_hip.train import FlatAdamW, synthetic_batch, train_step
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = M1.Uformer(img_size=128, embed_dim=32, win_size=8, token_projection='linear', token_mlp='leff').to(dev).train()
opt = FlatAdamW(model, lr=2e-4); opt.zero_grad()
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    cr = My_CR.ContrastLoss().to(dev)
tgt, inp = synthetic_batch(32, 128, device=dev)
for _ in range(3):
    train_step(model, CharbonnierLoss(), cr, opt, None, inp, tgt)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    train_step(model, CharbonnierLoss(), cr, opt, None, inp, tgt)
    torch.cuda.synchronize()
want = ("aten::copy_", "aten::add", "aten::add_", "aten::cat", "aten::sum", "aten::mul", "aten::clone", "aten::contiguous", "aten::fill_", "aten::zero_")
rows = []
for e in prof.events():
    if e.name in want and e.device_time_total > 4:
        stack = [s for s in (e.stack or []) if "dehaze" in s or "My_" in s or "losses" in s]
        rows.append((e.device_time_total, e.name, str(e.input_shapes)[:60], stack[0][-70:] if stack else ""))
import collections, sys
if len(sys.argv) > 1 and sys.argv[1] == "--small":       # group the small launches: (op, shapes, first model-side frame) -> count, total us
    agg = collections.defaultdict(lambda: [0, 0.0])
    for e in prof.events():
        if e.device_time_total > 0 and e.device_time_total < 14 and e.name.startswith("aten::") and e.name not in ("aten::clone", "aten::contiguous", "aten::zero_", "aten::zeros", "aten::zeros_like", "aten::to", "aten::_to_copy", "aten::reshape", "aten::ones", "aten::ones_like", "aten::full", "aten::sub", "aten::rsub"):
            stack = [s for s in (e.stack or []) if "dehaze" in s or "My_" in s or "losses" in s or "train.py" in s]
            k = (e.name, str(e.input_shapes)[:50], stack[0][-60:] if stack else "")
            agg[k][0] += 1; agg[k][1] += e.device_time_total
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
        print(f"{t:7.1f} us {c:4d}x  {k[0]:16s} {k[1]:50s} {k[2]}")
    print("total", sum(v[1] for v in agg.values()), "us in", sum(v[0] for v in agg.values()), "launches")
    sys.exit(0)
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"{len(rows)} ops, {tot/1e3:.2f} ms")
for r in rows[:45]:
    print(f"{r[0]:8.1f} us  {r[1]:18s} {r[2]:60s} {r[3]}")
