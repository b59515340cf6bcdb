"""Which torch-side elementwise kernels does one training step still launch, from where?  torch.profiler with stacks, one step.
   python tools/micro/find_copies.py [--bf16]"""
import os, sys, warnings
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..",
                                "research-and-implementation-of-image-dehazing-algorithm-based-on-vision-transformer_amd"))
import torch
import My_model_1 as M1, My_CR
from losses import CharbonnierLoss
from dehaze_hip.train import FlatAdamW, synthetic_batch, train_step
from torch.profiler import profile, ProfilerActivity
bf16 = "--bf16" in sys.argv
dev = torch.device("cuda:0")
torch.manual_seed(1234)
E, ps, bs = (64, 256, 8) if bf16 else (32, 128, 32)
model = M1.Uformer(img_size=ps, embed_dim=E, win_size=8, token_projection='linear', token_mlp='leff').to(dev).train()
if bf16:
    model.act_dtype = torch.bfloat16
opt = FlatAdamW(model, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.02)
opt.zero_grad()
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    cr = My_CR.ContrastLoss(ablation=False).to(dev)
target, input_ = synthetic_batch(bs, ps, seed=1234, device=dev)
char = CharbonnierLoss()
for _ in range(4):
    train_step(model, char, cr, opt, None, input_, target, 1.0, 1.0)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    train_step(model, char, cr, opt, None, input_, target, 1.0, 1.0)
    torch.cuda.synchronize()
ka = prof.key_averages(group_by_input_shape=True, group_by_stack_n=12)
rows = []
for k in ka:
    dt = getattr(k, "self_device_time_total", 0) or getattr(k, "self_cuda_time_total", 0)
    if k.key.startswith("aten::") and dt > 0:
        st = [s_ for s_ in (k.stack or []) if ("dehaze" in s_ or "My_" in s_ or "losses" in s_)]
        rows.append((dt, k.count, k.key, str(k.input_shapes)[:70], st[0][-75:] if st else ""))
rows.sort(key=lambda r: -r[0])
for dt, c, n, sh, st in rows[:45]:
    print(f"{dt:8.1f} us {c:3d} x  {n:24s} {sh:70s} {st}")
print("sum of aten self device time: %.1f us" % sum(r[0] for r in rows))
