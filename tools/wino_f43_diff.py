"""Weak #3 of the round-5 review, settled by measurement: what does Winograd F(4x4,3x3) in the DIFFERENTIATED forward pass of the VGG19
feature stack (dehaze_hip.vgg.F43_DIFF; the pass whose roundings decide the ReLU masks of the backward pass) do to the training step?
Two models, same seed, same batches, same sampled keys / DropPath draws: N AdamW steps with the differentiated forward on F(2x2) and on
F(4x4); per step the total loss and the contrastive term of both; before the first step the image gradient `da` of the contrastive term on a
fixed batch at the (identical) initial weights; after the last step the distance of the weights.  (Step TIME: tools/step_ms.sh with
DHZ_WINO_F43_DIFF=0 / 1 - this script draws its batches on the host.)

    python tools/wino_f43_diff.py [steps=50]
"""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "research-and-implementation-of-image-dehazing-algorithm-based-on-vision-transformer_amd"), ROOT]
import torch
import My_model_1 as M1, My_CR
from losses import CharbonnierLoss
from dehaze_hip import vgg
from dehaze_hip.train import FlatAdamW, synthetic_batch, train_step

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dev = torch.device("cuda:0")


def run(diff43):
    vgg.F43_DIFF = diff43
    torch.manual_seed(1234)
    model = M1.Uformer(img_size=128, embed_dim=32, win_size=8, token_projection='linear', token_mlp='leff').to(dev).train()
    opt = FlatAdamW(model, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.02)
    opt.zero_grad()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        cr = My_CR.ContrastLoss(ablation=False).to(dev)
    char = CharbonnierLoss()
    # image gradient of the contrastive term at the INITIAL weights on a fixed batch: identical inputs for both settings, so every
    # difference is the rounding of the differentiated forward pass (values and ReLU masks)
    target, input_ = synthetic_batch(32, 128, seed=7, device=dev)
    with torch.no_grad():
        a = model(input_).clamp(0, 1)
    a = a.detach().requires_grad_(True)
    l0 = cr(a, target, input_)[0]
    l0.backward()
    da0, l0 = a.grad.detach().clone(), l0.item()
    torch.manual_seed(4321)
    out = []
    t0 = None
    for i in range(steps):
        target, input_ = synthetic_batch(32, 128, seed=100 + i, device=dev)
        if i == 5:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        loss, lrec, lcr = train_step(model, char, cr, opt, None, input_, target, 1.0, 1.0)
        out.append((loss.item(), lcr.item()))
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / (steps - 5)
    return out, ms, (da0, l0), {k: v.detach().clone() for k, v in model.state_dict().items()}


a, ms_a, da_a, sd_a = run(False)
b, ms_b, da_b, sd_b = run(True)
print(f"# differentiated VGG forward: F(2x2) {ms_a:.3f} ms/step, F(4x4) {ms_b:.3f} ms/step over {steps - 5} steps")
print("# step   loss F22      loss F43      |d|        cr F22        cr F43       |d|")
for i, ((l0, c0), (l1, c1)) in enumerate(zip(a, b)):
    if i < 5 or i % 5 == 4:
        print(f"{i:5d}  {l0:.7f}  {l1:.7f}  {abs(l0 - l1):.2e}   {c0:.7f}  {c1:.7f}  {abs(c0 - c1):.2e}")
(da_a, l_a), (da_b, l_b) = da_a, da_b
d = (da_a - da_b).abs()
print(f"# contrastive term at the initial weights, fixed batch: loss {l_a:.7f} / {l_b:.7f}; d(loss)/d(restored): max |d| / max |da| = "
      f"{d.max().item() / da_a.abs().max().item():.3e}, mean |d| / mean |da| = {d.mean().item() / da_a.abs().mean().item():.3e}, "
      f"elements that differ by more than 1e-3 of max |da|: {(d > 1e-3 * da_a.abs().max()).float().mean().item():.3e}")
num = sum((sd_a[k].double() - sd_b[k].double()).pow(2).sum().item() for k in sd_a if sd_a[k].dtype.is_floating_point)
den = sum(sd_a[k].double().pow(2).sum().item() for k in sd_a if sd_a[k].dtype.is_floating_point)
print(f"# weights after {steps} steps: relative L2 distance {(num / den) ** 0.5:.3e}")
