"""Verdict item 5: concurrency with PARTITIONED grids, measured once, properly.  The no-gradient VGG19 passes of the contrastive loss
(ground truth + hazy input: 2/3 of the forward Winograd work, matrix-bound) on a side stream restricted to N compute units
(hipExtStreamCreateWithCUMask) beside the model's forward on the main stream.  ms per training step (config 2), alternating passes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "research-and-implementation-of-image-dehazing-algorithm-based-on-vision-transformer_amd"), ROOT):
    sys.path.insert(0, p)
import warnings, torch
import My_model_1 as M1, My_CR
from losses import CharbonnierLoss
from dehaze_hip.train import FlatAdamW, SideStream, synthetic_batch, train_step
dev = torch.device("cuda:0")
torch.manual_seed(1234)
model = M1.Uformer(img_size=128, embed_dim=32, win_size=8, token_projection='linear', token_mlp='leff').to(dev).train()
opt = FlatAdamW(model, lr=2e-4, weight_decay=0.02); opt.zero_grad()
char = CharbonnierLoss()
with warnings.catch_warnings():
    warnings.simplefilter("ignore"); cr = My_CR.ContrastLoss(ablation=False).to(dev)
target, input_ = synthetic_batch(32, 128, seed=1234, device=dev)
def run(side, steps=20, warm=5):
    for _ in range(warm): train_step(model, char, cr, opt, None, input_, target, 1.0, 1.0, side=side)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): loss, _, _ = train_step(model, char, cr, opt, None, input_, target, 1.0, 1.0, side=side)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps, float(loss)
variants = [("default stream only", None, None), ("side stream, all CUs; main: default stream", SideStream(dev, 0), None)]
for n in (64, 96, 128):
    try:
        variants.append((f"side on {n} CUs; main: default stream (all CUs)", SideStream(dev, n), None))
        variants.append((f"side on {n} CUs; main on the other {256 - n}", SideStream(dev, n), SideStream(dev, 256 - n, first=n)))
        variants.append((f"no side stream; main on {256 - n} CUs", None, SideStream(dev, 256 - n, first=n)))
    except Exception as e:
        print("CU mask unavailable:", e); break
for rep in range(2):
    for name, side, main in variants:
        if main is not None:
            main.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(main.stream):
                ms, loss = run(side)
        else:
            ms, loss = run(side)
        print(f"pass {rep + 1}: {name:52s} {ms:7.2f} ms/step   loss {loss:.5f}", flush=True)
