"""feasibility: the whole training step under a HIP graph (torch.cuda.graph) - capture, replay, ms per step against eager"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "research-and-implementation-of-image-dehazing-algorithm-based-on-vision-transformer_amd"), ROOT):
    sys.path.insert(0, p)
import warnings, torch
import My_model_1 as M1, My_CR
from losses import CharbonnierLoss
from dehaze_hip.train import FlatAdamW, synthetic_batch, train_step
dev = torch.device("cuda:0")
cfg = sys.argv[1] if len(sys.argv) > 1 else "2"
E, ps, bs, bf = (32, 128, 32, False) if cfg == "2" else (64, 256, 8, True)
torch.manual_seed(1234)
model = M1.Uformer(img_size=ps, embed_dim=E, win_size=8, token_projection='linear', token_mlp='leff').to(dev).train()
if bf: model.act_dtype = torch.bfloat16
opt = FlatAdamW(model, lr=2e-4, weight_decay=0.02); opt.zero_grad()
char = CharbonnierLoss()
with warnings.catch_warnings():
    warnings.simplefilter("ignore"); cr = My_CR.ContrastLoss(ablation=False).to(dev)
target, input_ = synthetic_batch(bs, ps, seed=1234, device=dev)
step = lambda: train_step(model, char, cr, opt, None, input_, target, 1.0, 1.0)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n
for _ in range(5): step()
print("eager  %.2f ms/step" % timeit(step), flush=True)
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(s)
blocks = [b for st in model.stages() for b in st.blocks]
static_idx = torch.randint(64, (len(blocks), 64, 25)).to(torch.uint8).to(dev)
model._stage_sample_indices = lambda device: [setattr(b, "_staged_idx", static_idx[i]) for i, b in enumerate(blocks)]   # static device table
os.environ.setdefault("AMD_LOG_LEVEL", "0")
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        loss, _, _ = step()
    print("captured", flush=True)
    print("replay %.2f ms/step  loss %.5f" % (timeit(g.replay), float(loss)), flush=True)
except Exception as e:
    import traceback
    tb = traceback.format_exc().splitlines()
    print("capture failed:", type(e).__name__, str(e)[:120])
    print("\n".join(l for l in tb if "File" in l or l.startswith("    "))[-3000:])
