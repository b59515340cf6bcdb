"""ms/step of the default bench.py workload with module-level switches of dehaze_hip.fused changed first - the step-level A/B for
dispatch decisions that were taken on isolated-kernel timings:
    python tools/step_flags.py LEFF_FUSED=False          python tools/step_flags.py "LEFF_FUSED_C=(32,)" ENABLED=False"""
import io, json, os, sys, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "research-and-implementation-of-image-dehazing-algorithm-based-on-vision-transformer_amd"), ROOT]
import torch
torch.zeros(1, device="cuda:0")          # HIP initialised by torch before the kernel library is loaded
from dehaze_hip import fused
flags = [a for a in sys.argv[1:] if "=" in a and not a.startswith("--")]
for f in flags:
    k, v = f.split("=", 1)
    assert hasattr(fused, k), k
    setattr(fused, k, eval(v))
import bench
sys.argv = ["bench.py", "--steps", "30", "--warmup", "8", "--no-cpu-baseline", "--no-kernel-timing"] + [a for a in sys.argv[1:] if a not in flags]
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
d = json.loads([l for l in buf.getvalue().splitlines() if l.startswith("{")][-1])
print(f"{' '.join(flags) or '(defaults)':40s} {d['ms_per_step']:.3f} ms/step")
