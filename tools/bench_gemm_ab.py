"""A/B of one DHZ_* switch of the DIAGNOSTIC library on the token-Linear GEMM shapes, interleaved in one process:
   DHZ_LIB_PATH=gpurun_out/diag/libdehaze_hip_diag.so python tools/bench_gemm_ab.py DHZ_GEMM_TILE 4,4 4,2 2,2"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "research-and-implementation-of-image-dehazing-algorithm-based-on-vision-transformer_amd"), ROOT]
import torch
from dehaze_hip import ops
dev = torch.device("cuda:0")
var, vals = sys.argv[1], sys.argv[2:]
only_c = [int(c) for c in os.environ.get("ONLY_C", "").split(",") if c]


def timeit(f, reps=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print(f"{'T':>7} {'K':>5} {'N':>5} | fwd " + " ".join(f"{var}={v:>3}" for v in vals) + " | dgrad " + " ".join(f"{var}={v:>3}" for v in vals))
tot = [[0.0] * len(vals), [0.0] * len(vals)]
for T, C in [(524288, 32), (131072, 64), (32768, 128), (8192, 256), (2048, 512), (8192, 512), (32768, 256), (131072, 128), (524288, 64)]:
    if only_c and C not in only_c:
        continue
    for K, N in [(C, 3 * C), (C, C), (C, 4 * C), (4 * C, C)]:
        x = torch.randn(T, K, device=dev); W = torch.randn(N, K, device=dev) * 0.1; b = torch.randn(N, device=dev)
        dy = torch.randn(T, N, device=dev)
        best = [[1e9] * len(vals), [1e9] * len(vals)]
        for rnd in range(6):
            for i, v in enumerate(vals):
                os.environ[var] = v
                for d, f in enumerate([lambda: ops.gemm_fwd(x, W, b), lambda: ops.gemm_dgrad(dy, W)]):
                    if rnd == 0:
                        f()
                    best[d][i] = min(best[d][i], timeit(f))
        for d in range(2):
            for i in range(len(vals)):
                tot[d][i] += best[d][i]
        print(f"{T:7d} {K:5d} {N:5d} | " + " ".join(f"{t:8.1f}" for t in best[0]) + " | " + " ".join(f"{t:8.1f}" for t in best[1]))
print("sum fwd", [round(t) for t in tot[0]], "dgrad", [round(t) for t in tot[1]])
