"""dhz_linear_fwd / dhz_linear_dgrad against the library GEMM (torch.addmm / @ -> hipBLASLt) on the token-Linear shapes of the
config-2 step (bs = 32), interleaved rounds in one process.  us per call, algorithmic TB/s and TFLOP/s."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "research-and-implementation-of-image-dehazing-algorithm-based-on-vision-transformer_amd"), ROOT]
import torch
from dehaze_hip import ops
dev = torch.device("cuda:0")


def timeit(fs, rounds=5, reps=5):
    best = [1e9] * len(fs)
    for f in fs:
        f()
    torch.cuda.synchronize()
    for _ in range(rounds):
        for i, f in enumerate(fs):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                f()
            e1.record()
            torch.cuda.synchronize()
            best[i] = min(best[i], e0.elapsed_time(e1) / reps * 1e3)
    return best


nolib = "nolib" in sys.argv[1:]            # phase-ablation runs: only this library's kernels
bf16 = "bf16" in sys.argv[1:]              # config-4 shapes (bs = 8, 256x256, E = 64) in bf16 storage
only = [int(a) for a in sys.argv[1:] if a not in ("nolib", "bf16")]
print(f"{'T':>7} {'K':>5} {'N':>5} | fwd: lib us  mine us   TB/s    TF | dgrad: lib us  mine us   TB/s    TF")
tot = [0, 0, 0, 0]
shapes = [(524288, 32), (131072, 64), (32768, 128), (8192, 256), (2048, 512), (8192, 512), (32768, 256), (131072, 128), (524288, 64)]
if bf16:
    shapes = [(524288, 64), (131072, 128), (32768, 256), (8192, 512), (2048, 1024), (8192, 1024), (32768, 512), (131072, 256), (524288, 128)]
dt = torch.bfloat16 if bf16 else torch.float32
esz = 2 if bf16 else 4
for T, C in shapes:
    if only and C not in only:
        continue
    for K, N in [(C, 3 * C), (C, C), (C, 4 * C), (4 * C, C)]:
        x = torch.randn(T, K, device=dev).to(dt); W = torch.randn(N, K, device=dev) * 0.1; b = torch.randn(N, device=dev)
        dy = torch.randn(T, N, device=dev).to(dt)
        Wl, bl = W.to(dt), b.to(dt)                      # the library's operands
        fs = [lambda: torch.addmm(bl, x, Wl.t()), lambda: ops.gemm_fwd(x, W, b), lambda: dy @ Wl, lambda: ops.gemm_dgrad(dy, W)]
        if nolib:
            fs[0] = fs[1]; fs[2] = fs[3]
        r = timeit(fs)
        gb = T * (N + K) * esz / 1e12; tf = 2 * T * N * K / 1e12
        for i in range(4):
            tot[i] += r[i]
        print(f"{T:7d} {K:5d} {N:5d} | {r[0]:12.1f} {r[1]:8.1f} {gb/r[1]*1e6:6.2f} {tf/r[1]*1e6:5.1f} | {r[2]:14.1f} {r[3]:8.1f} {gb/r[3]*1e6:6.2f} {tf/r[3]*1e6:5.1f}")
print(f"sum (one block of every stage): fwd lib {tot[0]:.0f} mine {tot[1]:.0f} us | dgrad lib {tot[2]:.0f} mine {tot[3]:.0f} us")
