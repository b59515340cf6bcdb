"""Forward token Linear as TN (weight [N,K], what nn.Linear stores) vs NN (pre-transposed [K,N] copy) library GEMM."""
import torch
dev = torch.device("cuda:0")
print(f"{'T':>7} {'K':>5} {'N':>5} | TN us  | NN us  | TN no bias")
tot = [0, 0, 0]
for T, C in [(524288, 32), (131072, 64), (32768, 128), (8192, 256), (2048, 512), (8192, 512), (32768, 256), (131072, 128), (524288, 64)]:
    for K, N in [(C, 3 * C), (C, C), (C, 4 * C), (4 * C, C)]:
        x = torch.randn(T, K, device=dev); W = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
        Wt = W.t().contiguous()
        res = []
        for f in (lambda: torch.addmm(b, x, W.t()), lambda: torch.addmm(b, x, Wt), lambda: x @ W.t()):
            for _ in range(3): f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): f()
            e1.record(); torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) / 10 * 1e3)
        for i in range(3): tot[i] += res[i]
        print(f"{T:7d} {K:5d} {N:5d} | {res[0]:7.1f} | {res[1]:7.1f} | {res[2]:7.1f}")
print("sum", tot)
