"""Phase timing of the Winograd kernel from in-kernel s_memtime stamps (profiling build tools/micro/libwino_stamp.so,
built with -DWINO_STAMP).  Prints, per wave of workgroup 300 and per channel-group iteration, the cycles of step 0, step 1
and the barrier wait."""
import ctypes, os, sys
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(HERE, "libwino_stamp.so"))
dev = torch.device("cuda:0"); s = torch.cuda.current_stream().cuda_stream
B, C, K, H = 64, int(os.environ.get("C", 512)), int(os.environ.get("K", 512)), int(os.environ.get("H", 16))
V = ctypes.c_void_p
x = torch.randn(B, C // 8, H, H, 8, device=dev); y = torch.empty(B, K // 8, H, H, 8, device=dev)
w = torch.randn(K, C, 3, 3, device=dev) * 0.05; b = torch.randn(K, device=dev); up = torch.empty(16 * K * C, device=dev)
lib.dhz_winograd_prepack(V(w.data_ptr()), V(up.data_ptr()), K, C, 0, V(s))
stamp = torch.zeros(8 * 16 * 8, dtype=torch.int64, device=dev)
lib.dhz_debug_wino_stamp(V(stamp.data_ptr()))
for _ in range(3):
    rc = lib.dhz_winograd_conv3x3(V(x.data_ptr()), None, V(up.data_ptr()), V(b.data_ptr()), V(y.data_ptr()), B, H, H, C, K, 1, V(s))
torch.cuda.synchronize()
st = stamp.cpu().view(8, 16, 8)
t0 = st[:, 0, 0].min()
print("rc", rc, " (cycles; s_memtime ticks)")
for wv in range(8):
    row = []
    for it in range(2, 6):
        a, b1, c1, d = (st[wv, it, j].item() for j in range(4))
        t4, t5, t6, t7 = (st[wv, it, j].item() for j in range(4, 8))
        ts = a if wv >= 4 else b1          # transform phase start: iteration start (half 1) / end of step 0 (half 0)
        row.append(f"{b1-a:5d}/{c1-b1:5d}/{d-c1:4d} [T: wp {t4-ts:4d} gl {t5-t4:4d} p0 {t6-t5:4d} p1 {t7-t6:4d}]")
    print(f"wave {wv} start {st[wv,2,0].item()-t0:6d} | step0/step1/barrier: " + "  ".join(row))
per_iter = (st[:, 9, 0] - st[:, 2, 0]).float().mean().item() / 7
print("mean cycles per iteration:", per_iter)
