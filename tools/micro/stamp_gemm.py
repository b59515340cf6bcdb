"""In-kernel s_memtime stamps of linear_gemm_kernel (phase timeline of ONE workgroup, per wave and loop iteration).  Builds its own
copy of the library with -DDHZ_GEMM_STAMP=<workgroup id> (the product library carries no stamps; stamps are kept in LDS and dumped
at the end of the kernel):
    python tools/micro/stamp_gemm.py T K N [fwd|dgrad] [wg]
per iteration (loop rotated by half a stage): wt = counted vmcnt wait, bar = barrier, s1 = second half of the stage (64 MFMAs at
WM = WN = 4) with the next-but-one stage's DMA and the next stage's fragment reads inside, ep = epilogue (stores at a tile end),
s0 = first half of the next stage."""
import ctypes, os, subprocess, sys, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
P = os.path.join(R, "research-and-implementation-of-image-dehazing-algorithm-based-on-vision-transformer_amd")
C = os.path.join(P, "csrc")
T, K, N = (int(x) for x in sys.argv[1:4])
mode = sys.argv[4] if len(sys.argv) > 4 else "fwd"
wg = sys.argv[5] if len(sys.argv) > 5 else "100"
D = os.path.join(R, "gpurun_out", "diag"); os.makedirs(D, exist_ok=True)
so = os.path.join(D, "libgemm_stamp.so")
subprocess.run(f"/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DDHZ_GEMM_STAMP={wg} -I{R}/include -I{C} -I{C}/build "
               f"{C}/linear_gemm.hip {C}/api.hip -o {so}", shell=True, check=True)
lib = ctypes.CDLL(so)
dev = torch.device("cuda:0"); s = torch.cuda.current_stream().cuda_stream; V = ctypes.c_void_p
x = torch.randn(T, K if mode == "fwd" else N, device=dev); W = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
y = torch.empty(T, N if mode == "fwd" else K, device=dev)
NS = 24
st = torch.zeros(4 * NS * 8, dtype=torch.int64, device=dev)
assert lib.dhz_debug_stamp(V(st.data_ptr())) == 0
for _ in range(3):
    if mode == "fwd":
        lib.dhz_linear_fwd(V(x.data_ptr()), K, V(W.data_ptr()), V(b.data_ptr()), V(y.data_ptr()), N, T, N, K, V(s))
    else:
        lib.dhz_linear_dgrad(V(x.data_ptr()), N, V(W.data_ptr()), V(y.data_ptr()), K, T, N, K, V(s))
torch.cuda.synchronize()
a = st.cpu().view(4, NS, 8)
t0 = int(a[:, 0, 0].min())
d = lambda x, y: (int(x) - int(y)) & 0xffffffff
for w in range(4):
    out = []
    for i in range(NS):
        r = a[w, i]
        if r[0] == 0:
            break
        out.append(f"[{d(r[0], t0):6d}] wt {d(r[1], r[0]) if r[1] else 0:4d} bar {d(r[2], r[1]) if r[1] else 0:4d} s1 {d(r[3], r[2]):5d} ep {d(r[4], r[3]):4d} s0 {d(r[5], r[4]) if r[5] else 0:5d}")
    print(f"wave {w}: " + " | ".join(out))
