"""In-kernel s_memtime stamps of linear_gemm_kernel (phase timeline of ONE workgroup, per wave and stage).  Builds its own copy of
the library with -DDHZ_GEMM_STAMP=<workgroup id> (the product library carries no stamps):
    python tools/micro/stamp_gemm.py T K N [fwd|dgrad] [wg]
columns per stage: gl = issue of the next stage's global loads, mm = LDS reads + MFMAs (issue), sw = vmcnt waits + LDS writes,
bar = barrier; after a tile: ep = epilogue stores (issue)."""
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
st = torch.zeros(4 * 40 * 8, dtype=torch.int64, device=dev)
assert lib.dhz_debug_stamp(V(st.data_ptr())) == 0
for _ in range(3):
    if mode == "fwd":
        lib.dhz_linear_fwd(V(x.data_ptr()), K, V(W.data_ptr()), V(b.data_ptr()), V(y.data_ptr()), N, T, N, K, V(s))
    else:
        lib.dhz_linear_dgrad(V(x.data_ptr()), N, V(W.data_ptr()), V(y.data_ptr()), K, T, N, K, V(s))
torch.cuda.synchronize()
a = st.cpu().view(4, 40, 8)
t0 = int(a[:, 0, 0].min())
for w in range(4):
    out = []
    for i in range(40):
        r = a[w, i]
        if r[0] == 0 and r[5] == 0:
            break
        if r[0]:
            out.append(f"[{int(r[0]) - t0:6d}] gl {int(r[1] - r[0]):4d} mm {int(r[2] - r[1]):5d} sw {int(r[3] - r[2]) if r[3] else 0:5d} bar {int(r[4] - (r[3] if r[3] else r[2])):5d}")
        else:
            out.append(f"   ep {int(r[6] - r[5]):5d}")
    print(f"wave {w}: " + " | ".join(out))
