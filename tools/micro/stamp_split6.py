"""In-kernel s_memtime stamps of split6_gemm_kernel (phase timeline of ONE workgroup, per wave and loop iteration).  Builds its own
copy of the kernel file with -DDHZ_S6_STAMP=<workgroup id>:   python tools/micro/stamp_split6.py T K N [fwd|dgrad] [wg] [extra -D flags]
per iteration: m0 = DMA issue + upper-row fragment reads + lower-row MFMAs (issue), aw = wait for the raw activations, sp = split +
LDS writes + next loads + next fragment reads (issue), m1 = upper-row MFMAs, ep = epilogue, wt = counted wait, bar = barrier"""
import ctypes, os, subprocess, sys, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
P = os.path.join(R, "research-and-implementation-of-image-dehazing-algorithm-based-on-vision-transformer_amd")
C = os.path.join(P, "csrc")
T, K, N = (int(x) for x in sys.argv[1:4])
mode = sys.argv[4] if len(sys.argv) > 4 else "fwd"
wg = sys.argv[5] if len(sys.argv) > 5 else "100"
extra = " ".join(sys.argv[6:])
D = os.path.join(R, "gpurun_out", "diag"); os.makedirs(D, exist_ok=True)
so = os.path.join(D, "libs6_stamp.so")
subprocess.run(f"/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DDHZ_S6_STAMP={wg} {extra} -I{R}/include -I{C} -I{C}/build "
               f"{C}/split6_gemm.hip {C}/api.hip -o {so}", shell=True, check=True)
lib = ctypes.CDLL(so)
dev = torch.device("cuda:0"); s = torch.cuda.current_stream().cuda_stream; V = ctypes.c_void_p
x = torch.randn(T, K if mode == "fwd" else N, device=dev); W = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
y = torch.empty(T, N if mode == "fwd" else K, device=dev)
pl = [torch.empty(N * K, dtype=torch.bfloat16, device=dev) for _ in range(3)]
lib.dhz_split3_planes(V(W.data_ptr()), ctypes.c_int64(N * K), V(pl[0].data_ptr()), V(pl[1].data_ptr()), V(pl[2].data_ptr()), V(s))
NS = 24
st = torch.zeros(8 * NS * 8, dtype=torch.int64, device=dev)
assert lib.dhz_debug_s6_stamp(V(st.data_ptr())) == 0
for _ in range(3):
    if mode == "fwd":
        lib.dhz_linear_fwd_split6(V(x.data_ptr()), K, V(pl[0].data_ptr()), V(pl[1].data_ptr()), V(pl[2].data_ptr()), V(b.data_ptr()), V(y.data_ptr()), N, T, N, K, V(s))
    else:
        lib.dhz_linear_dgrad_split6(V(x.data_ptr()), N, V(pl[0].data_ptr()), V(pl[1].data_ptr()), V(pl[2].data_ptr()), V(y.data_ptr()), K, T, N, K, V(s))
torch.cuda.synchronize()
a = st.cpu().view(8, NS, 8)
t0 = int(a[:, 0, 0].min())
d = lambda x, y: (int(x) - int(y)) & 0xffffffff
for w in range(8):
    out = []
    for i in range(NS):
        r = a[w, i]
        if r[0] == 0:
            break
        if os.environ.get("DHZ_S6_TILE") == "2":       # wide kernel: frag reads + wait | stream 1 | loads + stream 2 | epilogue + wait | barrier
            out.append(f"[{d(r[0], t0):6d}] fr {d(r[1], r[0]):4d} s1 {d(r[2], r[1]):4d} s2 {d(r[3], r[2]):4d} ep+wt {d(r[4], r[3]):4d} bar {d(r[5], r[4]):4d}")
        else:
            out.append(f"[{d(r[0], t0):6d}] s1 {d(r[1], r[0]):4d} s2 {d(r[2], r[1]):4d} ep+wt {d(r[3], r[2]):4d} bar {d(r[4], r[3]):4d}")
    print(f"wave {w}: " + " | ".join(out[4:14]))
