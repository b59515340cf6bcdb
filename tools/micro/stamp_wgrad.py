"""Reader for in-kernel s_memtime stamps (phase timeline of one workgroup).  The stamped library is an ad-hoc build: a
copy of csrc/linear_wgrad.hip with `STAMP(slot)` stores of __builtin_amdgcn_s_memtime() at the phase boundaries into a device
buffer installed through an extra `dhz_debug_stamp(ptr)` export, compiled next to this script (hipcc -shared, with
csrc/api.hip).  Product builds carry no stamps.  Findings are recorded in DESIGN.md section 4."""
import ctypes, os, sys, torch
HERE = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(HERE, "liblw_stamp.so"))
dev = torch.device("cuda:0"); s = torch.cuda.current_stream().cuda_stream; V = ctypes.c_void_p
T, N, K = (int(x) for x in sys.argv[1:4])
dy = torch.randn(T, N, device=dev); x = torch.randn(T, K, device=dev); dw = torch.zeros(N, K, device=dev); db = torch.zeros(N, device=dev)
st = torch.zeros(4 * 16 * 8, dtype=torch.int64, device=dev)
lib.dhz_debug_stamp(V(st.data_ptr()))
for _ in range(3):
    lib.dhz_linear_wgrad(V(dy.data_ptr()), N, V(x.data_ptr()), K, T, N, K, V(dw.data_ptr()), V(db.data_ptr()), V(s))
torch.cuda.synchronize()
a = st.cpu().view(4, 16, 8)
for w in range(4):
    print(f"wave {w}: " + "  ".join(f"gl {a[w,i,1]-a[w,i,0]:4d} mm {a[w,i,2]-a[w,i,1]:5d} sw {a[w,i,3]-a[w,i,2]:5d} bar {a[w,i,4]-a[w,i,3]:4d}" for i in range(3, 8)))
print("per stage:", float((a[:, 9, 0] - a[:, 3, 0]).float().mean()) / 6)
