"""Reader for in-kernel s_memtime stamps (phase timeline of one workgroup).  The stamped library is an ad-hoc build: a
copy of csrc/fused_attn.hip with `STAMP(slot)` stores of __builtin_amdgcn_s_memtime() at the phase boundaries into a device
buffer installed through an extra `dhz_debug_stamp(ptr)` export, compiled next to this script (hipcc -shared, with
csrc/api.hip).  Product builds carry no stamps.  Findings are recorded in DESIGN.md section 4."""
import ctypes, os, sys, torch
HERE = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(HERE, "libfa_stamp.so"))
dev = torch.device("cuda:0"); s = torch.cuda.current_stream().cuda_stream; V = ctypes.c_void_p
C = int(sys.argv[1]) if len(sys.argv) > 1 else 32; res = int(sys.argv[2]) if len(sys.argv) > 2 else 128; B = 32; H = C // 32
T = B * res * res
x = torch.randn(T, C, device=dev); gamma = torch.rand(C, device=dev) + .5; beta = torch.randn(C, device=dev) * .1
wq, wk, wv, wo = (torch.randn(C, C, device=dev) * C ** -0.5 for _ in range(4)); bq = torch.zeros(3 * C, device=dev); bo = torch.zeros(C, device=dev)
wqkv_p = torch.empty(3 * C * C, device=dev); wo_p = torch.empty(C * C, device=dev)
lib.dhz_fused_attn_prepack(V(wq.data_ptr()), V(wk.data_ptr()), V(wv.data_ptr()), V(wo.data_ptr()), V(wqkv_p.data_ptr()), V(wo_p.data_ptr()), C, V(s))
idx = torch.randint(64, (64, 25), dtype=torch.uint8).to(dev); bias = torch.randn(H, 64, 64, device=dev) * .1
out = torch.empty_like(x); xn = torch.empty_like(x); qkv = torch.empty(T, 3 * C, device=dev); ctx = torch.empty_like(x)
stats = torch.empty(T, 2, device=dev); rank = torch.empty(T // 64 * H * 64, dtype=torch.uint8, device=dev)
st = torch.zeros(4 * 8 * 16, dtype=torch.int64, device=dev)
lib.dhz_debug_stamp(V(st.data_ptr()))
for _ in range(3):
    rc = lib.dhz_fused_window_attn_fwd(V(x.data_ptr()), V(gamma.data_ptr()), V(beta.data_ptr()), V(wqkv_p.data_ptr()), V(bq.data_ptr()), V(wo_p.data_ptr()), V(bo.data_ptr()),
        V(idx.data_ptr()), (None if os.environ.get("NOBIAS") else V(bias.data_ptr())), None, None, V(out.data_ptr()), V(xn.data_ptr()), V(qkv.data_ptr()), V(ctx.data_ptr()), V(stats.data_ptr()), V(rank.data_ptr()), B, res, res, C, 0, V(s))
torch.cuda.synchronize()
a = st.cpu().view(4, 8, 16)
names = ["LN", "QKV", "bar", "save", "S", "M", "bar", "rank", "smax", "bar", "PV", "bar", "proj", "bar", "epi"]
order = [0, 1, 2, 15, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14]
print("rc", rc)
for w in range(4):
    for win in (3, 4):
        t = [a[w, win, o].item() for o in order]
        print(f"wave {w} win {win}: " + " ".join(f"{n}:{t[i+1]-t[i]}" for i, n in enumerate(names)) + f" | total {t[-1]-t[0]}")
