"""In-kernel shader clock of linear_gemm_kernel under its own load (MI355X_MICROARCH.md, DVFS give-back item 6): every workgroup
records d(s_memtime) and d(s_memrealtime) around its whole life in a -DDHZ_GEMM_CLOCK build of its own; after >= 1 s of back-to-back
launches the median quotient x 100 MHz is the clock the chip holds on this kernel.    python tools/micro/clock_gemm.py T K N [fwd|dgrad]"""
import ctypes, os, subprocess, sys, time, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
C = os.path.join(R, "research-and-implementation-of-image-dehazing-algorithm-based-on-vision-transformer_amd", "csrc")
D = os.path.join(R, "gpurun_out", "diag"); os.makedirs(D, exist_ok=True)
so = os.path.join(D, "libgemm_clock.so")
subprocess.run(f"/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DDHZ_GEMM_CLOCK -I{R}/include -I{C} -I{C}/build "
               f"{C}/linear_gemm.hip {C}/api.hip -o {so}", shell=True, check=True)
lib = ctypes.CDLL(so)
dev = torch.device("cuda:0"); s = torch.cuda.current_stream().cuda_stream; V = ctypes.c_void_p
shapes = [tuple(int(x) for x in sys.argv[1:4])] if len(sys.argv) > 3 else [(8192, 512, 2048), (131072, 128, 512), (524288, 64, 256)]
mode = sys.argv[4] if len(sys.argv) > 4 else "fwd"
ck = torch.zeros(8 * 512, dtype=torch.int64, device=dev)
assert lib.dhz_debug_clock(V(ck.data_ptr())) == 0
for T, K, N in shapes:
    x = torch.randn(T, K if mode == "fwd" else N, device=dev); W = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
    y = torch.empty(T, N if mode == "fwd" else K, device=dev)
    def run():
        if mode == "fwd":
            lib.dhz_linear_fwd(V(x.data_ptr()), K, V(W.data_ptr()), V(b.data_ptr()), V(y.data_ptr()), N, T, N, K, V(s))
        else:
            lib.dhz_linear_dgrad(V(x.data_ptr()), N, V(W.data_ptr()), V(y.data_ptr()), K, T, N, K, V(s))
    t0 = time.time(); n = 0
    while time.time() - t0 < 1.5:
        for _ in range(50): run()
        torch.cuda.synchronize(); n += 50
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    a = ck.cpu().view(512, 8).double()
    a = a[a[:, 1] > 0]
    ghz = (a[:, 0] / a[:, 1] * 0.1).median().item()
    cyc = a[:, 0].median().item()
    q = torch.quantile(a[:, 0], torch.tensor([0.0, 0.1, 0.5, 0.9, 1.0], dtype=torch.float64)).tolist()
    print(f"   wave 0 of a workgroup, medians: DMA/LDS wait {a[:, 2].median().item():.0f}, barrier {a[:, 3].median().item():.0f}, epilogue issue {a[:, 4].median().item():.0f} cycles")
    print("   workgroup life cycles min / p10 / median / p90 / max:", " ".join(f"{v:.0f}" for v in q), f"= {q[4] / ghz / 1e3:.1f} us for the longest")
    tf = 2 * T * K * N / us / 1e6
    print(f"T={T} K={K} N={N} {mode}: {us:.1f} us, {tf:.1f} TF; in-kernel clock {ghz:.3f} GHz; workgroup life {cyc:.0f} cycles; "
          f"fp32 MFMA peak at this clock {157.3 * ghz / 2.4:.1f} TF -> {tf / (157.3 * ghz / 2.4):.3f} of it")
