"""Generates the stamped ad-hoc build of csrc/fused_attn.hip that tools/micro/stamp_fused.py reads: STAMP(slot) stores of
s_memtime at the phase boundaries of workgroup 0 (first 8 windows of every wave).  Output: tools/micro/libfa_stamp.so."""
import glob, os, re, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(os.path.dirname(HERE))
csrc = glob.glob(os.path.join(ROOT, "research-*", "csrc"))[0]
s = open(os.path.join(csrc, "fused_attn.hip")).read()
s = s.replace("namespace {\n", "__device__ long long* g_stamp;\nnamespace {\n", 1)
s = s.replace("#pragma unroll 1\n    for (; win < nwin; win += gridDim.x) {",
              "int wcount = 0;\n#define STAMP(k) do { if (blockIdx.x == 0 && lane == 0 && wcount < 8 && g_stamp) g_stamp[(w * 8 + wcount) * 16 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)\n"
              "#pragma unroll 1\n    for (; win < nwin; win += gridDim.x, ++wcount) {\n        STAMP(0);", 1)
def after(marker, code, nth=0):
    global s
    i = -1
    for _ in range(nth + 1):
        i = s.index(marker, i + 1)
    j = i + len(marker)
    s = s[:j] + code + s[j:]
def before(marker, code):
    global s
    i = s.index(marker)
    s = s[:i] + code + s[i:]
before("        // no barrier: wave w normalised exactly the 16 rows", "        STAMP(1);\n")
before("            __syncthreads();\n            if (SAVE) {   // packed QKV rows", "            STAMP(2);\n")
s = s.replace("            __syncthreads();\n            if (SAVE) {   // packed QKV rows", "            __syncthreads();\n            STAMP(15);\n            if (SAVE) {   // packed QKV rows", 1)
before("            // ---- 2a. S = Q_h K_h^T", "            STAMP(3);\n")
before("            // ---- 2b. sparsity measure", "            STAMP(4);\n")
s = s.replace("            __syncthreads();                       // S and M complete", "            STAMP(5);\n            __syncthreads();                       // S and M complete\n            STAMP(6);", 1)
before("            // ---- 2d. P = softmax", "            STAMP(7);\n")
s = s.replace("            __syncthreads();                       // P complete", "            STAMP(8);\n            __syncthreads();                       // P complete\n            STAMP(9);", 1)
s = s.replace("            __syncthreads();                       // O complete", "            STAMP(10);\n            __syncthreads();                       // O complete\n            STAMP(11);", 1)
s = s.replace("            __syncthreads();                       // O (in the Q tile), K, V, S, P are rewritten", "            STAMP(12);\n            __syncthreads(); STAMP(13);            // O (in the Q tile), K, V, S, P are rewritten", 1)
before("        // no barrier: the staging tile is rewritten by this same wave's LayerNorm of the next window", "        STAMP(14);\n")
s += '\nextern "C" int dhz_debug_stamp(long long* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_stamp), &p, sizeof(p)); }\n'
out = os.path.join(HERE, "fa_stamp.hip")
open(out, "w").write(s)
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", f"-I{ROOT}/include", f"-I{csrc}",
       out, os.path.join(csrc, "api.hip"), "-o", os.path.join(HERE, "libfa_stamp.so")]
print(" ".join(cmd)); sys.exit(subprocess.call(cmd))
