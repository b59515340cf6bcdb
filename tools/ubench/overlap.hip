// Microbenchmark: which instruction classes of ONE wave execute concurrently with matrix instructions of ANOTHER wave of the same SIMD?
// 512-thread workgroups, one per CU: waves 0..3 ("M", one per SIMD) issue 8 independent MFMAs per iteration, waves 4..7 ("V", their SIMD
// partners) issue 8 independent instructions of the class under test.  Reported: M alone, V alone, both - "both == M + V" means the class is
// serialised against the matrix pipe, "both == max(M, V)" means it runs beside it.
//     hipcc --offload-arch=gfx950 -O3 -Wno-unused-value tools/ubench/overlap.hip -o tools/ubench/overlap && tools/ubench/overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

#define OPS(X)                                                                                            \
    X(0, "v_pk_fma_f32", "v_pk_fma_f32 %0, %0, %1, %2", "+v"(x2[i]), "v"(c2), "v"(d2))                   \
    X(1, "v_fma_f32", "v_fma_f32 %0, %0, %1, %2", "+v"(x[i]), "v"(c), "v"(d))                            \
    X(2, "v_add_f32", "v_add_f32 %0, %0, %1", "+v"(x[i]), "v"(d), "v"(d))                                \
    X(3, "v_mul_f32", "v_mul_f32 %0, %0, %1", "+v"(x[i]), "v"(c), "v"(d))                                \
    X(4, "v_exp_f32", "v_exp_f32 %0, %0", "+v"(x[i]), "v"(c), "v"(d))                                    \
    X(5, "v_cvt_pk_bf16_f32", "v_cvt_pk_bf16_f32 %0, %0, %1", "+v"(x[i]), "v"(c), "v"(d))                \
    X(6, "v_mad_u32_u24", "v_mad_u32_u24 %0, %0, %1, %2", "+v"(x[i]), "v"(c), "v"(d))                    \
    X(7, "v_add_u32", "v_add_u32 %0, %0, %1", "+v"(x[i]), "v"(c), "v"(d))                                \
    X(8, "v_and_b32", "v_and_b32 %0, %0, %1", "+v"(x[i]), "v"(c), "v"(d))                                \
    X(9, "v_lshlrev_b32", "v_lshlrev_b32 %0, 1, %0", "+v"(x[i]), "v"(c), "v"(d))                         \
    X(10, "v_perm_b32", "v_perm_b32 %0, %0, %1, %2", "+v"(x[i]), "v"(c), "v"(d))                         \
    X(11, "v_and_or_b32", "v_and_or_b32 %0, %0, %1, %2", "+v"(x[i]), "v"(c), "v"(d))                     \
    X(12, "v_mov_b32", "v_mov_b32 %0, %1", "+v"(x[i]), "v"(c), "v"(d))                                   \
    X(13, "v_cndmask_b32", "v_cndmask_b32 %0, %0, %1, vcc", "+v"(x[i]), "v"(c), "v"(d))                  \
    X(14, "v_max_f32", "v_max_f32 %0, %0, %1", "+v"(x[i]), "v"(c), "v"(d))                               \
    X(15, "v_cmp_lt_f32", "v_cmp_lt_f32 vcc, %0, %1", "+v"(x[i]), "v"(c), "v"(d))                        \
    X(16, "v_sub_f32", "v_sub_f32 %0, %0, %1", "+v"(x[i]), "v"(c), "v"(d))                               \
    X(17, "v_mov_b32 dpp", "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf", "+v"(x[i]), "v"(c), "v"(d)) \
    X(18, "v_sub_u32+and (2 ops)", "v_sub_u32 %0, %0, %1\n v_and_b32 %0, %0, %2", "+v"(x[i]), "v"(c), "v"(d))

template <int KIND, int OP>   // KIND 0: v_mfma_f32_16x16x4_f32, 1: v_mfma_f32_16x16x32_bf16
__global__ __launch_bounds__(512, 1) void k(float* out, int iters, int mode) {
    const int w = threadIdx.x >> 6;
    const bool isM = (mode & 8) ? w >= 4 : w < 4;                  // bit 3: the matrix waves are the younger ones
    if (mode & 4) { if (isM) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(3); }   // bit 2: vector waves at priority 3
    if (mode & 16) { if (isM) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0); }  // bit 4: matrix waves at priority 3
    if (isM) {
        if (!(mode & 1)) return;
        f32x4 acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
        float a = threadIdx.x * 1e-3f, b = 1.f;
        bf16x8 ab = {1, 2, 3, 4, 5, 6, 7, 8};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
                else acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, ab, acc[i], 0, 0, 0);
            }
        }
        float s = 0;
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        out[blockIdx.x * 512 + threadIdx.x] = s;
    } else {
        if (!(mode & 2)) return;
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        float x[8];
        f32x2 x2[8];
        for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 1e-3f + i; x2[i] = f32x2{x[i], 1.f}; }
        const float c = 1.0001f, d = 1e-3f;
        const f32x2 c2 = {c, c}, d2 = {d, d};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
#define X(ID, NAME, ASM, O, I1, I2) if (OP == ID) asm volatile(ASM : O : I1, I2 : "vcc");
                OPS(X)
#undef X
            }
        }
        float s = 0;
        for (int i = 0; i < 8; ++i) s += x[i] + x2[i][0];
        out[blockIdx.x * 512 + threadIdx.x] = s;
    }
}

template <int KIND, int OP>
float run(float* out, int iters, int mode) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND, OP>), dim3(256), dim3(512), 0, 0, out, iters, mode);
    hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<KIND, OP>), dim3(256), dim3(512), 0, 0, out, iters, mode);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 3 * 1000;
}

template <int OP>
void row(float* out, const char* name, int iters) {
    const float m0 = run<0, OP>(out, iters, 1), v0 = run<0, OP>(out, iters, 2), b0 = run<0, OP>(out, iters, 3);
    const float m1 = run<1, OP>(out, iters, 1), v1 = run<1, OP>(out, iters, 2), b1 = run<1, OP>(out, iters, 3);
    auto verdict = [](float m, float v, float b) { const float f = (b - (m > v ? m : v)) / (m < v ? m : v); return f > 0.75f ? "ADD" : f < 0.25f ? "overlap" : "partial"; };
    printf("%-22s | fp32 16x16x4: M %7.1f  V %7.1f  both %7.1f  %-7s | bf16 16x16x32: M %7.1f  V %7.1f  both %7.1f  %-7s\n", name, m0, v0, b0,
           verdict(m0, v0, b0), m1, v1, b1, verdict(m1, v1, b1));
}

template <int OP>
void prio(float* out, const char* name, int iters) {
    for (int kind = 0; kind < 2; ++kind) {
        printf("%-12s vs %s  M %7.1f V %7.1f | both: equal priority %7.1f, V at prio 3 %7.1f, M at prio 3 %7.1f, M younger %7.1f, M younger + V prio 3 %7.1f\n", name,
               kind ? "bf16 16x16x32" : "fp32 16x16x4 ", kind ? run<1, OP>(out, iters, 1) : run<0, OP>(out, iters, 1), kind ? run<1, OP>(out, iters, 2) : run<0, OP>(out, iters, 2),
               kind ? run<1, OP>(out, iters, 3) : run<0, OP>(out, iters, 3), kind ? run<1, OP>(out, iters, 3 | 4) : run<0, OP>(out, iters, 3 | 4),
               kind ? run<1, OP>(out, iters, 3 | 16) : run<0, OP>(out, iters, 3 | 16), kind ? run<1, OP>(out, iters, 3 | 8) : run<0, OP>(out, iters, 3 | 8),
               kind ? run<1, OP>(out, iters, 3 | 8 | 4) : run<0, OP>(out, iters, 3 | 8 | 4));
    }
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 512 * 4);
    const int iters = 20000;
    prio<1>(out, "v_fma_f32", iters);
    prio<8>(out, "v_and_b32", iters);
    prio<0>(out, "v_pk_fma_f32", iters);
    printf("us per launch; %d iterations x 8 instructions per wave; M = matrix waves alone, V = the other four waves alone (one per SIMD)\n", iters);
#define X(ID, NAME, ASM, O, I1, I2) row<ID>(out, NAME, iters);
    OPS(X)
#undef X
    return 0;
}
