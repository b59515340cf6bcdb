// Does fp32 VALU work overlap with an fp32-input MFMA stream on gfx950?  (tools/micro, not part of the product)
//   mode 0: 8 waves/CU-block, all MFMA        mode 1: all VALU
//   mode 2: waves 0-3 MFMA, waves 4-7 VALU (SIMD partners: waves w and w+4 share a SIMD)
//   mode 3: every wave interleaves MFMA and VALU in ONE instruction stream (same totals as mode 2 per SIMD pair)
//   mode 4: waves 0-3 MFMA, 4-7 exit          mode 5: waves 4-7 VALU, 0-3 exit
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define NM 16
#define NV 64
__global__ __launch_bounds__(512) void k(float* out, int iters, int mode) {
    const int w = threadIdx.x >> 6;
    f32x4 acc[NM];
    float v[NV];
    const float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-6f;
    bf16x8 av, bv;
    for (int i = 0; i < 8; ++i) { av[i] = (__bf16)(a + i); bv[i] = (__bf16)(b - i); }
    for (int i = 0; i < NM; ++i) acc[i] = f32x4{a, b, a, b};
    for (int i = 0; i < NV; ++i) v[i] = a + i;
    const bool do_m = mode == 0 || mode == 3 || ((mode == 2 || mode == 4) && w < 4);
    const bool do_v = mode == 1 || mode == 3 || ((mode == 2 || mode == 5) && w >= 4);
    if (!do_m && !do_v) return;
    if (mode == 3) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc[i], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < NV / NM; ++j) v[i * (NV / NM) + j] = fmaf(v[i * (NV / NM) + j], b, a);
            }
        }
    } else if (do_m) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc[i], 0, 0, 0);
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NV; ++i) v[i] = fmaf(v[i], b, a);
        }
    }
    float s = 0.f;
    for (int i = 0; i < NM; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < NV; ++i) s += v[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
int main() {
    float* d; hipMalloc(&d, 256 * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    for (int mode = 0; mode < 6; ++mode) {
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d, iters, mode);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d, iters, mode);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // per SIMD: mode 0: 2 waves x NM mfma; mode 2/4: 1 wave
        printf("mode %d: %.3f ms  (per iteration %.1f ns)\n", mode, ms, ms * 1e6 / iters);
    }
    return 0;
}
