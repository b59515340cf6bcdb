// tools/micro: as coexec.hip with v_mfma_f32_32x32x2_f32 (64-cycle issue, 16 accumulator registers) - does the LONGER fp32 matrix
// instruction leave vector-issue slots to a VALU partner wave?   8 MFMAs (= the FLOPs of 16 16x16x4 ones) and/or 64 v_fma per iteration.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define NM 8
#define NV 64
__global__ __launch_bounds__(512) void k(float* out, int iters, int mode) {
    const int w = threadIdx.x >> 6;
    f32x16 acc[NM];
    float v[NV];
    const float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-6f;
    for (int i = 0; i < NM; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = a + j;
    for (int i = 0; i < NV; ++i) v[i] = a + i;
    const bool do_m = mode == 0 || mode == 3 || ((mode == 2 || mode == 4) && w < 4);
    const bool do_v = mode == 1 || mode == 3 || ((mode == 2 || mode == 5) && w >= 4);
    if (!do_m && !do_v) return;
    if (mode == 3) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < NV / NM; ++j) v[i * (NV / NM) + j] = fmaf(v[i * (NV / NM) + j], b, a);
            }
        }
    } else if (do_m) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NV; ++i) v[i] = fmaf(v[i], b, a);
        }
    }
    float s = 0.f;
    for (int i = 0; i < NM; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
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
        printf("mode %d: %.3f ms  (per iteration %.1f ns)\n", mode, ms, ms * 1e6 / iters);
    }
    return 0;
}
