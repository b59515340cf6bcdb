// fp32 MFMA issue-rate probe: NACC independent accumulators per wave, dependent distance = NACC MFMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(512) void k(float* out, int iters, float a, float b) {
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(int threads, int blocks_per_cu, float* out) {
    const int iters = 4000, grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(threads), 0, 0, out, iters, 1.f, 1.f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(threads), 0, 0, out, iters, 1.f, 1.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double waves = (double)grid * threads / 64, mf = waves * iters * NACC;
    const double tf = mf * 2048 / (ms * 1e-3) / 1e12;
    const double waves_per_simd = (double)blocks_per_cu * threads / 64 / 4;
    printf("NACC %2d threads %3d blocks/CU %d (waves/SIMD %.0f): %.3f ms  %.1f TFLOP/s  %.1f ns per MFMA per SIMD\n", NACC, threads, blocks_per_cu,
           waves_per_simd, ms, tf, ms * 1e6 / (iters * NACC * waves_per_simd));
}
int main() {
    float* out; hipMalloc(&out, 256 * 8 * 512 * 4);
    run<2>(256, 1, out); run<4>(256, 1, out); run<8>(256, 1, out); run<32>(256, 1, out);
    run<2>(512, 1, out); run<4>(512, 1, out); run<32>(512, 1, out); run<32>(256, 2, out); run<32>(256, 4, out);
    return 0;
}
