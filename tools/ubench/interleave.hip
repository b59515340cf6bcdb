// Microbenchmark: what does a vector instruction cost when it sits BETWEEN the matrix instructions of the same wave?
// One wave per SIMD (256 threads per CU) - or two (512) - runs `iters` x 8 independent MFMAs with NV vector instructions after each.
//     hipcc --offload-arch=gfx950 -O3 -Wno-unused-value tools/ubench/interleave.hip -o tools/ubench/interleave && tools/ubench/interleave
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND, int NV, int VOP, int THREADS>
__global__ __launch_bounds__(THREADS, 1) void k(float* out, int iters) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float a = threadIdx.x * 1e-3f, b = 1.f;
    bf16x8 ab = {1, 2, 3, 4, 5, 6, 7, 8};
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x + i;
    const float c = 1.0001f, d = 1e-3f;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 xp[8];
    for (int i = 0; i < 8; ++i) xp[i] = f32x2{x[i], 1.f};
    const f32x2 cp = {c, d};
    int sc = iters;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (KIND == 0) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
            else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %1, %0" : "+v"(acc[i]) : "v"(ab));
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                if (VOP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[(i + v) & 7]) : "v"(c), "v"(d));
                if (VOP == 1) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x[(i + v) & 7]) : "v"(c));
                if (VOP == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(xp[(i + v) & 7]) : "v"(cp));
                if (VOP == 5) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x[(i + v) & 7]) : "v"(c), "v"(d));
                if (VOP == 6) asm volatile("v_mov_b32 %0, %1" : "+v"(x[(i + v) & 7]) : "v"(c));
                if (VOP == 7) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x[(i + v) & 7]) : "v"(c));
                if (VOP == 8) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(xp[(i + v) & 7]) : "v"(cp));
                if (VOP == 9) asm volatile("ds_write_b128 %0, %1" :: "v"((threadIdx.x & 63) * 16), "v"(acc[7]) : "memory");
                if (VOP == 3) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sc));
                if (VOP == 4) asm volatile("ds_read_b32 %0, %1" : "=v"(x[(i + v) & 7]) : "v"((threadIdx.x & 63) * 4) : "memory");
            }
        }
        if (VOP == 4 || VOP == 9) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + x[i] + xp[i][0] + xp[i][1];
    out[blockIdx.x * THREADS + threadIdx.x] = s + sc;
}

template <int KIND, int NV, int VOP, int THREADS>
float run(float* out, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND, NV, VOP, THREADS>), dim3(256), dim3(THREADS), 0, 0, out, iters);
    hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<KIND, NV, VOP, THREADS>), dim3(256), dim3(THREADS), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 3 * 1000;
}

template <int KIND, int VOP, int THREADS>
void row(float* out, const char* name, int iters) {
    const float t0 = run<KIND, 0, VOP, THREADS>(out, iters), t1 = run<KIND, 1, VOP, THREADS>(out, iters), t2 = run<KIND, 2, VOP, THREADS>(out, iters),
                t4 = run<KIND, 4, VOP, THREADS>(out, iters);
    const double per = (double)iters * 8 * (THREADS / 256);        // MFMAs per SIMD
    printf("%-10s %s, %d wave(s)/SIMD: 0 / 1 / 2 / 4 per MFMA: %7.1f %7.1f %7.1f %7.1f us  -> cost per instruction (1 / 2 / 4 per MFMA): %5.1f %5.1f %5.1f ns x 1e-3 = cycles@2.4GHz %4.1f %4.1f %4.1f\n",
           name, KIND ? "bf16 16x16x32" : "fp32 16x16x4 ", THREADS / 256, t0, t1, t2, t4, (t1 - t0) / per * 1e3, (t2 - t0) / per / 2 * 1e3, (t4 - t0) / per / 4 * 1e3,
           (t1 - t0) / per * 2400, (t2 - t0) / per / 2 * 2400, (t4 - t0) / per / 4 * 2400);
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 512 * 4);
    const int iters = 20000;
    row<0, 0, 256>(out, "v_fma_f32", iters);   row<1, 0, 256>(out, "v_fma_f32", iters);
    row<0, 7, 256>(out, "v_sub_f32", iters);   row<1, 7, 256>(out, "v_sub_f32", iters);
    row<0, 2, 256>(out, "v_pk_add_f32", iters); row<1, 2, 256>(out, "v_pk_add_f32", iters);
    row<0, 1, 256>(out, "v_and_b32", iters);   row<1, 1, 256>(out, "v_and_b32", iters);
    row<0, 5, 256>(out, "v_perm_b32", iters);  row<1, 5, 256>(out, "v_perm_b32", iters);
    row<0, 6, 256>(out, "v_mov_b32", iters);   row<1, 6, 256>(out, "v_mov_b32", iters);
    row<0, 8, 256>(out, "v_lshl_add_u64", iters); row<1, 8, 256>(out, "v_lshl_add_u64", iters);
    row<0, 4, 256>(out, "ds_read_b32", iters); row<1, 4, 256>(out, "ds_read_b32", iters);
    row<0, 9, 256>(out, "ds_write_b128", iters); row<1, 9, 256>(out, "ds_write_b128", iters);
    row<1, 0, 512>(out, "v_fma_f32", iters);   row<1, 2, 512>(out, "v_pk_add_f32", iters);
    row<1, 1, 512>(out, "v_and_b32", iters);   row<0, 1, 512>(out, "v_and_b32", iters);
    return 0;
}
