// LDS fragment-read rate of the split6 GEMM's images (64-byte rows, chunk swizzle P[(row >> 2) & 3]) against a lane-linear
// pattern: 512 threads per CU, every wave issues NREAD ds_read_b128 per iteration and waits once.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/lds_read_rate.hip -o gpurun_out/lds_read_rate && gpurun_out/lds_read_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int swz64(int row) { return (0x78 >> (2 * ((row >> 2) & 3))) & 3; }
__device__ __forceinline__ int off64(int row, int ch) { return row * 64 + 16 * (ch ^ swz64(row)); }
template <int MODE>
__global__ __launch_bounds__(512, 1) void k(unsigned* out, long long* cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, i16 = lane & 15, g = lane >> 4;
    for (int i = t; i < 144 * 1024 / 4; i += 512) reinterpret_cast<unsigned*>(smem)[i] = i * 2654435761u;
    __syncthreads();
    u32x4 acc = {0, 0, 0, 0};
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        u32x4 v[18];
#pragma unroll
        for (int r = 0; r < 18; ++r) {
            int off;
            const int img = r / 6, f = r % 6;                       // 3 pieces x (4 A fragments + 2 B fragments)
            if (MODE == 0) off = img * 8192 + off64(((w >> 2) * 4 + (f & 3)) * 16 + i16, g) + (f >= 4 ? 73728 + ((w & 3) * 2 + (f & 1)) * 1024 - ((w >> 2) * 4 + (f & 3)) * 1024 : 0);
            else if (MODE == 1) off = img * 8192 + f * 1024 + lane * 16;      // lane-linear
            else off = img * 8192 + ((f * 16 + i16) * 64 + 16 * g);          // un-swizzled 64-byte rows
            off = (off + (it & 1) * 24576) & (144 * 1024 - 16);
            v[r] = *reinterpret_cast<const u32x4*>(smem + off);
        }
#pragma unroll
        for (int r = 0; r < 18; ++r) acc ^= v[r];
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 512 + t] = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
    if (t == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    unsigned* out; long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
    const int iters = 2000;
    for (int mode = 0; mode < 3; ++mode) {
        auto fn = mode == 0 ? k<0> : mode == 1 ? k<1> : k<2>;
        hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(fn, dim3(256), dim3(512), 144 * 1024, 0, out, cyc, iters);
        hipDeviceSynchronize();
        long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        double avg = 0; for (int i = 0; i < 256; ++i) avg += h[i]; avg /= 256;
        printf("mode %d (%s): %.0f cycles per iteration of 8 waves x 18 ds_read_b128 = %.1f cycles per wave-read, %.0f B/clk/CU\n", mode,
               mode == 0 ? "split6 images" : mode == 1 ? "lane-linear" : "64-byte rows, no swizzle", avg / iters, avg / iters / 144, 144.0 * 1024 / (avg / iters));
    }
    return 0;
}
