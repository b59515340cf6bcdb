// HBM write rate of a [T][N] bf16 (or fp32: 16-byte elements are what matters) matrix written by 64 x (128 or 256 byte) wave tiles in
// three store patterns (no loads; 16-byte stores):
//   0: the GEMM epilogues' pattern - lane (i16, g) writes token 16 a + i16, bytes 64 h + 16 g .. : every instruction covers 16 tokens x
//      a 64-BYTE run; the other half of each 128-byte line comes from a later instruction
//   1: lane -> token lane / 8 (+ 8 k), chunk lane % 8: every instruction covers 8 tokens x a full 128-byte line
//   2: as 0 but the two halves of a line in consecutive instructions (h inner loop) - what pattern 0 already does per a
//   hipcc --offload-arch=gfx950 -O3 tools/micro/store_runs.hip -o gpurun_out/store_runs && gpurun_out/store_runs
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(512, 1) void k(unsigned char* out, int T, int rowbytes, int ntiles, int tiles_n) {
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, i16 = lane & 15, g = lane >> 4;
    const int wm = w >> 1, wn = w & 1;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int tm = tile / tiles_n, tn = tile % tiles_n;
        unsigned char* base = out + (size_t)(tm * 256 + wm * 64) * rowbytes + tn * 256 + wn * 128;       // wave tile: 64 tokens x 128 bytes
        const u32x4 v = {(unsigned)tile, (unsigned)lane, 3u, 4u};
        if (MODE == 0 || MODE == 2) {
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int h = 0; h < 2; ++h) *reinterpret_cast<u32x4*>(base + (size_t)(16 * a + i16) * rowbytes + 64 * h + 16 * g) = v;
        } else {
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) *reinterpret_cast<u32x4*>(base + (size_t)(8 * kk + (lane >> 3)) * rowbytes + 16 * (lane & 7)) = v;
        }
    }
}
template <int MODE>
float run(unsigned char* buf, int T, int rowbytes) {
    const int tiles_n = rowbytes / 256, ntiles = (T / 256) * tiles_n;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, buf, T, rowbytes, ntiles, tiles_n);
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, buf, T, rowbytes, ntiles, tiles_n);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 10 * 1e3f;
}
int main() {
    const int shapes[][2] = {{131072, 2048}, {524288, 1024}, {524288, 256}, {131072, 4096}};
    for (auto& sh : shapes) {
        const int T = sh[0], rb = sh[1];
        unsigned char* buf; hipMalloc(&buf, (size_t)T * rb);
        const float a = run<0>(buf, T, rb), b = run<1>(buf, T, rb);
        const double mb = (double)T * rb / 1e6;
        printf("T=%d row=%d B (%.0f MB): 64-byte runs %.1f us (%.2f TB/s)   128-byte lines %.1f us (%.2f TB/s)\n", T, rb, mb, a, mb / a,
               b, mb / b);
        hipFree(buf);
    }
    return 0;
}
