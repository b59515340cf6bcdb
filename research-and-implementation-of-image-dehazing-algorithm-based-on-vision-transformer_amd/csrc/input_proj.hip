// K9: InputProj (M1:659-682): Conv2d(3, E, 3x3, pad 1) + LeakyReLU(0.01) from the NCHW image straight into the token layout
// [B, H*W, E], and its backward (weight / bias gradients; the image itself needs no gradient).  27 MACs per output element:
// nothing for the matrix pipe - HBM-bound streaming kernels (3 floats in, E floats out per pixel).
#include "common.h"

namespace {

constexpr int TS = 16;               // 16 x 16 pixel tile per workgroup (256 threads)
constexpr int HS = TS + 2;

// forward: thread -> (pixel, 4-channel group); the E/4 lanes of a pixel write its E floats as one contiguous run
template <int E>
__global__ __launch_bounds__(256) void input_proj_fwd_kernel(const float* __restrict__ img, const float* __restrict__ w,
                                                             const float* __restrict__ bias, float* __restrict__ y, int H, int W,
                                                             float slope, int tiles_x, int tiles_y) {
    __shared__ float xs[3][HS][HS];
    __shared__ __attribute__((aligned(16))) float ws[27][E];           // [ci*9 + ky*3 + kx][co]
    const int t = threadIdx.x;
    int bid = blockIdx.x;
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int b = bid / tiles_y;
    const int x0 = tx * TS - 1, y0 = ty * TS - 1;
    for (int e = t; e < 3 * HS * HS; e += 256) {
        const int c = e / (HS * HS), r = e % (HS * HS), yy = y0 + r / HS, xx = x0 + r % HS;
        xs[c][r / HS][r % HS] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? img[((size_t)(b * 3 + c) * H + yy) * W + xx] : 0.f;
    }
    for (int e = t; e < 27 * E; e += 256) ws[e / E][e % E] = w[(e % E) * 27 + e / E];
    __syncthreads();
    constexpr int Q = E / 4;                       // lanes per pixel
    constexpr int PPP = 256 / Q;                   // pixels per pass
    const int cq = t % Q;
    const float4 bv = *reinterpret_cast<const float4*>(bias + 4 * cq);
    for (int p = t / Q; p < TS * TS; p += PPP) {
        const int py = p / TS, px = p % TS;
        const int yy = ty * TS + py, xx = tx * TS + px;
        if (yy >= H || xx >= W) continue;
        float4 a = bv;
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float v = xs[c][py + ky][px + kx];
                    const float4 wv = *reinterpret_cast<const float4*>(&ws[c * 9 + ky * 3 + kx][4 * cq]);
                    a.x += v * wv.x; a.y += v * wv.y; a.z += v * wv.z; a.w += v * wv.w;
                }
        a.x = a.x > 0.f ? a.x : slope * a.x; a.y = a.y > 0.f ? a.y : slope * a.y;
        a.z = a.z > 0.f ? a.z : slope * a.z; a.w = a.w > 0.f ? a.w : slope * a.w;
        *reinterpret_cast<float4*>(y + ((size_t)b * H * W + (size_t)yy * W + xx) * E + 4 * cq) = a;
    }
}

// backward: dpre = dy * (y > 0 ? 1 : slope); dw[co][ci][ky][kx] += sum_p dpre[p][co] img[ci, p + off]; db[co] += sum_p dpre[p][co].
// One workgroup per group of 16 x 16 tiles (persistent); thread -> output slots (co, tap) with co fastest, accumulated in
// registers over all its tiles; one atomic per slot and workgroup at the end.
template <int E>
__global__ __launch_bounds__(256) void input_proj_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                             const float* __restrict__ img, float* __restrict__ dw,
                                                             float* __restrict__ db, int H, int W, float slope, int tiles_x,
                                                             int tiles_y, int ntiles) {
    constexpr int NOUT = 28 * E;                   // 27 taps + bias, co fastest
    constexpr int NPT = (NOUT + 255) / 256;        // outputs per thread
    __shared__ float xs[3][HS][HS];
    __shared__ __attribute__((aligned(16))) float ds[TS * TS][E + 4];
    const int t = threadIdx.x;
    float accv[NPT];
#pragma unroll
    for (int i = 0; i < NPT; ++i) accv[i] = 0.f;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
        const int x0 = tx * TS - 1, y0 = ty * TS - 1;
        __syncthreads();
        for (int e = t; e < 3 * HS * HS; e += 256) {
            const int c = e / (HS * HS), r = e % (HS * HS), yy = y0 + r / HS, xx = x0 + r % HS;
            xs[c][r / HS][r % HS] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? img[((size_t)(b * 3 + c) * H + yy) * W + xx] : 0.f;
        }
        for (int e = t; e < TS * TS * (E / 4); e += 256) {
            const int p = e / (E / 4), cq = e % (E / 4);
            const int yy = ty * TS + p / TS, xx = tx * TS + p % TS;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (yy < H && xx < W) {
                const size_t o = ((size_t)b * H * W + (size_t)yy * W + xx) * E + 4 * cq;
                const float4 g = *reinterpret_cast<const float4*>(dy + o), yv = *reinterpret_cast<const float4*>(y + o);
                v = make_float4(g.x * (yv.x > 0.f ? 1.f : slope), g.y * (yv.y > 0.f ? 1.f : slope), g.z * (yv.z > 0.f ? 1.f : slope),
                                g.w * (yv.w > 0.f ? 1.f : slope));
            }
            *reinterpret_cast<float4*>(&ds[p][4 * cq]) = v;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const int o = t + 256 * i;
            if (o < NOUT) {
                const int co = o % E, tap = o / E;               // tap 27 = bias
                float a = 0.f;
                if (tap < 27) {
                    const int c = tap / 9, ky = (tap % 9) / 3, kx = tap % 3;
                    for (int p = 0; p < TS * TS; ++p) a += ds[p][co] * xs[c][p / TS + ky][p % TS + kx];
                } else {
                    for (int p = 0; p < TS * TS; ++p) a += ds[p][co];
                }
                accv[i] += a;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
        const int o = t + 256 * i;
        if (o < NOUT) {
            const int co = o % E, tap = o / E;
            if (tap < 27) atomicAdd(dw + co * 27 + tap, accv[i]);
            else atomicAdd(db + co, accv[i]);
        }
    }
}

}  // namespace

extern "C" int dhz_input_proj_fwd(const float* img, const float* w, const float* bias, float* y, int B, int H, int W, int E,
                                  float slope, void* stream) {
    DHZ_REQUIRE(img && w && bias && y, "dhz_input_proj_fwd: null pointer");
    DHZ_REQUIRE(B > 0 && H > 0 && W > 0 && (E == 32 || E == 64), "dhz_input_proj_fwd: E=%d (supported: 32, 64)", E);
    const int tiles_x = (W + TS - 1) / TS, tiles_y = (H + TS - 1) / TS;
    hipStream_t s = (hipStream_t)stream;
    if (E == 32) hipLaunchKernelGGL(input_proj_fwd_kernel<32>, dim3(B * tiles_x * tiles_y), dim3(256), 0, s, img, w, bias, y, H, W, slope, tiles_x, tiles_y);
    else hipLaunchKernelGGL(input_proj_fwd_kernel<64>, dim3(B * tiles_x * tiles_y), dim3(256), 0, s, img, w, bias, y, H, W, slope, tiles_x, tiles_y);
    DHZ_CHECK_LAUNCH("dhz_input_proj_fwd");
    return DHZ_OK;
}

extern "C" int dhz_input_proj_bwd(const float* dy, const float* y, const float* img, float* dw, float* db, int B, int H, int W, int E,
                                  float slope, void* stream) {
    DHZ_REQUIRE(dy && y && img && dw && db, "dhz_input_proj_bwd: null pointer");
    DHZ_REQUIRE(B > 0 && H > 0 && W > 0 && (E == 32 || E == 64), "dhz_input_proj_bwd: E=%d (supported: 32, 64)", E);
    const int tiles_x = (W + TS - 1) / TS, tiles_y = (H + TS - 1) / TS;
    const int ntiles = B * tiles_x * tiles_y;
    const int grid = ntiles < 1024 ? ntiles : 1024;
    hipStream_t s = (hipStream_t)stream;
    if (E == 32) hipLaunchKernelGGL(input_proj_bwd_kernel<32>, dim3(grid), dim3(256), 0, s, dy, y, img, dw, db, H, W, slope, tiles_x, tiles_y, ntiles);
    else hipLaunchKernelGGL(input_proj_bwd_kernel<64>, dim3(grid), dim3(256), 0, s, dy, y, img, dw, db, H, W, slope, tiles_x, tiles_y, ntiles);
    DHZ_CHECK_LAUNCH("dhz_input_proj_bwd");
    return DHZ_OK;
}
