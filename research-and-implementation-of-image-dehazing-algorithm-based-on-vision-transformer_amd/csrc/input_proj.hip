// K9: InputProj (M1:659-682): Conv2d(3, E, 3x3, pad 1) + LeakyReLU(0.01) from the NCHW image straight into the token layout
// [B, H*W, E], and its backward (weight / bias gradients; the image itself needs no gradient).  27 MACs per output element:
// nothing for the matrix pipe - HBM-bound streaming kernels (3 floats in, E floats out per pixel).
#include "common.h"

namespace {

constexpr int TS = 16;               // 16 x 16 pixel tile per workgroup (256 threads)
constexpr int HS = TS + 2;

// forward: thread -> (pixel, 4-channel group); the E/4 lanes of a pixel write its E floats as one contiguous run
template <int E, typename T>
__global__ __launch_bounds__(256) void input_proj_fwd_kernel(const float* __restrict__ img, const float* __restrict__ w,
                                                             const float* __restrict__ bias, T* __restrict__ y, int H, int W,
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
        st4(y + ((size_t)b * H * W + (size_t)yy * W + xx) * E + 4 * cq, a);
    }
}

// backward: dpre = dy * (y > 0 ? 1 : slope); [dw | db][co][tap] += sum_p dpre[p][co] xcol[p][tap] with xcol[p][27] = 1 (the bias
// column) - a [E x 32] = dpre^T [E x pixels] . xcol [pixels x 32] contraction per 16 x 16-pixel tile on the matrix pipe: both
// dpre is staged in LDS (row = pixel, stride = 16 mod 32 floats so that the four pixel rows of a k-step hit disjoint banks), xcol is
// read straight out of the halo tile,
// wave w contracts pixels 64 w .. 64 w + 63, accumulators live across the tiles of a persistent workgroup, one cross-wave
// reduction and one atomic per output and workgroup at the end.
template <int E, typename T>
__global__ __launch_bounds__(256) void input_proj_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y,
                                                             const float* __restrict__ img, float* __restrict__ dw,
                                                             float* __restrict__ db, int H, int W, float slope, int tiles_x,
                                                             int tiles_y, int ntiles) {
    constexpr int TY = 8, NP = TY * TS;             // 8 x 16-pixel tiles: 51 KiB of LDS at E = 32 -> three workgroups per CU
    constexpr int SD = E + 16;                     // row stride of the dpre image (floats): 16 mod 32
    constexpr int MT = E / 16;                     // output-channel tiles
    __shared__ float xs[3][TY + 2][HS];
    __shared__ __attribute__((aligned(16))) float ds[NP * SD];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int i16 = lane & 15, g = lane >> 4;
    f32x4 acc[MT][2];
#pragma unroll
    for (int a = 0; a < MT; ++a) { acc[a][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[a][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    auto tap_base = [&](int tap) { const int c = tap / 9, r = tap - 9 * c, ky = r / 3, kx = r - 3 * ky; return (c * (TY + 2) + ky) * HS + kx; };
    const int base0 = tap_base(i16);
    const int base1 = 16 + i16 < 27 ? tap_base(16 + i16) : -1;
    const float const1 = 16 + i16 == 27 ? 1.f : 0.f;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
        const int x0 = tx * TS - 1, y0 = ty * TY - 1;
        __syncthreads();
        for (int e = t; e < 3 * (TY + 2) * HS; e += 256) {
            const int c = e / ((TY + 2) * HS), r = e % ((TY + 2) * HS), yy = y0 + r / HS, xx = x0 + r % HS;
            xs[c][r / HS][r % HS] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? img[((size_t)(b * 3 + c) * H + yy) * W + xx] : 0.f;
        }
        for (int e = t; e < NP * (E / 4); e += 256) {
            const int p = e / (E / 4), cq = e % (E / 4);
            const int yy = ty * TY + p / TS, xx = tx * TS + p % TS;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (yy < H && xx < W) {
                const size_t o = ((size_t)b * H * W + (size_t)yy * W + xx) * E + 4 * cq;
                const float4 gq = ld4(dy + o), yv = ld4(y + o);
                v = make_float4(gq.x * (yv.x > 0.f ? 1.f : slope), gq.y * (yv.y > 0.f ? 1.f : slope), gq.z * (yv.z > 0.f ? 1.f : slope),
                                gq.w * (yv.w > 0.f ? 1.f : slope));
            }
            *reinterpret_cast<float4*>(&ds[p * SD + 4 * cq]) = v;
        }
        __syncthreads();
        // B[k = pixel][j = tap] = x[c, py + ky, px + kx] is read straight from the halo tile (a lane owns taps i16 and 16 + i16:
        // its tap offsets are constants; tap 27 is the constant 1 of the bias column, 28..31 are zero)
        const float* Af = ds + (32 * w + g) * SD + i16;          // A[i = co][k = pixel] = ds[pixel][co]
        const float* xf = &xs[0][0][0];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int p = 32 * w + 4 * s + g;
            const int off = (p / TS) * HS + (p % TS);
            float af[MT], bf[2];
#pragma unroll
            for (int a = 0; a < MT; ++a) af[a] = Af[4 * s * SD + 16 * a];
            bf[0] = xf[base0 + off];
            bf[1] = base1 >= 0 ? xf[base1 + off] : const1;
#pragma unroll
            for (int a = 0; a < MT; ++a) { acc[a][0] = mfma16(af[a], bf[0], acc[a][0]); acc[a][1] = mfma16(af[a], bf[1], acc[a][1]); }
        }
    }
    // cross-wave reduction through LDS (the dpre image is dead), then one atomic per output
    __syncthreads();
    float* red = ds;                                              // [4 waves][E][32]
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int bb = 0; bb < 2; ++bb)
#pragma unroll
            for (int j = 0; j < 4; ++j) red[(w * E + 16 * a + 4 * g + j) * 32 + 16 * bb + i16] = acc[a][bb][j];
    __syncthreads();
    for (int o = t; o < E * 28; o += 256) {
        const int co = o / 28, tap = o % 28;
        const float v = red[co * 32 + tap] + red[(E + co) * 32 + tap] + red[(2 * E + co) * 32 + tap] + red[(3 * E + co) * 32 + tap];
        if (tap < 27) atomicAdd(dw + co * 27 + tap, v);
        else atomicAdd(db + co, v);
    }
}

}  // namespace

#define DT_SWITCH(dtype, who, CALL)                                                      \
    do {                                                                                 \
        if ((dtype) == DHZ_F32) { typedef float T; CALL; }                               \
        else if ((dtype) == DHZ_BF16) { typedef bf16s T; CALL; }                         \
        else { dhz_set_error("%s: unknown dtype %d", who, (int)(dtype)); return DHZ_EINVAL; } \
    } while (0)

extern "C" int dhz_input_proj_fwd_dt(const float* img, const float* w, const float* bias, void* y, int B, int H, int W, int E,
                                     float slope, int dtype, void* stream) {
    DHZ_REQUIRE(img && w && bias && y, "dhz_input_proj_fwd: null pointer");
    DHZ_REQUIRE(B > 0 && H > 0 && W > 0 && (E == 16 || E == 32 || E == 64), "dhz_input_proj_fwd: E=%d (supported: 16, 32, 64)", E);
    const int tiles_x = (W + TS - 1) / TS, tiles_y = (H + TS - 1) / TS;
    hipStream_t s = (hipStream_t)stream;
#define LAUNCH(EE) hipLaunchKernelGGL((input_proj_fwd_kernel<EE, T>), dim3(B * tiles_x * tiles_y), dim3(256), 0, s, img, w, bias, (T*)y, H, W, slope, tiles_x, tiles_y)
    DT_SWITCH(dtype, "dhz_input_proj_fwd", if (E == 16) LAUNCH(16); else if (E == 32) LAUNCH(32); else LAUNCH(64));
#undef LAUNCH
    DHZ_CHECK_LAUNCH("dhz_input_proj_fwd");
    return DHZ_OK;
}
extern "C" int dhz_input_proj_fwd(const float* img, const float* w, const float* bias, float* y, int B, int H, int W, int E,
                                  float slope, void* stream) {
    return dhz_input_proj_fwd_dt(img, w, bias, y, B, H, W, E, slope, DHZ_F32, stream);
}

extern "C" int dhz_input_proj_bwd_dt(const void* dy, const void* y, const float* img, float* dw, float* db, int B, int H, int W, int E,
                                     float slope, int dtype, void* stream) {
    DHZ_REQUIRE(dy && y && img && dw && db, "dhz_input_proj_bwd: null pointer");
    DHZ_REQUIRE(B > 0 && H > 0 && W > 0 && (E == 16 || E == 32 || E == 64), "dhz_input_proj_bwd: E=%d (supported: 16, 32, 64)", E);
    const int tiles_x = (W + TS - 1) / TS, tiles_y = (H + 8 - 1) / 8;
    const int ntiles = B * tiles_x * tiles_y;
    const int ncu = dhz_num_cus();
    const int grid = ntiles < ncu ? ntiles : ncu;      // every workgroup ends with 28 E same-address atomics: keep them few
    hipStream_t s = (hipStream_t)stream;
#define LAUNCH(EE) hipLaunchKernelGGL((input_proj_bwd_kernel<EE, T>), dim3(grid), dim3(256), 0, s, (const T*)dy, (const T*)y, img, dw, db, H, W, slope, tiles_x, tiles_y, ntiles)
    DT_SWITCH(dtype, "dhz_input_proj_bwd", if (E == 16) LAUNCH(16); else if (E == 32) LAUNCH(32); else LAUNCH(64));
#undef LAUNCH
    DHZ_CHECK_LAUNCH("dhz_input_proj_bwd");
    return DHZ_OK;
}
extern "C" int dhz_input_proj_bwd(const float* dy, const float* y, const float* img, float* dw, float* db, int B, int H, int W, int E,
                                  float slope, void* stream) {
    return dhz_input_proj_bwd_dt(dy, y, img, dw, db, B, H, W, E, slope, DHZ_F32, stream);
}
