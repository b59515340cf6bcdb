// K9b - the output projection of the U-shaped net (My_model_1.py:696-723): Conv2d(C -> 3, 3x3, pad 1) from the token layout
// [B, H*W, C] to an NCHW image [B, 3, H, W], its backward-data and its weight/bias gradient.
//
// The library serves this "thin" convolution (3 output channels) with implicit-GEMM kernels built for wide outputs:
// 235 us forward and 340 us backward at B = 32, 128 x 128, C = 64 - 0.6 TB/s on a problem whose whole traffic is the
// 134 MB token tensor.  With 3 output channels there is nothing for the matrix pipe to do (N = 3 pads to 16), so these
// are VALU kernels shaped by the data movement:
//   forward : one workgroup = 8 x 16 output pixels; the 10 x 18 halo tile of tokens goes to LDS once (row stride C + 4:
//             a lane <-> pixel b128 read is conflict-free); lane <-> pixel, wave pair <-> half of the channels, the
//             27*C weights are broadcast reads of an LDS copy in (channel quad, tap, output) order; the two channel
//             halves meet in LDS; 3 coalesced row stores.
//   dgrad   : lane <-> channel (the token row is the contiguous axis of the output), the 27 weights of the lane's channel
//             live in registers, the 3-channel gradient tile is read as LDS broadcasts.
//   wgrad   : persistent workgroups, lane <-> channel, 27 accumulators per lane over all pixels of the workgroup's tiles,
//             the token halo tile in LDS, dy broadcast; one atomic per (output, channel, tap) per workgroup at the end.
#include "common.h"

namespace {

constexpr int TW = 16, TH = 8;                 // output pixels per workgroup
constexpr int HW_ = TW + 2, HH_ = TH + 2;      // halo tile
constexpr int NPOS = HH_ * HW_;                // 180

template <int C>
struct ThinSmem {
    static constexpr int XS = C + 4;
    float x[NPOS * XS];                        // token halo tile
    float w[27 * C];                           // forward: [c/4][tap][o][4]
    float part[2][3][TH * TW];                 // forward: per channel-half partial sums
};

// stage the 10 x 18 halo tile (zero outside the image) of a token tensor [B, H*W, C] or (BLOCKED) of a channel-blocked
// NCHW8c map [B, C/8, H, W, 8] (the layout of the VGG feature engine)
template <int C, bool BLOCKED = false, typename T = float>
__device__ __forceinline__ void stage_tokens(float* xs, const T* __restrict__ x, int bimg, int ty, int tx, int H, int W) {
    constexpr int XS = C + 4, C4 = C / 4;
    const int t = threadIdx.x;
    const size_t ib = (size_t)bimg * H * W;
    for (int e = t; e < NPOS * C4; e += 256) {
        const int pos = BLOCKED ? e % NPOS : e / C4, c4 = BLOCKED ? e / NPOS : e % C4;     // consecutive lanes walk the contiguous axis
        const int yy = ty * TH - 1 + pos / HW_, xx = tx * TW - 1 + pos % HW_;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
            const size_t o = BLOCKED ? (((size_t)bimg * (C / 8) + c4 / 2) * H * W + (size_t)yy * W + xx) * 8 + (c4 & 1) * 4
                                     : (ib + (size_t)yy * W + xx) * C + c4 * 4;
            v = ld4v(x + o);
        }
        *reinterpret_cast<f32x4*>(&xs[pos * XS + c4 * 4]) = v;
    }
}

// TRANSPOSED: w is a [C, 3, 3, 3] tensor (a 3 -> C convolution's weight) and the kernel computes that layer's
// backward-data: y[b, o, p] = sum_{c, ky, kx} w[c][o][2 - ky][2 - kx] x[b, c, p + (ky - 1, kx - 1)]
template <int C, bool BLOCKED, bool TRANSPOSED, typename T = float>
__global__ __launch_bounds__(256) void thin_conv_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ bias, float* __restrict__ y, int H,
                                                            int W, int tiles_x, int tiles_y) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    ThinSmem<C>& sm = *reinterpret_cast<ThinSmem<C>*>(smem_raw);
    constexpr int XS = C + 4, CH = C / 2;
    const int t = threadIdx.x;
    const int tx = blockIdx.x % tiles_x, ty = (blockIdx.x / tiles_x) % tiles_y, bimg = blockIdx.x / (tiles_x * tiles_y);
    // weights -> LDS in (channel quad, tap, output, channel % 4) order: w[o][c][tap] -> sm.w[((c/4 * 9 + tap) * 3 + o) * 4 + c%4]
    for (int e = t; e < 27 * C; e += 256) {
        int o, c, tap;
        if (TRANSPOSED) { c = e / 27; o = (e / 9) % 3; tap = 8 - e % 9; }      // w[c][o][ky][kx] acts at tap (2-ky, 2-kx)
        else { o = e / (9 * C); c = (e / 9) % C; tap = e % 9; }
        sm.w[((c / 4 * 9 + tap) * 3 + o) * 4 + (c & 3)] = w[e];
    }
    stage_tokens<C, BLOCKED, T>(sm.x, x, bimg, ty, tx, H, W);
    __syncthreads();
    const int half = t >> 7, pix = t & 127;
    const int py = pix / TW, px = pix % TW;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    const float* xb = sm.x + (py * HW_ + px) * XS + half * CH;
    const float* wb = sm.w + (half * (CH / 4)) * 9 * 12;
#pragma unroll 2
    for (int c4 = 0; c4 < CH / 4; ++c4) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const f32x4 xv = *reinterpret_cast<const f32x4*>(xb + ((tap / 3) * HW_ + tap % 3) * XS + c4 * 4);
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(wb + (c4 * 9 + tap) * 12);         // broadcast reads
            const f32x4 w1 = *reinterpret_cast<const f32x4*>(wb + (c4 * 9 + tap) * 12 + 4);
            const f32x4 w2 = *reinterpret_cast<const f32x4*>(wb + (c4 * 9 + tap) * 12 + 8);
            a0 += xv[0] * w0[0] + xv[1] * w0[1] + xv[2] * w0[2] + xv[3] * w0[3];
            a1 += xv[0] * w1[0] + xv[1] * w1[1] + xv[2] * w1[2] + xv[3] * w1[3];
            a2 += xv[0] * w2[0] + xv[1] * w2[1] + xv[2] * w2[2] + xv[3] * w2[3];
        }
    }
    sm.part[half][0][pix] = a0; sm.part[half][1][pix] = a1; sm.part[half][2][pix] = a2;
    __syncthreads();
    for (int e = t; e < 3 * TH * TW; e += 256) {
        const int o = e / (TH * TW), p = e % (TH * TW);
        const int yy = ty * TH + p / TW, xx = tx * TW + p % TW;
        if (yy < H && xx < W)
            y[(((size_t)bimg * 3 + o) * H + yy) * W + xx] = sm.part[0][o][p] + sm.part[1][o][p] + (bias ? bias[o] : 0.f);
    }
}

// dy halo tile [NPOS][4] (3 used), zero outside the image
__device__ __forceinline__ void stage_dy(float* ds, const float* __restrict__ dy, int bimg, int ty, int tx, int H, int W) {
    for (int e = threadIdx.x; e < NPOS * 3; e += 256) {
        const int o = e / NPOS, pos = e % NPOS;
        const int yy = ty * TH - 1 + pos / HW_, xx = tx * TW - 1 + pos % HW_;
        float v = 0.f;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) v = dy[(((size_t)bimg * 3 + o) * H + yy) * W + xx];
        ds[pos * 4 + o] = v;
    }
}

// dx[b, p, c] = sum_{o, ky, kx} w[o][c][ky][kx] * dy[b, o, p + (1 - ky, 1 - kx)]
template <int C, typename T = float>
__global__ __launch_bounds__(256) void thin_conv_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                              T* __restrict__ dx, int H, int W, int tiles_x, int tiles_y) {
    __shared__ __attribute__((aligned(16))) float ds[NPOS * 4];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int tx = blockIdx.x % tiles_x, ty = (blockIdx.x / tiles_x) % tiles_y, bimg = blockIdx.x / (tiles_x * tiles_y);
    stage_dy(ds, dy, bimg, ty, tx, H, W);
    __syncthreads();
    const size_t ib = (size_t)bimg * H * W;
#pragma unroll 1
    for (int cb = 0; cb < C; cb += 64) {
        const int c = cb + lane;
        float wr[3][9];
#pragma unroll
        for (int o = 0; o < 3; ++o)
#pragma unroll
            for (int k = 0; k < 9; ++k) wr[o][k] = w[(o * C + c) * 9 + k];
#pragma unroll 2
        for (int i = 0; i < TH * TW / 4; ++i) {
            const int pix = wv * (TH * TW / 4) + i;
            const int py = pix / TW, px = pix % TW;
            const int yy = ty * TH + py, xx = tx * TW + px;
            float acc = 0.f;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const f32x4 g = *reinterpret_cast<const f32x4*>(&ds[((py + 2 - ky) * HW_ + px + 2 - kx) * 4]);   // broadcast
                    acc += wr[0][ky * 3 + kx] * g[0] + wr[1][ky * 3 + kx] * g[1] + wr[2][ky * 3 + kx] * g[2];
                }
            if (yy < H && xx < W) st1(dx + (ib + (size_t)yy * W + xx) * C + c, acc);
        }
    }
}

// dw[o][c][ky][kx] += sum_p dy[o][p] x[p + (ky - 1, kx - 1)][c] ; db[o] += sum_p dy[o][p]
template <int C, typename T = float>
__global__ __launch_bounds__(256) void thin_conv_wgrad_kernel(const float* __restrict__ dy, const T* __restrict__ x,
                                                              float* __restrict__ dw, float* __restrict__ db, int B, int H,
                                                              int W, int tiles_x, int tiles_y) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    ThinSmem<C>& sm = *reinterpret_cast<ThinSmem<C>*>(smem_raw);
    constexpr int XS = C + 4;
    float* ds = sm.w;                                        // dy tile [NPOS][4] (the weight slot is free here)
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    constexpr int NCB = C / 64;
    float acc[NCB][3][9];
    float accb[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NCB; ++j)
#pragma unroll
        for (int o = 0; o < 3; ++o)
#pragma unroll
            for (int k = 0; k < 9; ++k) acc[j][o][k] = 0.f;
    const int ntiles = B * tiles_x * tiles_y;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, bimg = tile / (tiles_x * tiles_y);
        __syncthreads();
        stage_tokens<C, false, T>(sm.x, x, bimg, ty, tx, H, W);
        stage_dy(ds, dy, bimg, ty, tx, H, W);
        __syncthreads();
#pragma unroll 1
        for (int i = 0; i < TH * TW / 4; ++i) {
            const int pix = wv * (TH * TW / 4) + i;
            const int py = pix / TW, px = pix % TW;
            if (ty * TH + py >= H || tx * TW + px >= W) continue;
            const f32x4 g = *reinterpret_cast<const f32x4*>(&ds[((py + 1) * HW_ + px + 1) * 4]);    // broadcast
            if (lane == 0) { accb[0] += g[0]; accb[1] += g[1]; accb[2] += g[2]; }
#pragma unroll
            for (int j = 0; j < NCB; ++j)
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    const float xv = sm.x[((py + k / 3) * HW_ + px + k % 3) * XS + 64 * j + lane];
                    acc[j][0][k] += g[0] * xv; acc[j][1][k] += g[1] * xv; acc[j][2][k] += g[2] * xv;
                }
        }
    }
    // 4 waves hold partial sums of the same (o, c, k): fold through LDS, then one atomic each
    __syncthreads();
    float* red = sm.x;                                       // [4 waves][27 * C]
#pragma unroll
    for (int j = 0; j < NCB; ++j)
#pragma unroll
        for (int o = 0; o < 3; ++o)
#pragma unroll
            for (int k = 0; k < 9; ++k) red[wv * 27 * C + (o * C + 64 * j + lane) * 9 + k] = acc[j][o][k];
    if (lane == 0) { red[4 * 27 * C + wv * 4 + 0] = accb[0]; red[4 * 27 * C + wv * 4 + 1] = accb[1]; red[4 * 27 * C + wv * 4 + 2] = accb[2]; }
    __syncthreads();
    for (int e = t; e < 27 * C; e += 256)
        atomicAdd(dw + e, red[e] + red[27 * C + e] + red[2 * 27 * C + e] + red[3 * 27 * C + e]);
    if (db && t < 3) atomicAdd(db + t, red[4 * 27 * C + t] + red[4 * 27 * C + 4 + t] + red[4 * 27 * C + 8 + t] + red[4 * 27 * C + 12 + t]);
}

// First VGG19 layer (My_CR.py:65, features[0..1]): Conv2d(3 -> 64, 3x3, pad 1) + bias + ReLU from an NCHW image
// [B, 3, H, W] straight into the channel-blocked layout [B, 8, H, W, 8] of the Winograd stack - the library needs a
// convolution, a bias pass, a ReLU pass and a layout change for this.  Thin on the INPUT side: 27 MACs per output, lane <->
// pixel, wave pair <-> 32 of the 64 output channels, weights as LDS broadcasts, 512-byte runs per (channel group, row) out.
__global__ __launch_bounds__(256) void conv3x3_in3_blocked_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                  const float* __restrict__ bias, float* __restrict__ y,
                                                                  int H, int W, int tiles_x, int tiles_y, int relu) {
    __shared__ __attribute__((aligned(16))) float xs[3 * NPOS];        // [c][10 x 18]
    __shared__ __attribute__((aligned(16))) float ws[27 * 64];         // [c * 9 + tap][k]
    const int t = threadIdx.x;
    const int tx = blockIdx.x % tiles_x, ty = (blockIdx.x / tiles_x) % tiles_y, bimg = blockIdx.x / (tiles_x * tiles_y);
    for (int e = t; e < 27 * 64; e += 256) ws[(e % 27) * 64 + e / 27] = w[e];          // w[k][c][tap] -> ws[c*9+tap][k]
    for (int e = t; e < 3 * NPOS; e += 256) {
        const int c = e / NPOS, pos = e % NPOS;
        const int yy = ty * TH - 1 + pos / HW_, xx = tx * TW - 1 + pos % HW_;
        xs[e] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? x[(((size_t)bimg * 3 + c) * H + yy) * W + xx] : 0.f;
    }
    __syncthreads();
    const int half = t >> 7, pix = t & 127;
    const int py = pix / TW, px = pix % TW;
    f32x4 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = *reinterpret_cast<const f32x4*>(bias + half * 32 + 4 * j);
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const float xv = xs[c * NPOS + (py + tap / 3) * HW_ + px + tap % 3];
            const float* wr = ws + (c * 9 + tap) * 64 + half * 32;                    // wave-uniform: broadcast reads
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const f32x4 wv = *reinterpret_cast<const f32x4*>(wr + 4 * j);
                acc[j] += xv * wv;
            }
        }
    const int yy = ty * TH + py, xx = tx * TW + px;
    if (yy < H && xx < W) {
        const float lo = relu ? 0.f : -__builtin_inff();
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            f32x4 v = acc[j];
            v[0] = fmaxf(v[0], lo); v[1] = fmaxf(v[1], lo); v[2] = fmaxf(v[2], lo); v[3] = fmaxf(v[3], lo);
            const int k = half * 32 + 4 * j;                                           // channel group k / 8, offset k % 8
            *reinterpret_cast<f32x4*>(y + ((((size_t)bimg * 8 + k / 8) * H + yy) * W + xx) * 8 + (k & 7)) = v;
        }
    }
}

template <int C, typename T>
int launch_all(int which, const void* a, const float* b, const float* c, void* d, float* e, int B, int H, int W, hipStream_t s) {
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH, ntiles = B * tiles_x * tiles_y;
    const size_t smem = sizeof(ThinSmem<C>);
    if (which == 0) {            // a = tokens x (T), b = w, c = bias, d = image y (float)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&thin_conv_fwd_kernel<C, false, false, T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL((thin_conv_fwd_kernel<C, false, false, T>), dim3(ntiles), dim3(256), smem, s, (const T*)a, b, c, (float*)d, H, W, tiles_x, tiles_y);
    } else if (which == 1) {     // a = image gradient dy (float), b = w, d = token gradient dx (T)
        hipLaunchKernelGGL((thin_conv_dgrad_kernel<C, T>), dim3(ntiles), dim3(256), 0, s, (const float*)a, b, (T*)d, H, W, tiles_x, tiles_y);
    } else {                     // a = dy (float), b -> tokens x (T) passed through c's slot: see dispatch
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&thin_conv_wgrad_kernel<C, T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        const int cap = 3 * dhz_num_cus();
        const int grid = ntiles < cap ? ntiles : cap;
        hipLaunchKernelGGL((thin_conv_wgrad_kernel<C, T>), dim3(grid), dim3(256), smem, s, (const float*)a, (const T*)c, (float*)d, e, B, H, W, tiles_x, tiles_y);
    }
    return 0;
}

// which = 0: (x tokens, w, bias) -> y;  1: (dy, w) -> dx tokens;  2: (dy, x tokens via `c`) -> dw (d), db (e)
int dispatch(int which, int C, int dtype, const void* a, const float* b, const void* c, void* d, float* e, int B, int H, int W, hipStream_t s) {
    if (dtype == DHZ_F32) {
        if (C == 64) return launch_all<64, float>(which, a, b, (const float*)c, d, e, B, H, W, s);
        if (C == 128) return launch_all<128, float>(which, a, b, (const float*)c, d, e, B, H, W, s);
    } else if (dtype == DHZ_BF16) {
        if (C == 64) return launch_all<64, bf16s>(which, a, b, (const float*)c, d, e, B, H, W, s);
        if (C == 128) return launch_all<128, bf16s>(which, a, b, (const float*)c, d, e, B, H, W, s);
    }
    return 1;
}

}  // namespace

extern "C" int dhz_thin_conv3x3_fwd_dt(const void* x, const float* w, const float* bias, float* y, int B, int H, int W, int C,
                                       int dtype, void* stream) {
    DHZ_REQUIRE(x && w && y && B > 0 && H > 0 && W > 0, "dhz_thin_conv3x3_fwd: bad arguments");
    DHZ_REQUIRE(C == 64 || C == 128, "dhz_thin_conv3x3_fwd: C=%d unsupported (64, 128)", C);
    DHZ_REQUIRE(dtype == DHZ_F32 || dtype == DHZ_BF16, "dhz_thin_conv3x3_fwd: unknown dtype %d", dtype);
    dispatch(0, C, dtype, x, w, bias, y, nullptr, B, H, W, (hipStream_t)stream);
    DHZ_CHECK_LAUNCH("dhz_thin_conv3x3_fwd");
    return DHZ_OK;
}
extern "C" int dhz_thin_conv3x3_fwd(const float* x, const float* w, const float* bias, float* y, int B, int H, int W, int C,
                                    void* stream) {
    return dhz_thin_conv3x3_fwd_dt(x, w, bias, y, B, H, W, C, DHZ_F32, stream);
}

extern "C" int dhz_thin_conv3x3_dgrad_dt(const float* dy, const float* w, void* dx, int B, int H, int W, int C, int dtype,
                                         void* stream) {
    DHZ_REQUIRE(dy && w && dx && B > 0 && H > 0 && W > 0, "dhz_thin_conv3x3_dgrad: bad arguments");
    DHZ_REQUIRE(C == 64 || C == 128, "dhz_thin_conv3x3_dgrad: C=%d unsupported (64, 128)", C);
    DHZ_REQUIRE(dtype == DHZ_F32 || dtype == DHZ_BF16, "dhz_thin_conv3x3_dgrad: unknown dtype %d", dtype);
    dispatch(1, C, dtype, dy, w, nullptr, dx, nullptr, B, H, W, (hipStream_t)stream);
    DHZ_CHECK_LAUNCH("dhz_thin_conv3x3_dgrad");
    return DHZ_OK;
}
extern "C" int dhz_thin_conv3x3_dgrad(const float* dy, const float* w, float* dx, int B, int H, int W, int C, void* stream) {
    return dhz_thin_conv3x3_dgrad_dt(dy, w, dx, B, H, W, C, DHZ_F32, stream);
}

extern "C" int dhz_thin_conv3x3_wgrad_dt(const float* dy, const void* x, float* dw, float* db, int B, int H, int W, int C,
                                         int dtype, void* stream) {
    DHZ_REQUIRE(dy && x && dw && B > 0 && H > 0 && W > 0, "dhz_thin_conv3x3_wgrad: bad arguments");
    DHZ_REQUIRE(C == 64 || C == 128, "dhz_thin_conv3x3_wgrad: C=%d unsupported (64, 128)", C);
    DHZ_REQUIRE(dtype == DHZ_F32 || dtype == DHZ_BF16, "dhz_thin_conv3x3_wgrad: unknown dtype %d", dtype);
    dispatch(2, C, dtype, dy, nullptr, x, dw, db, B, H, W, (hipStream_t)stream);
    DHZ_CHECK_LAUNCH("dhz_thin_conv3x3_wgrad");
    return DHZ_OK;
}
extern "C" int dhz_thin_conv3x3_wgrad(const float* dy, const float* x, float* dw, float* db, int B, int H, int W, int C,
                                      void* stream) {
    return dhz_thin_conv3x3_wgrad_dt(dy, x, dw, db, B, H, W, C, DHZ_F32, stream);
}

extern "C" int dhz_thin_conv3x3_dgrad_blocked(const float* gb, const float* w, float* dx, int B, int H, int W, int C,
                                              void* stream) {
    DHZ_REQUIRE(gb && w && dx && B > 0 && H > 0 && W > 0, "dhz_thin_conv3x3_dgrad_blocked: bad arguments");
    DHZ_REQUIRE(C == 64, "dhz_thin_conv3x3_dgrad_blocked: C=%d unsupported (64)", C);
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH, ntiles = B * tiles_x * tiles_y;
    const size_t smem = sizeof(ThinSmem<64>);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&thin_conv_fwd_kernel<64, true, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL((thin_conv_fwd_kernel<64, true, true>), dim3(ntiles), dim3(256), smem, (hipStream_t)stream, gb, w, nullptr,
                       dx, H, W, tiles_x, tiles_y);
    DHZ_CHECK_LAUNCH("dhz_thin_conv3x3_dgrad_blocked");
    return DHZ_OK;
}

extern "C" int dhz_conv3x3_in3_blocked(const float* x, const float* w, const float* bias, float* y, int B, int H, int W,
                                       int K, int relu, void* stream) {
    DHZ_REQUIRE(x && w && bias && y && B > 0 && H > 0 && W > 0, "dhz_conv3x3_in3_blocked: bad arguments");
    DHZ_REQUIRE(K == 64, "dhz_conv3x3_in3_blocked: K=%d unsupported (64)", K);
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    hipLaunchKernelGGL(conv3x3_in3_blocked_kernel, dim3(B * tiles_x * tiles_y), dim3(256), 0, (hipStream_t)stream, x, w, bias, y,
                       H, W, tiles_x, tiles_y, relu);
    DHZ_CHECK_LAUNCH("dhz_conv3x3_in3_blocked");
    return DHZ_OK;
}
