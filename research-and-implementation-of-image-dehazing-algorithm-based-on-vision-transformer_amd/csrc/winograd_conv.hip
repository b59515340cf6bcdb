// K11 - 3x3 / stride 1 / pad 1 convolution of the VGG19 feature extractor behind the contrastive loss
// (My_CR.py:56-86), as Winograd F(2x2, 3x3) with the 16 transform-domain products on the fp32 matrix pipe.
//
// MIOpen serves these layers with a VALU Winograd kernel (miopenSp3AsmConv f2x3: ~50 TFLOP/s of real
// multiplies = ~110 TFLOP/s direct-conv equivalent, uniform over the layers - first profile); the same
// algorithm on v_mfma_f32_16x16x4_f32 moves the multiplies to the matrix pipe.
//
// Layout: channel-blocked NCHW8c - x[b][c/8][h][w][c%8] - so that a pixel's 8-channel group is 32 contiguous
// bytes and an image row of a group is one contiguous run (coalesced halo-patch loads, coalesced stores).
//
// One 256-thread workgroup = 64 output tiles (8x8 tiles = 16x16 pixels of one image) x 32 output channels.
// Loop over input-channel groups of 8:
//   1. the 18x18 halo patch of the group (optionally multiplied by the ReLU mask of a saved activation - the
//      backward-data pass) and the group's pre-transformed filters U[16][8][32] go to LDS;
//   2. input transform V = B^T d B: thread = (channel, tile), 16 LDS reads -> 16 LDS writes, V[xi][c][tile];
//   3. for each of the 16 transform positions xi: M_xi[32 x 64] += U_xi[32 x 8] V_xi[8 x 64] on MFMA; wave w owns
//      the 16 tiles 16w..16w+15 for ALL xi, so the inverse transform Y = A^T M A is lane-local in the end.
// Epilogue: inverse transform, + bias, optional ReLU, staging through LDS, coalesced 32-byte-group stores.
// The backward-data convolution is the same kernel with rotated / transposed filters (prepacked once: the VGG
// weights are frozen, My_CR.py:75-77).
#include "common.h"

namespace {

constexpr int TILES = 64;          // tiles per workgroup (8 x 8)
constexpr int KB = 32;             // output channels per workgroup
constexpr int CC = 8;              // input channels per step (= the layout's channel block)
constexpr int PR = 24;             // patch row stride (18 used): 2*PR % 64 == 48 keeps the b64 transform reads conflict-free
constexpr int PPL = 18 * PR + 8;   // patch plane stride (floats); 4*PPL % 64 == 32
constexpr int VROW = 160;          // V row [xi][channel pair]: 64 tiles x 2 channels + 32 pad (row stride % 64 == 32)
constexpr int YS = 260;            // output staging stride per output channel (16x16 px + pad)
constexpr int NP4 = (18 * 18 * 2 + 255) / 256;    // float4 patch loads per thread (3)
constexpr int NU4 = 16 * CC * KB / 4 / 256;       // float4 filter loads per thread (4)

struct WinoSmem {
    float patch[CC * PPL];         //  14.1 KB  planar [c][py][px]
    float u[16 * 4 * 64];          //  16.4 KB  [xi][channel pair g][k & 15][k >> 4][c & 1]  (= the prepacked order)
    float v[16 * 4 * VROW];        //  41.0 KB  [xi][channel pair][tile][c & 1]; later the output staging tile
};

// Pipeline per 8-channel group cb (2 barriers):
//   transform(patch -> V) ; filters(cb) registers -> LDS ; barrier ; issue global loads of group cb+1 (patch, mask,
//   filters -> registers) ; 64 MFMA per wave over U,V (b128 / b64 fragment reads) ; patch(cb+1) registers -> LDS ; barrier
// so the global-load latency of the next group hides behind the MFMA phase of the current one.
// MFMA contraction mapping: step s in {0,1}, lane group g  <->  channel 2g+s (any bijection works as long as A and B agree),
// which makes a lane's two B values (and its four A values) contiguous in LDS.
template <bool RELU, bool MASKED>
__global__ __launch_bounds__(256, 2) void winograd_conv3x3_kernel(const float* __restrict__ x,
                                                              const float* __restrict__ act_mask,
                                                              const float* __restrict__ upack,
                                                              const float* __restrict__ bias, float* __restrict__ y,
                                                              int H, int W, int C, int K) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    WinoSmem& sm = *reinterpret_cast<WinoSmem*>(smem_raw);
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int i16 = lane & 15, g = lane >> 4;
    const int KBn = K / KB, CBn = C / CC;
    const int kb = blockIdx.x % KBn;
    const int blk = blockIdx.x / KBn;
    const int bx_n = W / 16, by_n = H / 16;
    const int bimg = blk / (bx_n * by_n);
    const int by = (blk / bx_n) % by_n, bx = blk % bx_n;
    const int oy0 = by * 16, ox0 = bx * 16;                 // output block origin; patch origin is (oy0-1, ox0-1)
    const size_t plane = (size_t)H * W * 8;                 // floats per (image, channel-group) plane

    f32x4 acc[16][2];
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) { acc[xi][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[xi][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    // per-thread patch slots: element e = t + 256 i -> (pixel p = e >> 1, channel half e & 1).  The loads are
    // unconditional (out-of-image / unused slots read offset 0 and are zeroed or skipped when written to LDS): a
    // predicated load would make the compiler wait for it right away instead of across the MFMA phase.
    int poff[NP4];            // offset inside a plane (floats)
    int pdst[NP4];            // LDS destination, -1 = unused slot
    bool pin[NP4];            // inside the image
#pragma unroll
    for (int i = 0; i < NP4; ++i) {
        const int e = t + 256 * i;
        const int half = e & 1, p = e >> 1;
        const int py = p / 18, px = p % 18;
        const int iy = oy0 - 1 + py, ix = ox0 - 1 + px;
        pin[i] = e < 18 * 18 * 2 && iy >= 0 && iy < H && ix >= 0 && ix < W;
        poff[i] = pin[i] ? (iy * W + ix) * 8 + half * 4 : 0;
        pdst[i] = e < 18 * 18 * 2 ? (half * 4) * PPL + py * PR + px : -1;
    }
    float4 rp[NP4], rm[NP4];
    float4 ru0, ru1, ru2, ru3;                          // NU4 == 4 filter float4 per thread
    static_assert(NU4 == 4, "filter prefetch registers");
    auto gload = [&](int cb) {
        const float* xp = x + ((size_t)bimg * CBn + cb) * plane;
        const float* mp = MASKED ? act_mask + ((size_t)bimg * CBn + cb) * plane : nullptr;
#pragma unroll
        for (int i = 0; i < NP4; ++i) {
            rp[i] = *reinterpret_cast<const float4*>(xp + poff[i]);
            if (MASKED) rm[i] = *reinterpret_cast<const float4*>(mp + poff[i]);
        }
        const float4* up = reinterpret_cast<const float4*>(upack + ((size_t)kb * CBn + cb) * (16 * CC * KB));
        ru0 = up[t]; ru1 = up[t + 256]; ru2 = up[t + 512]; ru3 = up[t + 768];
    };
    auto write_patch = [&]() {
#pragma unroll
        for (int i = 0; i < NP4; ++i) {
            if (pdst[i] < 0) continue;
            float4 val = rp[i];
            if (MASKED) {
                val.x = rm[i].x > 0.f ? val.x : 0.f; val.y = rm[i].y > 0.f ? val.y : 0.f;
                val.z = rm[i].z > 0.f ? val.z : 0.f; val.w = rm[i].w > 0.f ? val.w : 0.f;
            }
            if (!pin[i]) val = make_float4(0.f, 0.f, 0.f, 0.f);
            float* dst = sm.patch + pdst[i];
            dst[0] = val.x; dst[PPL] = val.y; dst[2 * PPL] = val.z; dst[3 * PPL] = val.w;
        }
    };
    auto write_u = [&]() {
        float4* us = reinterpret_cast<float4*>(sm.u);
        us[t] = ru0; us[t + 256] = ru1; us[t + 512] = ru2; us[t + 768] = ru3;
    };

    float bk[2][4];                                      // bias of this lane's 8 output channels (16 tr + 4 g + r)
#pragma unroll
    for (int tr = 0; tr < 2; ++tr)
#pragma unroll
        for (int r = 0; r < 4; ++r) bk[tr][r] = bias ? bias[kb * KB + 16 * tr + 4 * g + r] : 0.f;

    gload(0);
    write_patch();
    __syncthreads();

    for (int cb = 0; cb < CBn; ++cb) {
        // ---- input transform V = B^T d B ; thread = (channel c, tile): 2 tiles per thread
        {
            const int c = t >> 5;
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                const int tile = (t & 31) + 32 * pass;
                const int ty = tile >> 3, tx = tile & 7;
                const float* pp = sm.patch + c * PPL + (2 * ty) * PR + 2 * tx;
                float d[4][4];
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const float2 lo = *reinterpret_cast<const float2*>(pp + a * PR);
                    const float2 hi = *reinterpret_cast<const float2*>(pp + a * PR + 2);
                    d[a][0] = lo.x; d[a][1] = lo.y; d[a][2] = hi.x; d[a][3] = hi.y;
                }
                float tmp[4][4];                    // B^T d
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    tmp[0][b] = d[0][b] - d[2][b];
                    tmp[1][b] = d[1][b] + d[2][b];
                    tmp[2][b] = d[2][b] - d[1][b];
                    tmp[3][b] = d[1][b] - d[3][b];
                }
                float* vp = sm.v + (c >> 1) * VROW + tile * 2 + (c & 1);
#pragma unroll
                for (int a = 0; a < 4; ++a) {       // (B^T d) B
                    vp[(4 * a + 0) * 4 * VROW] = tmp[a][0] - tmp[a][2];
                    vp[(4 * a + 1) * 4 * VROW] = tmp[a][1] + tmp[a][2];
                    vp[(4 * a + 2) * 4 * VROW] = tmp[a][2] - tmp[a][1];
                    vp[(4 * a + 3) * 4 * VROW] = tmp[a][1] - tmp[a][3];
                }
            }
        }
        write_u();                                   // filters of this group (loaded one MFMA phase ago)
        __syncthreads();
        if (cb + 1 < CBn) gload(cb + 1);
        // ---- M_xi += U_xi V_xi for the wave's 16 tiles, all 16 xi, both 16-row halves of the 32 output channels
        {
            const float* up = sm.u + (g * 16 + i16) * 4;
            const float* vp = sm.v + g * VROW + (16 * w + i16) * 2;
            // fragments of position xi+1 are requested before the 4 MFMAs of xi are issued (two register sets)
            float4 a = *reinterpret_cast<const float4*>(up);                                // {k lo,c0},{k lo,c1},{k hi,c0},{k hi,c1}
            float2 b = *reinterpret_cast<const float2*>(vp);                                // {c0},{c1}
#pragma unroll
            for (int xi = 0; xi < 16; ++xi) {
                float4 an = a; float2 bn = b;
                if (xi + 1 < 16) {
                    an = *reinterpret_cast<const float4*>(up + (xi + 1) * 256);
                    bn = *reinterpret_cast<const float2*>(vp + (xi + 1) * 4 * VROW);
                }
                acc[xi][0] = mfma16(a.x, b.x, acc[xi][0]);
                acc[xi][1] = mfma16(a.z, b.x, acc[xi][1]);
                acc[xi][0] = mfma16(a.y, b.y, acc[xi][0]);
                acc[xi][1] = mfma16(a.w, b.y, acc[xi][1]);
                a = an; b = bn;
            }
        }
        if (cb + 1 < CBn) write_patch();
        __syncthreads();
    }

    // ---- epilogue: Y = A^T M A (lane-local), + bias, ReLU; stage [32 k][16 x 16 px] in LDS; coalesced stores
    float* ys = sm.v;                                   // [k 32][py 16][px 16] stride YS
    {
        const int tile = 16 * w + i16, ty = tile >> 3, tx = tile & 7;
#pragma unroll
        for (int tr = 0; tr < 2; ++tr)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = 16 * tr + 4 * g + r;
                float m[16];
#pragma unroll
                for (int xi = 0; xi < 16; ++xi) m[xi] = acc[xi][tr][r];
                float t0[4], t1[4];
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    t0[b] = m[b] + m[4 + b] + m[8 + b];
                    t1[b] = m[4 + b] - m[8 + b] - m[12 + b];
                }
                const float bv = bk[tr][r];
                float y00 = t0[0] + t0[1] + t0[2] + bv, y01 = t0[1] - t0[2] - t0[3] + bv;
                float y10 = t1[0] + t1[1] + t1[2] + bv, y11 = t1[1] - t1[2] - t1[3] + bv;
                if (RELU) { y00 = fmaxf(y00, 0.f); y01 = fmaxf(y01, 0.f); y10 = fmaxf(y10, 0.f); y11 = fmaxf(y11, 0.f); }
                float* o = ys + k * YS + (2 * ty) * 16 + 2 * tx;
                *reinterpret_cast<float2*>(o) = make_float2(y00, y01);
                *reinterpret_cast<float2*>(o + 16) = make_float2(y10, y11);
            }
    }
    __syncthreads();
    {
        // output group kg (4 per workgroup) : y[b][kb*4 + kg][oy][ox][8]; one float4 = half a pixel group
        const int KG = K / 8;
        for (int e = t; e < 4 * 256 * 2; e += 256) {
            const int half = e & 1, p = (e >> 1) & 255, kg = e >> 9;
            const int py = p >> 4, px = p & 15;
            const float* s0 = ys + (kg * 8 + half * 4) * YS + p;
            const float4 v4 = make_float4(s0[0], s0[YS], s0[2 * YS], s0[3 * YS]);
            *reinterpret_cast<float4*>(y + (((size_t)bimg * KG + kb * 4 + kg) * H + oy0 + py) * W * 8 +
                                       (size_t)(ox0 + px) * 8 + half * 4) = v4;
        }
    }
}

// U = G g G^T for every (k, c), packed [k/32][c/8][xi 16][(c%8)/2][k%16][(k%32)/16][c%2] (the kernel's LDS order);
// `transposed_rot` selects the backward-data
// filters g'[c][k][i][j] = g[k][c][2-i][2-j] (the roles of C and K swap).
__global__ void winograd_prepack_kernel(const float* __restrict__ wgt, float* __restrict__ upack, int Kout, int Cin,
                                        int transposed_rot) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;      // over Kout * Cin
    if (e >= Kout * Cin) return;
    const int k = e / Cin, c = e % Cin;
    float gk[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            gk[i][j] = transposed_rot ? wgt[((size_t)c * Kout + k) * 9 + (2 - i) * 3 + (2 - j)]      // wgt is [Cin][Kout][3][3]
                                      : wgt[((size_t)k * Cin + c) * 9 + i * 3 + j];               // wgt is [Kout][Cin][3][3]
    float tg[4][3];                                           // G g
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        tg[0][j] = gk[0][j];
        tg[1][j] = 0.5f * (gk[0][j] + gk[1][j] + gk[2][j]);
        tg[2][j] = 0.5f * (gk[0][j] - gk[1][j] + gk[2][j]);
        tg[3][j] = gk[2][j];
    }
    const int kk = k % KB, cc = c % CC;
    float* dst = upack + (((size_t)(k / KB) * (Cin / CC) + c / CC) * 16) * (CC * KB) +
                 (((cc >> 1) * 16 + (kk & 15)) * 4 + (kk >> 4) * 2 + (cc & 1));
#pragma unroll
    for (int i = 0; i < 4; ++i) {                             // (G g) G^T
        dst[(4 * i + 0) * CC * KB] = tg[i][0];
        dst[(4 * i + 1) * CC * KB] = 0.5f * (tg[i][0] + tg[i][1] + tg[i][2]);
        dst[(4 * i + 2) * CC * KB] = 0.5f * (tg[i][0] - tg[i][1] + tg[i][2]);
        dst[(4 * i + 3) * CC * KB] = tg[i][2];
    }
}

// NCHW <-> NCHW8c (blocked) layout conversion
__global__ void nchw_to_blocked_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int C, int HW,
                                       int to_blocked) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)B * C * HW;
    if (e >= total) return;
    // e indexes the blocked tensor [b][c/8][hw][8]
    const int c8 = e & 7;
    const size_t rest = e >> 3;
    const int hw = rest % HW;
    const size_t r2 = rest / HW;
    const int cg = r2 % (C / 8), b = r2 / (C / 8);
    const size_t plain = ((size_t)b * C + cg * 8 + c8) * HW + hw;
    if (to_blocked) dst[e] = src[plain]; else dst[plain] = src[e];
}

}  // namespace

extern "C" int dhz_winograd_prepack(const float* weight, float* upack, int Kout, int Cin, int transposed_rot,
                                    void* stream) {
    DHZ_REQUIRE(weight && upack && Kout % KB == 0 && Cin % CC == 0, "dhz_winograd_prepack: Kout=%d Cin=%d", Kout, Cin);
    const int n = Kout * Cin;
    hipLaunchKernelGGL(winograd_prepack_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, weight, upack,
                       Kout, Cin, transposed_rot);
    DHZ_CHECK_LAUNCH("dhz_winograd_prepack");
    return DHZ_OK;
}

extern "C" int dhz_winograd_conv3x3(const float* x, const float* act_mask, const float* upack, const float* bias,
                                    float* y, int B, int H, int W, int C, int K, int relu, void* stream) {
    DHZ_REQUIRE(x && upack && y, "dhz_winograd_conv3x3: null pointer");
    DHZ_REQUIRE(B > 0 && H % 16 == 0 && W % 16 == 0 && C % CC == 0 && K % KB == 0,
                "dhz_winograd_conv3x3: unsupported shape B=%d H=%d W=%d C=%d K=%d", B, H, W, C, K);
    const int grid = B * (H / 16) * (W / 16) * (K / KB);
    const size_t smem = sizeof(WinoSmem);
    hipStream_t s = (hipStream_t)stream;
#define GO(R, M)                                                                                                   \
    do {                                                                                                           \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&winograd_conv3x3_kernel<R, M>),                   \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);                          \
        hipLaunchKernelGGL((winograd_conv3x3_kernel<R, M>), dim3(grid), dim3(256), smem, s, x, act_mask, upack, bias, \
                           y, H, W, C, K);                                                                         \
    } while (0)
    if (relu) { if (act_mask) GO(true, true); else GO(true, false); }
    else { if (act_mask) GO(false, true); else GO(false, false); }
#undef GO
    DHZ_CHECK_LAUNCH("dhz_winograd_conv3x3");
    return DHZ_OK;
}

extern "C" int dhz_layout_blocked8(const float* src, float* dst, int B, int C, int HW, int to_blocked, void* stream) {
    DHZ_REQUIRE(src && dst && C % 8 == 0 && B > 0 && HW > 0, "dhz_layout_blocked8: bad arguments");
    const size_t n = (size_t)B * C * HW;
    hipLaunchKernelGGL(nchw_to_blocked_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src,
                       dst, B, C, HW, to_blocked);
    DHZ_CHECK_LAUNCH("dhz_layout_blocked8");
    return DHZ_OK;
}
