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
// One 512-thread workgroup = two 16x16-pixel output blocks (waves 0-3 / 4-7) x 32 output channels.  Wave w owns 16
// tiles (4 x 16 output pixels) of its block for ALL 16 transform positions xi, so
//   * its halo patch (6 x 18 px x 8 ch) and its transformed input V live in a wave-private LDS region: patch
//     load, input transform and MFMA of a wave need no workgroup barrier, and the inverse transform Y = A^T M A
//     and the output staging are lane- / wave-local;
//   * only the pre-transformed filters U (16 KB per 8-channel group, shared by the 8 waves) go through a
//     workgroup-wide ring of 3 LDS buffers: ONE barrier per channel group.
// Per channel group cb a wave issues 64 MFMAs (M_xi[32 k x 16 tiles] += U_xi[32 x 8] V_xi[8 x 16], 16 xi).  Measured
// on gfx950: VALU/LDS instructions of the *partner* wave on a SIMD get ~1 issue slot per 20 cycles beside an fp32
// MFMA stream (a separate transform phase - in the same wave or ping-ponged against the partner - ran the matrix
// pipe at 56 %), while a wave's OWN independent instructions between its MFMAs are nearly free.  So the whole
// input transform of group cb+1 (patch registers -> LDS, next global loads, patch reads, B^T d B in packed fp32,
// filters -> LDS) is spread over the 32 gaps between MFMA pairs of group cb, its 16 x float2 results are held in
// registers and stored to V once the last MFMA of cb has read V.  The barrier sits at a different position of
// the stream in the two halves (pair-step 1 / 5), which staggers the SIMD partners by half a group so that one
// wave's V-store burst falls into the other's MFMA stream.
// MFMA contraction mapping: step s in {0,1}, lane group g  <->  channel 2g+s (any bijection works as long as A and
// B agree), which makes a lane's two B values (and its four A values) contiguous in LDS (b64 / b128 reads).
// The backward-data convolution is the same kernel with rotated / transposed filters (prepacked once: the VGG
// weights are frozen, My_CR.py:75-77) and the ReLU mask of the saved activation applied to the patch.
#include <stdlib.h>
#include <type_traits>
#include "common.h"

#ifndef WINO_ABL
#define WINO_ABL 0      // timing diagnostics (tools/variants.sh): 1 no input transform, 2 no global loads in the loop, 4 no MFMAs, 8 no V / filter LDS traffic of the next group
#endif

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int KB = 32;             // output channels per workgroup
constexpr int CC = 8;              // input channels per step (= the layout's channel block)
constexpr int RS = 48;             // patch row stride (floats): 18 px x 2 channels used; 2*RS % 64 == 32
constexpr int PPL = 320;           // patch plane (channel pair) stride: 6 rows x 48 = 288 used; % 64 == 0
constexpr int UF = 16 * CC * KB;   // floats of one (kb, cb) filter slice (4096)
constexpr int VW = 16 * 4 * 32;    // floats of one wave's V: [xi][channel pair][16 tiles x 2 channels]
constexpr int WAVE_LDS = VW + 4 * PPL;    // V + patch, contiguous per wave (3328 floats = 13 KB)
constexpr int YS = 68;             // output staging stride per output channel (4 x 16 px + pad); 32 * YS <= WAVE_LDS
constexpr int NP4 = 4;             // float4 patch slots per lane: 6 x 18 px x 2 halves = 216 of 256
constexpr int NUBUF = 3;           // filter ring
constexpr size_t WINO_SMEM = (size_t)(NUBUF * UF + 8 * WAVE_LDS + KB) * sizeof(float);   // 152 KB: one workgroup per CU

__device__ __forceinline__ void wave_sync() {
    // LDS operations of one wave execute in order; only the compiler must not move accesses across phase boundaries
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Epilogue: y = conv(x) [+ bias] [ReLU]  (forward)   or   y = mask > 0 ? conv(x) + addend : 0  (backward data: `mask` is
// the saved post-ReLU activation at the OUTPUT positions, i.e. the ReLU of the layer below, `addend` the gradient that
// reaches that activation from a loss tap) - so the hot loop is the same for both passes.
//
// Q8: 8 x 8 maps (the last VGG layer of the loss, conv5_1).  FOUR images form one virtual 16 x 16 block (2 x 2 arrangement): the
// patch offsets carry the image of every pixel, the 4 x 4 input window of a tile is masked where it reaches into the
// neighbouring image (that is where the tile's own image has its zero padding), and the stores scatter back per image.
template <bool FWD, bool Q8>
__global__ __launch_bounds__(512) void winograd_conv3x3_kernel(const float* __restrict__ x,
                                                           const float* __restrict__ upack,
                                                           const float* __restrict__ bias, int relu,
                                                           const float* __restrict__ out_mask,
                                                           const float* __restrict__ out_addend,
                                                           float* __restrict__ y, int H, int W, int C, int K, int nblk, int xcd_group,
                                                           int nimg) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, half = w >> 2, wl = w & 3;
    const int i16 = lane & 15, g = lane >> 4;
    float* us = smem;                                   // filters [NUBUF][UF]: [xi][channel pair][k & 15][k >> 4][c & 1]
    float* vw = smem + NUBUF * UF + w * WAVE_LDS;       // this wave's V
    float* pw = vw + VW;                                // this wave's patch [channel pair][6][RS]: (px, c & 1) interleaved
    const int KBn = K / KB, CBn = C / CC;
    // Workgroups are dealt round-robin over the 8 XCDs (private L2s).  xcd_group != 0: renumber so that consecutive
    // logical ids - the K/32 output-channel blocks of one spatial block pair, which read the same input patches - run on
    // the same XCD and the patch comes from HBM once instead of once per XCD.
    int lid = blockIdx.x;
    if (xcd_group) lid = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int kb = lid % KBn;
    int blk = (lid / KBn) * 2 + half;
    const bool live = blk < nblk;                       // odd block count: the last workgroup's second half recomputes, stores nothing
    if (!live) blk = nblk - 1;
    const int bx_n = Q8 ? 1 : W / 16, by_n = Q8 ? 1 : H / 16;
    const int bimg = Q8 ? 0 : blk / (bx_n * by_n);              // Q8: the image is a property of the pixel (below)
    const int by = (blk / bx_n) % by_n, bx = blk % bx_n;
    const int wy0 = by * 16 + 4 * wl, ox0 = bx * 16;        // the wave's output rows wy0..wy0+3; patch origin (wy0-1, ox0-1)
    const size_t plane = (size_t)H * W * 8;                 // floats per (image, channel-group) plane
    const int pbar = half ? 5 : 1;                          // pair-step that carries the workgroup barrier

    f32x4 acc[16][2];
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) { acc[xi][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[xi][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    // per-lane patch slots: element e = lane + 64 i -> (pixel p = e >> 1 of the 6 x 18 patch, channel half e & 1).
    // Out-of-image pixels are zero padding: their LDS slots are zeroed once below and never written again; the
    // global loads stay unconditional (such slots read offset 0) so that no load is waited for at its issue point.
    unsigned poff[NP4];       // offset inside a plane (floats)
    int pdst[NP4];            // LDS destination; unused slots and zero padding go to the pad words behind each plane
    static_assert(PPL - 6 * RS >= 32, "dump words");
#pragma unroll
    for (int i = 0; i < NP4; ++i) {
        const int e = lane + 64 * i;
        const int hf = e & 1, p = e >> 1;
        const int py = p / 18, px = p % 18;
        const int iy = wy0 - 1 + py, ix = ox0 - 1 + px;
        const bool in = e < 6 * 18 * 2 && iy >= 0 && iy < (Q8 ? 16 : H) && ix >= 0 && ix < (Q8 ? 16 : W);
        if (Q8) {
            const int img = min(4 * blk + 2 * (iy >> 3) + (ix >> 3), nimg - 1);
            poff[i] = in ? (unsigned)(img * (C / CC) * 512 + ((iy & 7) * 8 + (ix & 7)) * 8 + hf * 4) : 0u;
        } else
            poff[i] = in ? (unsigned)((iy * W + ix) * 8 + hf * 4) : 0u;
        pdst[i] = (2 * hf) * PPL + (in ? py * RS + px * 2 : 6 * RS + (lane & 15) * 2);
    }
    {
        float4* z = reinterpret_cast<float4*>(pw);
#pragma unroll
        for (int i = 0; i < 4 * PPL / 4 / 64; ++i) z[lane + 64 * i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float4 rp[NP4];
    float4 ru0, ru1;                                     // UF / 4 / 512 = 2 filter float4 per thread
    const float* xbase = x + (size_t)bimg * CBn * plane;
    const float4* ubase = reinterpret_cast<const float4*>(upack + (size_t)kb * CBn * UF) + t;
    auto gload_patch = [&](int cb) {
        const float* xp = xbase + (size_t)cb * plane;
#pragma unroll
        for (int i = 0; i < NP4; ++i) rp[i] = *reinterpret_cast<const float4*>(xp + poff[i]);
    };
    auto gload_u = [&](int cb) {
        const float4* up = ubase + (size_t)cb * (UF / 4);
        ru0 = up[0]; ru1 = up[512];
    };
    auto write_patch_slot = [&](int i) {
        const float4 val = rp[i];
        float* dst = pw + pdst[i];
        *reinterpret_cast<float2*>(dst) = make_float2(val.x, val.y);
        *reinterpret_cast<float2*>(dst + PPL) = make_float2(val.z, val.w);
    };
    auto write_u = [&](int buf) {
        float4* ud = reinterpret_cast<float4*>(us + buf * UF);
        ud[t] = ru0; ud[t + 512] = ru1;
    };

    // ---- input transform of the lane's (channel pair g, tile i16), both channels packed: d rows -> B^T d -> (B^T d) B
    f32x2 d[4][4];                                       // patch values, then B^T d, then the 16 V values (in place)
    const float* pread = pw + g * PPL + (2 * (i16 >> 3)) * RS + (i16 & 7) * 4;
    // Q8: window row 0 / 3 (column 0 / 3) of a tile that starts / ends an 8-pixel image belongs to the neighbouring image
    float fr0 = 1.f, fr3 = 1.f, fc0 = 1.f, fc3 = 1.f;
    if (Q8) {
        const int r0 = 4 * wl + 2 * (i16 >> 3), c0 = 2 * (i16 & 7);
        fr0 = (r0 & 7) == 0 ? 0.f : 1.f; fr3 = ((r0 + 2) & 7) == 0 ? 0.f : 1.f;
        fc0 = (c0 & 7) == 0 ? 0.f : 1.f; fc3 = ((c0 + 2) & 7) == 0 ? 0.f : 1.f;
    }
    auto read_row = [&](int a) {
        const float4 q0 = *reinterpret_cast<const float4*>(pread + a * RS);       // px 0,1 x (c0,c1)
        const float4 q1 = *reinterpret_cast<const float4*>(pread + a * RS + 4);   // px 2,3
        d[a][0] = f32x2{q0.x, q0.y}; d[a][1] = f32x2{q0.z, q0.w};
        d[a][2] = f32x2{q1.x, q1.y}; d[a][3] = f32x2{q1.z, q1.w};
        if (Q8) {
            const float fr = a == 0 ? fr0 : a == 3 ? fr3 : 1.f;
            d[a][0] *= fr * fc0; d[a][3] *= fr * fc3;
            if (a == 0 || a == 3) { d[a][1] *= fr; d[a][2] *= fr; }
        }
    };
    auto col_transform = [&](int b) {                    // B^T d, column b
        const f32x2 d0 = d[0][b], d1 = d[1][b], d2 = d[2][b], d3 = d[3][b];
        d[0][b] = d0 - d2; d[1][b] = d1 + d2; d[2][b] = d2 - d1; d[3][b] = d1 - d3;
    };
    auto row_transform = [&](int a) {                    // (B^T d) B, row a
        const f32x2 t0 = d[a][0], t1 = d[a][1], t2 = d[a][2], t3 = d[a][3];
        d[a][0] = t0 - t2; d[a][1] = t1 + t2; d[a][2] = t2 - t1; d[a][3] = t1 - t3;
    };
    float* vdst = vw + g * 32 + i16 * 2;
    auto write_v2 = [&](int xi) {                        // two transform positions xi, xi+1 (one row a, columns b, b+1)
        const int a = xi >> 2, b = xi & 3;
        *reinterpret_cast<float2*>(vdst + xi * 128) = make_float2(d[a][b].x, d[a][b].y);
        *reinterpret_cast<float2*>(vdst + (xi + 1) * 128) = make_float2(d[a][b + 1].x, d[a][b + 1].y);
    };
    auto write_v = [&]() {
#pragma unroll
        for (int xi = 0; xi < 16; xi += 2) write_v2(xi);
    };

    float* bias_s = smem + NUBUF * UF + 8 * WAVE_LDS;    // the 32 biases of this output-channel block
    if (FWD && t < KB) bias_s[t] = bias ? bias[kb * KB + t] : 0.f;

    // ---- prologue: group 0 transformed and in V, filters 0 in ring slot 0, loads of group 1 in flight
    gload_patch(0);
    gload_u(0);
    wave_sync();                                         // zero fill before the patch stores
#pragma unroll
    for (int i = 0; i < NP4; ++i) write_patch_slot(i);
    write_u(0);
    { const int c1 = CBn > 1 ? 1 : 0; gload_patch(c1); gload_u(c1); }
    wave_sync();
#pragma unroll
    for (int a = 0; a < 4; ++a) read_row(a);
#pragma unroll
    for (int b = 0; b < 4; ++b) col_transform(b);
#pragma unroll
    for (int a = 0; a < 4; ++a) row_transform(a);
    write_v();
    __syncthreads();

    const float* ufrag = us + (g * 16 + i16) * 4;
    const float* vfrag = vw + g * 32 + i16 * 2;
    // one channel group: 64 MFMAs with (WITH_NEXT) the 32 work slots of group cb+1 in the gaps between MFMA pairs
    auto group = [&](int cb, auto with_next) {
        constexpr bool WITH_NEXT = decltype(with_next)::value;
        const float* up = ufrag + (cb % NUBUF) * UF;
        const int nbuf = (cb + 1) % NUBUF;
        const int cb2 = cb + 2 < CBn ? cb + 2 : CBn - 1;      // loads past the last group re-read it (never stored to LDS)
        auto slot = [&](int sidx) {
            if (!WITH_NEXT) return;
            if ((WINO_ABL & 1) && sidx >= 13 && sidx <= 21) return;
            if ((WINO_ABL & 2) && (sidx == 1 || sidx == 6)) { if (sidx == 6) wave_sync(); return; }
            if ((WINO_ABL & 8) && (sidx == 0 || (sidx >= 2 && sidx <= 5) || (sidx >= 8 && sidx <= 11) || sidx >= 22)) return;
            switch (sidx) {
                case 0: write_u(nbuf); break;                                  // loaded during the previous group
                case 1: gload_u(cb2); break;
                case 2: write_patch_slot(0); break;
                case 3: write_patch_slot(1); break;
                case 4: write_patch_slot(2); break;
                case 5: write_patch_slot(3); break;
                case 6: wave_sync(); gload_patch(cb2); break;
                case 8: read_row(0); break;
                case 9: read_row(1); break;
                case 10: read_row(2); break;
                case 11: read_row(3); break;
                case 13: col_transform(0); break;
                case 14: col_transform(1); break;
                case 15: col_transform(2); break;
                case 16: col_transform(3); break;
                case 18: row_transform(0); break;
                case 19: row_transform(1); break;
                case 20: row_transform(2); break;
                case 21: row_transform(3); break;
                // V rows of positions the MFMA stream is done with are overwritten early (LDS executes a wave's operations in
                // order and the reads of pair-steps <= 3 / <= 5 were issued before slot 16 / 24): only 4 of the 16 V stores
                // are left for the burst between two groups
                case 22: write_v2(0); break;
                case 23: write_v2(2); break;
                case 24: write_v2(4); break;
                case 25: write_v2(6); break;
                case 26: write_v2(8); break;
                case 27: write_v2(10); break;
                default: break;
            }
        };
        // fragments of the next PAIR of positions are requested before the 8 MFMAs of this pair are issued
        float4 a0 = *reinterpret_cast<const float4*>(up);                       // {k lo,c0},{k lo,c1},{k hi,c0},{k hi,c1}
        float4 a1 = *reinterpret_cast<const float4*>(up + 256);
        float2 b0 = *reinterpret_cast<const float2*>(vfrag);                    // {c0},{c1}
        float2 b1 = *reinterpret_cast<const float2*>(vfrag + 128);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int xi = 2 * j;
            float4 a0n = a0, a1n = a1; float2 b0n = b0, b1n = b1;
            if (j + 1 < 8) {
                a0n = *reinterpret_cast<const float4*>(up + (xi + 2) * 256);
                a1n = *reinterpret_cast<const float4*>(up + (xi + 3) * 256);
                b0n = *reinterpret_cast<const float2*>(vfrag + (xi + 2) * 128);
                b1n = *reinterpret_cast<const float2*>(vfrag + (xi + 3) * 128);
            }
            if (j == 1 || j == 5) { if (j == pbar) __syncthreads(); }
            __builtin_amdgcn_sched_barrier(0);
            if (!(WINO_ABL & 4)) {
            acc[xi][0] = mfma16(a0.x, b0.x, acc[xi][0]);
            acc[xi][1] = mfma16(a0.z, b0.x, acc[xi][1]); } else { acc[xi][0][0] += a0.x + b0.x; }
            __builtin_amdgcn_sched_barrier(0);
            slot(4 * j + 0);
            __builtin_amdgcn_sched_barrier(0);
            if (!(WINO_ABL & 4)) {
            acc[xi + 1][0] = mfma16(a1.x, b1.x, acc[xi + 1][0]);
            acc[xi + 1][1] = mfma16(a1.z, b1.x, acc[xi + 1][1]); } else { acc[xi + 1][0][0] += a1.x + b1.x; }
            __builtin_amdgcn_sched_barrier(0);
            slot(4 * j + 1);
            __builtin_amdgcn_sched_barrier(0);
            if (!(WINO_ABL & 4)) {
            acc[xi][0] = mfma16(a0.y, b0.y, acc[xi][0]);
            acc[xi][1] = mfma16(a0.w, b0.y, acc[xi][1]); } else { acc[xi][1][0] += a0.y + b0.y + a0.z + a0.w; }
            __builtin_amdgcn_sched_barrier(0);
            slot(4 * j + 2);
            __builtin_amdgcn_sched_barrier(0);
            if (!(WINO_ABL & 4)) {
            acc[xi + 1][0] = mfma16(a1.y, b1.y, acc[xi + 1][0]);
            acc[xi + 1][1] = mfma16(a1.w, b1.y, acc[xi + 1][1]); } else { acc[xi + 1][1][0] += a1.y + b1.y + a1.z + a1.w; }
            __builtin_amdgcn_sched_barrier(0);
            slot(4 * j + 3);
            __builtin_amdgcn_sched_barrier(0);
            a0 = a0n; a1 = a1n; b0 = b0n; b1 = b1n;
        }
        wave_sync();                                     // every MFMA of this group has read V
        if (WITH_NEXT && !(WINO_ABL & 8)) { write_v2(12); write_v2(14); }
        wave_sync();
    };
    for (int cb = 0; cb + 1 < CBn; ++cb) group(cb, std::true_type{});
    group(CBn - 1, std::false_type{});

    // ---- epilogue, wave-local: Y = A^T M A (lane-local), + bias, ReLU -> staging [32 k][4 x 16 px] over the wave's own
    //      V/patch region; then 32-byte-group stores, 512 contiguous bytes per (channel group, row)
    float* ys = vw;
    {
        const int ty = i16 >> 3, tx = i16 & 7;
        const float lo = relu ? 0.f : -__builtin_inff();     // ReLU as a lower clamp: no branch per output
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
                const float bv = FWD ? bias_s[k] : 0.f;
                float y00 = t0[0] + t0[1] + t0[2] + bv, y01 = t0[1] - t0[2] - t0[3] + bv;
                float y10 = t1[0] + t1[1] + t1[2] + bv, y11 = t1[1] - t1[2] - t1[3] + bv;
                if (FWD) { y00 = fmaxf(y00, lo); y01 = fmaxf(y01, lo); y10 = fmaxf(y10, lo); y11 = fmaxf(y11, lo); }
                float* o = ys + k * YS + (2 * ty) * 16 + 2 * tx;
                *reinterpret_cast<float2*>(o) = make_float2(y00, y01);
                *reinterpret_cast<float2*>(o + 16) = make_float2(y10, y11);
            }
    }
    wave_sync();
    if (live) {
        const int KG = K / 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = lane + 64 * i;
            const int hf = e & 1, px = (e >> 1) & 15, py = (e >> 5) & 3, kg = e >> 7;
            const float* s0 = ys + (kg * 8 + hf * 4) * YS + py * 16 + px;
            float4 v4 = make_float4(s0[0], s0[YS], s0[2 * YS], s0[3 * YS]);
            size_t o;
            if (Q8) {
                const int vy = wy0 + py, img = 4 * blk + 2 * (vy >> 3) + (px >> 3);
                if (img >= nimg) continue;
                o = ((((size_t)img * KG + kb * 4 + kg) * 8 + (vy & 7)) * 8 + (px & 7)) * 8 + hf * 4;
            } else
                o = (((size_t)bimg * KG + kb * 4 + kg) * H + wy0 + py) * W * 8 + (size_t)(ox0 + px) * 8 + hf * 4;
            if (!FWD) {
                if (out_addend) {
                    const float4 ad = *reinterpret_cast<const float4*>(out_addend + o);
                    v4.x += ad.x; v4.y += ad.y; v4.z += ad.z; v4.w += ad.w;
                }
                if (out_mask) {
                    const float4 m = *reinterpret_cast<const float4*>(out_mask + o);
                    v4.x = m.x > 0.f ? v4.x : 0.f; v4.y = m.y > 0.f ? v4.y : 0.f;
                    v4.z = m.z > 0.f ? v4.z : 0.f; v4.w = m.w > 0.f ? v4.w : 0.f;
                }
            }
            *reinterpret_cast<float4*>(y + o) = v4;
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

// NCHW <-> NCHW8c (blocked) layout conversion; towards the blocked layout optionally y = max(x + bias[c], 0) (the bias + ReLU
// behind the library convolution that feeds the stack: two elementwise passes over the largest map saved)
__global__ void nchw_to_blocked_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int C, int HW,
                                       int to_blocked, const float* __restrict__ bias, int relu) {
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
    if (to_blocked) {
        float v = src[plain];
        if (bias) v += bias[cg * 8 + c8];
        if (relu) v = fmaxf(v, 0.f);
        dst[e] = v;
    } else {
        dst[plain] = src[e];
    }
}

// the same conversion for HW % 4 == 0: one thread = 4 pixels x 8 channels, so that both sides move 16-byte vectors and a
// wave touches 1 KB runs of every plane (the scalar kernel above reads 32-byte runs: 2.4 TB/s on the 268 MB first VGG map)
__global__ __launch_bounds__(256) void nchw_to_blocked_vec_kernel(const float* __restrict__ src, float* __restrict__ dst, int C,
                                                                  int HW, size_t total, int to_blocked,
                                                                  const float* __restrict__ bias, int relu) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;          // over [b][c/8][hw/4]
    if (e >= total) return;
    const int HW4 = HW / 4;
    const int q = e % HW4;
    const size_t r2 = e / HW4;
    const int cg = r2 % (C / 8);
    const size_t b = r2 / (C / 8);
    const size_t pbase = (b * C + cg * 8) * HW + 4 * q;                      // plain: channel c at pbase + c * HW
    const size_t bbase = ((b * (C / 8) + cg) * HW + 4 * q) * 8;              // blocked: pixel p at bbase + 8 p
    f32x4 v[8];
    if (to_blocked) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            v[c] = *reinterpret_cast<const f32x4*>(src + pbase + (size_t)c * HW);
            if (bias) { const float bc = bias[cg * 8 + c]; v[c][0] += bc; v[c][1] += bc; v[c][2] += bc; v[c][3] += bc; }
            if (relu) { v[c][0] = fmaxf(v[c][0], 0.f); v[c][1] = fmaxf(v[c][1], 0.f); v[c][2] = fmaxf(v[c][2], 0.f); v[c][3] = fmaxf(v[c][3], 0.f); }
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            *reinterpret_cast<f32x4*>(dst + bbase + 8 * p) = f32x4{v[0][p], v[1][p], v[2][p], v[3][p]};
            *reinterpret_cast<f32x4*>(dst + bbase + 8 * p + 4) = f32x4{v[4][p], v[5][p], v[6][p], v[7][p]};
        }
    } else {
        f32x4 u[8];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            u[2 * p] = *reinterpret_cast<const f32x4*>(src + bbase + 8 * p);
            u[2 * p + 1] = *reinterpret_cast<const f32x4*>(src + bbase + 8 * p + 4);
        }
#pragma unroll
        for (int c = 0; c < 8; ++c)
            *reinterpret_cast<f32x4*>(dst + pbase + (size_t)c * HW) =
                f32x4{u[(c >> 2)][c & 3], u[2 + (c >> 2)][c & 3], u[4 + (c >> 2)][c & 3], u[6 + (c >> 2)][c & 3]};
    }
}

// 2x2 / stride 2 max pooling in the blocked layout ([n = b * C/8][h][w][8]); one thread = 4 channels of one pooled pixel.
__global__ void maxpool2x2_blocked_fwd_kernel(const float4* __restrict__ x, float4* __restrict__ y, size_t total, int Ho,
                                              int Wo) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;     // over [n][ho][wo][2 halves]
    if (e >= total) return;
    const int hf = e & 1;
    const size_t p = e >> 1;
    const int wo = p % Wo;
    const size_t r = p / Wo;
    const int ho = r % Ho;
    const size_t n = r / Ho;
    const int W = 2 * Wo;
    const float4* src = x + ((n * (2 * Ho) + 2 * ho) * W + 2 * wo) * 2 + hf;
    const float4 a = src[0], b = src[2], c = src[2 * W], d = src[2 * W + 2];
    float4 m;
    m.x = fmaxf(fmaxf(a.x, b.x), fmaxf(c.x, d.x)); m.y = fmaxf(fmaxf(a.y, b.y), fmaxf(c.y, d.y));
    m.z = fmaxf(fmaxf(a.z, b.z), fmaxf(c.z, d.z)); m.w = fmaxf(fmaxf(a.w, b.w), fmaxf(c.w, d.w));
    y[e] = m;
}

// Backward of the pooling of a post-ReLU map `act`: the gradient goes to the first maximum of each window (the
// library's tie rule) and, fused, through the ReLU below it (act > 0) - a window of zeros passes nothing.
__global__ void maxpool2x2_blocked_bwd_kernel(const float4* __restrict__ gy, const float4* __restrict__ act,
                                              float4* __restrict__ gx, size_t total, int Ho, int Wo) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int hf = e & 1;
    const size_t p = e >> 1;
    const int wo = p % Wo;
    const size_t r = p / Wo;
    const int ho = r % Ho;
    const size_t n = r / Ho;
    const int W = 2 * Wo;
    const size_t o = ((n * (2 * Ho) + 2 * ho) * W + 2 * wo) * 2 + hf;
    const float4 a = act[o], b = act[o + 2], c = act[o + 2 * W], d = act[o + 2 * W + 2];
    const float4 g = gy[e];
    float4 ra, rb, rc, rd;
#define ROUTE(f)                                                                              \
    {                                                                                         \
        const float m = fmaxf(fmaxf(a.f, b.f), fmaxf(c.f, d.f));                              \
        const float gv = m > 0.f ? g.f : 0.f;                                                 \
        const bool ia = a.f == m, ib = !ia && b.f == m, ic = !ia && !ib && c.f == m;          \
        ra.f = ia ? gv : 0.f; rb.f = ib ? gv : 0.f; rc.f = ic ? gv : 0.f;                     \
        rd.f = (!ia && !ib && !ic) ? gv : 0.f;                                                \
    }
    ROUTE(x) ROUTE(y) ROUTE(z) ROUTE(w)
#undef ROUTE
    gx[o] = ra; gx[o + 2] = rb; gx[o + 2 * W] = rc; gx[o + 2 * W + 2] = rd;
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

extern "C" int dhz_winograd_conv3x3(const float* x, const float* upack, const float* bias, int relu, const float* out_mask,
                                    const float* out_addend, float* y, int B, int H, int W, int C, int K, void* stream) {
    DHZ_REQUIRE(x && upack && y, "dhz_winograd_conv3x3: null pointer");
    const bool q8 = H == 8 && W == 8;                             // four 8 x 8 images per 16 x 16 block
    DHZ_REQUIRE(B > 0 && ((H % 16 == 0 && W % 16 == 0) || q8) && C % CC == 0 && K % KB == 0,
                "dhz_winograd_conv3x3: unsupported shape B=%d H=%d W=%d C=%d K=%d", B, H, W, C, K);
    const bool fwd = !(out_mask || out_addend);
    DHZ_REQUIRE(fwd || !(bias || relu), "dhz_winograd_conv3x3: bias/relu and out_mask/out_addend are exclusive");
    const int nblk = q8 ? (B + 3) / 4 : B * (H / 16) * (W / 16);  // 16x16-pixel output blocks, two per workgroup
    const int grid = ((nblk + 1) / 2) * (K / KB);
    hipStream_t s = (hipStream_t)stream;
#ifdef DHZ_DIAG
    static const int xcd_env = getenv("DHZ_WINO_XCD") ? atoi(getenv("DHZ_WINO_XCD")) : -1;
#else
    constexpr int xcd_env = -1;
#endif
           // tuning aid: 0 / 1 forces
    // measured (FETCH_SIZE, batch 64): 2-4x fewer HBM reads on the K <= 256 layers (reads ~= the input once), no change
    // at K = 512, never slower
    const int xcd_group = (grid % 8 == 0) && (xcd_env >= 0 ? xcd_env : 1);
#define GO(F, Q)                                                                                                   \
    do {                                                                                                           \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&winograd_conv3x3_kernel<F, Q>),                   \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)WINO_SMEM);                     \
        hipLaunchKernelGGL((winograd_conv3x3_kernel<F, Q>), dim3(grid), dim3(512), WINO_SMEM, s, x, upack, bias, relu, \
                           out_mask, out_addend, y, H, W, C, K, nblk, xcd_group, B);                               \
    } while (0)
    if (q8) { if (fwd) GO(true, true); else GO(false, true); }
    else { if (fwd) GO(true, false); else GO(false, false); }
#undef GO
    DHZ_CHECK_LAUNCH("dhz_winograd_conv3x3");
    return DHZ_OK;
}

extern "C" int dhz_layout_blocked8(const float* src, float* dst, int B, int C, int HW, int to_blocked, const float* bias,
                                   int relu, void* stream) {
    DHZ_REQUIRE(src && dst && C % 8 == 0 && B > 0 && HW > 0, "dhz_layout_blocked8: bad arguments");
    DHZ_REQUIRE(to_blocked || !(bias || relu), "dhz_layout_blocked8: bias / relu only towards the blocked layout");
    const size_t n = (size_t)B * C * HW;
    if (HW % 4 == 0) {
        const size_t total = n / 32;
        hipLaunchKernelGGL(nchw_to_blocked_vec_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                           src, dst, C, HW, total, to_blocked, bias, relu);
    } else {
        hipLaunchKernelGGL(nchw_to_blocked_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src,
                           dst, B, C, HW, to_blocked, bias, relu);
    }
    DHZ_CHECK_LAUNCH("dhz_layout_blocked8");
    return DHZ_OK;
}

extern "C" int dhz_maxpool2x2_blocked_fwd(const float* x, float* y, int N, int H, int W, void* stream) {
    DHZ_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "dhz_maxpool2x2_blocked_fwd: bad arguments");
    const size_t total = (size_t)N * (H / 2) * (W / 2) * 2;
    hipLaunchKernelGGL(maxpool2x2_blocked_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4*>(x), reinterpret_cast<float4*>(y), total, H / 2, W / 2);
    DHZ_CHECK_LAUNCH("dhz_maxpool2x2_blocked_fwd");
    return DHZ_OK;
}

extern "C" int dhz_maxpool2x2_blocked_bwd(const float* gy, const float* act, float* gx, int N, int H, int W, void* stream) {
    DHZ_REQUIRE(gy && act && gx && N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "dhz_maxpool2x2_blocked_bwd: bad arguments");
    const size_t total = (size_t)N * (H / 2) * (W / 2) * 2;
    hipLaunchKernelGGL(maxpool2x2_blocked_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4*>(gy), reinterpret_cast<const float4*>(act), reinterpret_cast<float4*>(gx),
                       total, H / 2, W / 2);
    DHZ_CHECK_LAUNCH("dhz_maxpool2x2_blocked_bwd");
    return DHZ_OK;
}
