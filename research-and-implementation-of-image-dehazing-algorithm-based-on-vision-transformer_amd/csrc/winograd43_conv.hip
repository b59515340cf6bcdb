// K11b - the 3x3 / stride 1 / pad 1 convolutions of the VGG19 feature extractor (My_CR.py:56-86) on maps of 16 x 16 and more as
// Winograd F(4x4, 3x3): 36 transform-domain products per 16 outputs instead of F(2x2, 3x3)'s 16 per 4 (csrc/winograd_conv.hip) -
// 1.78 x fewer v_mfma_f32_16x16x4_f32 on the same fp32 matrix pipe.
//
// Accuracy (profiles/r05_winograd_f43_error_table.txt).  F(4x4) in fp32 with the textbook points (0, +-1, +-2) is 10 x less accurate than
// F(2x2) and fails the kernel-level tolerance of tests/test_gpu_winograd.py; two measures bring it to ~3 x F(2x2)'s error:
//   * interpolation points (0, +-3/4, +-3/2, inf): every constant of B^T and A^T is a dyadic rational, the transforms' amplification
//     is the smallest of the symmetric sets searched (symmetric: B^T d shares the even / odd parts between +-p: 14 packed
//     operations per 6-point transform);
//   * (the remaining error is dominated by the fp32 ACCUMULATION over input channels of transform-domain values that are larger
//     than the outputs; a flush of the accumulators through A^T . A every 128 channels halves it - measured on the first form of this
//     kernel - but does not fit the registers of this one);
//   * the frozen filters are transformed once (G g G^T in double, rounded once).
//
// Structure.  36 positions x 16 tiles x 32 output channels = 288 accumulator registers: 256 threads, ONE wave per SIMD with the whole
// register file (256 accumulators in the AGPR half, the rest with everything else in the VGPR half).  A wave owns a 16 x 16-pixel
// output block (16 tiles of 4 x 4) x 32 output channels for all 36 positions.  Lane (tile i16, channel pair g) transforms ITS OWN tile
// for ITS OWN two channels, and the MFMA contraction index of lane group g at step s is channel 2 g + s - so the 36 x 2 transformed
// values a lane computes ARE its B operands: the transformed input never leaves the registers.  Per group of 8 input channels:
//   (1) transform phase: 36 ds_read_b64 of the lane's 6 x 6 window, B^T d B as 12 six-point transforms of 14 packed operations;
//   (2) matrix phase: 144 MFMAs (A fragments: one ds_read_b128 per position pair and output-channel half); between them the LDS-DMA
//       requests of the NEXT group, one per MFMA quad (issued back to back they queue in the address unit: 100 cycles each).
// The phases are sequential on purpose: vector and fp32 matrix instructions of one wave do not overlap on this part (the first
// form of this kernel - 16 output channels per wave, the transform of group cb + 1 interleaved with the MFMAs of cb through two
// register sets - measured its transform and its DMA issue as ADDED to the matrix time and lost to F(2x2); what pays is the ratio:
// 32 output channels per wave halve the vector work and the operand bytes per MFMA).
// The four waves of a workgroup take four blocks and share the filter slice (32 k x 8 c x 36 positions = 36 KiB per group) through a
// two-slot ring (one barrier per group, between the phases); a wave's 18 x 18-pixel halo patch has ONE slot (it is rewritten by DMA during
// the matrix phase, after the transform has read it); out-of-image pixels are slots zeroed once that no DMA lane ever writes.
// Rounding: no mid-stream flush fits the register file at this tile (an output-domain accumulator is 128 more registers; a flush
// through the output tile inside the channel loop made hipcc spill 100 - 470 registers and cost 30 %): with the points above the
// error is 1.1e-6 rms at 64 input channels, 2.1e-6 at 256, 2.9e-6 at 512 on O(1) outputs (F(2x2): 3.7e-7 .. 9.4e-7) - 0.26 - 0.7 of the
// kernel tolerance up to 256 channels, 0.9 - 1.0 at 512.  More than 256 channels therefore run as chains of 256 in separate launches
// that hand their partial sums over through the output tensor (see the C entry point); dehaze_hip/vgg.py therefore keeps the DIFFERENTIATED forward pass on F(2x2) (its
// roundings decide the ReLU masks of the backward pass) and uses this kernel for the no-gradient passes and the backward-data products.
#include <stdlib.h>
#include <type_traits>
#include "common.h"

#ifndef W43_FORM
#define W43_FORM 2      // 2: one wave per SIMD, 32 output channels per wave;  3: two waves per SIMD in opposite phases (the paired form)
#endif
#ifndef W43_ABL
#define W43_ABL 0       // timing diagnostics: 1 no input transform, 2 no DMA, 4 no MFMAs
#endif

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

constexpr int KB = 32;              // output channels per workgroup (= per wave)
constexpr int CC = 8;               // input channels per group (the layout's channel block)
constexpr int NXI = 36;             // transform positions
constexpr int UF = NXI * CC * KB;   // floats of one (kb, cb) filter slice: 9216 = 36 KiB = 36 DMA runs of 1 KiB
constexpr int NUBUF = 2;
constexpr int URUNS = UF / 256;     // 36: nine per wave
static_assert(URUNS / 4 == 9, "the counted wait at the end of a group (vmcnt(9)) leaves exactly a wave's filter requests in flight");
// halo patch of a wave: 18 x 18 pixels x 8 channels as 16-byte chunks (4 channels of one pixel), plane hf = channels 4 hf .. 4 hf + 3:
//   chunk(hf, row, px) = hf * PLANE + row * 18 + px + (row >> 2)
// the one-chunk skew per four rows makes the ds_read_b64 of the 64 lanes (tile rows 4 ty + a, tile columns 4 tx + p, channel pair g)
// conflict-free: bank starts 16 tx + {0, 36, 8, 44}[ty] + 2 (g & 1) are the 32 even numbers
constexpr int PLANE = 332;
constexpr int PCHUNKS = 2 * PLANE;                  // 664 chunks = 10.4 KiB
constexpr int PRUNS = (PCHUNKS + 63) / 64;          // 11 DMA runs
constexpr int PFLOATS = PRUNS * 256;                // the patch slot (floats), runs are whole KiB
#ifndef W43_CHAIN
#define W43_CHAIN 32
#endif
constexpr int CHAIN_GROUPS = W43_CHAIN;             // channel groups (of 8) per accumulation chain: 256 channels
constexpr size_t W43_SMEM = (size_t)(NUBUF * UF + 4 * PFLOATS + KB) * sizeof(float);      // 72 + 44 KiB

// interpolation points 0, +-PA, +-PB, inf
constexpr float PA = 0.75f, PB = 1.5f;
constexpr float A2 = PA * PA, B2 = PB * PB, A2B2 = A2 * B2, SAB = A2 + B2, AB2 = PA * B2, A2B = A2 * PB;
constexpr float A3 = A2 * PA, B3 = B2 * PB;

__device__ __forceinline__ void dma16(const float* g, float* l) {
    __builtin_amdgcn_global_load_lds((glb_void*)g, (lds_void*)l, 16, 0, 0);
}
// the same request with its address as (64-bit base in SGPRs) + (unsigned 32-bit byte offset per lane): hipcc selects the all-VGPR address
// form for the builtin (a 64-bit vector add per request); written out, the request needs no vector instruction at all
__device__ __forceinline__ void dma16s(const char* sbase, unsigned voff, float* l) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :: "v"(voff), "s"(sbase), "s"((unsigned)(size_t)(lds_void*)l) : "memory", "m0");
}

// 288 accumulator registers against 256 AGPRs: left to itself hipcc keeps 240 in AGPRs, the rest in VGPRs, and SHUTTLES the latter through
// a[0:15] around every MFMA that touches them (4 v_accvgpr_write, the MFMA, s_nop 9, 4 v_accvgpr_read: 24 of the 144 MFMAs of a group cost
// twice their time and cannot overlap their neighbours).  The matrix instruction takes its C / D operand from either half of the register
// file, so the class is pinned per accumulator through the asm constraint: positions 0 .. 31 in AGPRs ("a"), positions 32 .. 35 in VGPRs
// ("v").  hipcc cannot see that these are matrix instructions: the hazards it would cover are covered by construction - an accumulator
// is reused four MFMAs (128 pipe cycles) later, and the epilogue waits 2 x s_nop 15 before it reads them.
__device__ __forceinline__ void mfma_acc_a(f32x4& acc, float a, float b) {
    asm("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_acc_v(f32x4& acc, float a, float b) {
    asm("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
// B^T d for one 6-vector (packed over the lane's two channels): rows for the points 0, +a, -a, +b, -b, inf.  Twelve fused operations: the odd
// parts PA d3 - PA B2 d1 = PA (d3 - B2 d1) and PB (d3 - A2 d1) take their factor in the FMA that forms the +- rows (written as sums of
// products hipcc emits 16 - the two extra multiplications of each of the twelve transforms were 12 % of the vector work of a group, and
// vector work ADDS to the matrix time on this part: tools/ubench/overlap.hip)
__device__ __forceinline__ f32x2 fma2(float a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(f32x2{a, a}, b, c); }
__device__ __forceinline__ void bt6(f32x2& d0, f32x2& d1, f32x2& d2, f32x2& d3, f32x2& d4, f32x2& d5) {
    const f32x2 ea = fma2(-B2, d2, d4), eb = fma2(-A2, d2, d4);
    const f32x2 ua = fma2(-B2, d1, d3), ub = fma2(-A2, d1, d3);
    const f32x2 t0 = fma2(A2B2, d0, fma2(-SAB, d2, d4));
    const f32x2 t5 = fma2(A2B2, d1, fma2(-SAB, d3, d5));
    d0 = t0; d1 = fma2(PA, ua, ea); d2 = fma2(-PA, ua, ea); d3 = fma2(PB, ub, eb); d4 = fma2(-PB, ub, eb); d5 = t5;
}
// A^T m for one 6-vector -> 4 outputs
__device__ __forceinline__ void at6(float m0, float m1, float m2, float m3, float m4, float m5, float& y0, float& y1, float& y2,
                                    float& y3) {
    const float s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
    y0 = m0 + s1 + s2;
    y1 = PA * d1 + PB * d2;
    y2 = A2 * s1 + B2 * s2;
    y3 = A3 * d1 + B3 * d2 + m5;
}

template <bool FWD>
__global__ __launch_bounds__(256, 1) void winograd43_conv3x3_kernel(const float* __restrict__ x, const float* __restrict__ upack,
                                                                    const float* __restrict__ bias, int relu,
                                                                    const float* __restrict__ out_mask,
                                                                    const float* __restrict__ out_addend, float* __restrict__ y, int H,
                                                                    int W, int C, int K, int nblk, int xcd_group, int cb0, int ncb,
                                                                    int chain) {
    // cb0, ncb: the channel groups [cb0, cb0 + ncb) this launch accumulates; chain bit 0: add the partial sums an earlier launch left
    // in y, bit 1: leave raw partial sums in y (no bias / ReLU / mask / addend) for a later launch - see dhz_winograd43_conv3x3
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int i16 = lane & 15, g = lane >> 4;
    const int ty = i16 >> 2, tx = i16 & 3, hf = g >> 1, sub = g & 1;
    float* const us = smem;                                        // filter ring [NUBUF][UF]
    float* const pw = smem + NUBUF * UF + w * PFLOATS;             // this wave's patch slot
    float* const bias_s = smem + NUBUF * UF + 4 * PFLOATS;
    const int KBn = K / KB, CBn = C / CC;
    int lid = blockIdx.x;
    if (xcd_group) lid = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int kb = lid % KBn;
    int blk = (lid / KBn) * 4 + w;
    const bool live = blk < nblk;                                  // ragged last workgroup: the extra waves recompute, store nothing
    if (!live) blk = nblk - 1;
    const int bx_n = W / 16, by_n = H / 16;
    const int bimg = blk / (bx_n * by_n);
    const int by = (blk / bx_n) % by_n, bx = blk % bx_n;
    const int oy0 = by * 16, ox0 = bx * 16;                        // the wave's output block; patch origin (oy0 - 1, ox0 - 1)
    const size_t plane = (size_t)H * W * 8;                        // floats per (image, channel-group) plane

    // ---- zero the patch slot (padding pixels and layout gaps are never written again) and stage the bias
    {
        f32x4* z = reinterpret_cast<f32x4*>(pw);
#pragma unroll
        for (int i = 0; i < PFLOATS / 4 / 64; ++i) z[lane + 64 * i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (FWD && t < KB) bias_s[t] = bias ? bias[kb * KB + t] : 0.f;

    // ---- per-lane DMA sources of the patch: run r moves LDS chunks 64 r .. 64 r + 63, lane -> chunk 64 r + lane
    int poff[PRUNS];                    // float offset inside a plane, or -1: no pixel (gap of the layout / outside the image)
#pragma unroll
    for (int r = 0; r < PRUNS; ++r) {
        const int c = 64 * r + lane;
        const int ph = c / PLANE, rr = c - ph * PLANE;
        const int k4 = rr / 73;                                    // row group (4 rows + one skew chunk = 73 chunks)
        const int q = rr - 73 * k4;
        const int row = 4 * k4 + q / 18, px = q % 18;
        const int iy = oy0 - 1 + row, ix = ox0 - 1 + px;
        const bool ok = ph < 2 && q < 72 && row < 18 && iy >= 0 && iy < H && ix >= 0 && ix < W;
        poff[r] = ok ? (iy * W + ix) * 8 + ph * 4 : -1;
    }
    // request addresses as (wave-uniform base) + (unsigned 32-bit lane offset): the form global_load_lds takes with its base in SGPRs - no
    // vector instruction per request (a 64-bit vector add between two MFMAs costs its whole latency of matrix-pipe time: tools/ubench/overlap.hip)
    const char* const xbase = reinterpret_cast<const char*>(x + ((size_t)bimg * CBn + cb0) * plane);
    const char* const ubase = reinterpret_cast<const char*>(upack + ((size_t)kb * CBn + cb0) * UF);
    const unsigned lane16 = lane * 16;
    auto dma_patch_run = [&](int cb, int r) {
        if (W43_ABL & 2) return;
        if (poff[r] >= 0) dma16s(xbase + (size_t)cb * plane * 4, (unsigned)(poff[r] * 4), pw + 256 * r);
    };
    auto dma_u_run = [&](int cb, int r) {                          // run w + 4 r of slice cb -> ring slot cb & 1
        if (W43_ABL & 2) return;
        const int run = w + 4 * r;
        dma16s(ubase + ((size_t)cb * UF + 256 * run) * 4, lane16, us + (cb & 1) * UF + 256 * run);
    };

    const float* const pread0 = pw + (hf * PLANE + 73 * ty + 4 * tx) * 4 + 2 * sub;
    constexpr int NXA = 32;             // positions whose accumulators live in AGPRs (32 x 2 x 4 = 256 registers); the other four in VGPRs
    f32x4 acc[NXA][2], accv[NXI - NXA][2];
#pragma unroll
    for (int xi = 0; xi < NXA; ++xi) { acc[xi][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[xi][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int xi = 0; xi < NXI - NXA; ++xi) { accv[xi][0] = f32x4{0.f, 0.f, 0.f, 0.f}; accv[xi][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    auto A = [&](int xi, int kh) -> f32x4& { return xi < NXA ? acc[xi][kh] : accv[xi - NXA][kh]; };

    // ---- prologue: patch 0 and filter slice 0
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // the zero fill is in LDS before any DMA lane can land on top of it
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < PRUNS; ++r) dma_patch_run(0, r);
#pragma unroll
    for (int r = 0; r < URUNS / 4; ++r) dma_u_run(0, r);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    const float* const ufrag = us + (g * 16 + i16) * 4;            // [xi pair][k half][g][k][xi & 1][c & 1]: one ds_read_b128 per two positions
    // One channel group, two phases (vector and fp32 matrix instructions of a wave do not overlap on this part - csrc/leff_fused.hip,
    // DESIGN 4a - so there is nothing to gain from interleaving them, and one set of transformed values is enough):
    //   (1) the lane's 6 x 6 x 2-channel patch window from LDS, B^T d B in registers;
    //   (2) barrier (filter slice cb complete for every wave, every wave done with slice cb - 1), 144 MFMAs; in their first half the
    //       DMA of the next group's operands, one request per MFMA pair so that the address unit never queues
    // the lane's 6 x 6 x 2-channel window: read from the patch slot into v, transformed IN PLACE (v[a][b] then holds position 6 a + b), consumed
    // by the matrix phase in position order - and REFILLED with the next group's window as soon as its positions are dead (rows 0 .. 4 after
    // pair-step 14, row 5 after the last one): the 36 LDS reads of a group (18 KiB per wave, 580 cycles of the CU's LDS bandwidth with nothing
    // else to run beside them) now overlap the last matrix instructions of the group before
    f32x2 v[6][6];
    auto read_rows = [&](int a0, int a1) {
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            if (a < a0 || a >= a1) continue;
            const float* p = pread0 + (a * 18 + (a >> 2)) * 4;
#pragma unroll
            for (int b = 0; b < 6; ++b) v[a][b] = *reinterpret_cast<const f32x2*>(p + 4 * b);
        }
    };
    if (!(W43_ABL & 1)) read_rows(0, 6);
#pragma unroll 1
    for (int cb = 0; cb < ncb; ++cb) {
        if (!(W43_ABL & 1)) {
            if (!(W43_ABL & 16)) {
#pragma unroll
            for (int b = 0; b < 6; ++b) bt6(v[0][b], v[1][b], v[2][b], v[3][b], v[4][b], v[5][b]);
#pragma unroll
            for (int a = 0; a < 6; ++a) bt6(v[a][0], v[a][1], v[a][2], v[a][3], v[a][4], v[a][5]);
            }
            // pin the transform HERE, as one dense block with six independent chains to interleave: left alone LLVM sinks each row to its
            // first use inside the matrix phase, where every vector instruction delays the next MFMA (2200 cycles per group measured)
#pragma unroll
            for (int a = 0; a < 6; ++a)
                asm volatile("" : "+v"(v[a][0]), "+v"(v[a][1]), "+v"(v[a][2]), "+v"(v[a][3]), "+v"(v[a][4]), "+v"(v[a][5]));
        } else {
#pragma unroll
            for (int a = 0; a < 6; ++a)
#pragma unroll
                for (int b = 0; b < 6; ++b) v[a][b] = f32x2{1.f, 1.f};
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // my runs of filter slice cb have landed
        __syncthreads();
        const bool next = cb + 1 < ncb;
        const float* up = ufrag + (cb & 1) * UF;
        f32x4 alo = *reinterpret_cast<const f32x4*>(up), ahi = *reinterpret_cast<const f32x4*>(up + 256);
#pragma unroll
        for (int j = 0; j < 18; ++j) {
            f32x4 nlo = alo, nhi = ahi;
            if (j + 1 < 18) {
                nlo = *reinterpret_cast<const f32x4*>(up + (j + 1) * 512);
                nhi = *reinterpret_cast<const f32x4*>(up + (j + 1) * 512 + 256);
            }
            const int xi = 2 * j;
            const f32x2 v0 = v[xi / 6][xi % 6], v1 = v[(xi + 1) / 6][(xi + 1) % 6];
            __builtin_amdgcn_sched_barrier(0);
            if (!(W43_ABL & 4)) {
                if (xi < NXA) {                                   // (xi is a constant after unrolling; xi and xi + 1 are on the same side)
                    mfma_acc_a(acc[xi][0], alo[0], v0[0]); mfma_acc_a(acc[xi][1], ahi[0], v0[0]);
                    mfma_acc_a(acc[xi + 1][0], alo[2], v1[0]); mfma_acc_a(acc[xi + 1][1], ahi[2], v1[0]);
                } else {
                    mfma_acc_v(accv[xi - NXA][0], alo[0], v0[0]); mfma_acc_v(accv[xi - NXA][1], ahi[0], v0[0]);
                    mfma_acc_v(accv[xi + 1 - NXA][0], alo[2], v1[0]); mfma_acc_v(accv[xi + 1 - NXA][1], ahi[2], v1[0]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // the next patch first (two requests per pair-step: it is needed as soon as this group ends), the next filter slice behind it
            // (needed one transform phase later: the wait at the end of the group leaves those nine requests in flight)
            if (next && 2 * j < PRUNS) dma_patch_run(cb + 1, 2 * j);
            if (next && j >= 6 && j - 6 < URUNS / 4) dma_u_run(cb + 1, j - 6);
            __builtin_amdgcn_sched_barrier(0);
            if (!(W43_ABL & 4)) {
                if (xi < NXA) {
                    mfma_acc_a(acc[xi][0], alo[1], v0[1]); mfma_acc_a(acc[xi][1], ahi[1], v0[1]);
                    mfma_acc_a(acc[xi + 1][0], alo[3], v1[1]); mfma_acc_a(acc[xi + 1][1], ahi[3], v1[1]);
                } else {
                    mfma_acc_v(accv[xi - NXA][0], alo[1], v0[1]); mfma_acc_v(accv[xi - NXA][1], ahi[1], v0[1]);
                    mfma_acc_v(accv[xi + 1 - NXA][0], alo[3], v1[1]); mfma_acc_v(accv[xi + 1 - NXA][1], ahi[3], v1[1]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (next && 2 * j + 1 < PRUNS) dma_patch_run(cb + 1, 2 * j + 1);
            __builtin_amdgcn_sched_barrier(0);
            alo = nlo; ahi = nhi;
            if (next && !(W43_ABL & 1)) {
                if (j == 14) {
                    // the next patch has landed (requested in pair-steps 0 .. 5; my nine filter requests, the last one issued in this pair-step,
                    // may still be in flight); positions 0 .. 29 are consumed: their registers take rows 0 .. 4 of the next window
                    asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
                    read_rows(0, 5);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (j == 17) read_rows(5, 6);
            }
        }
    }

    // the last matrix instructions (inline asm: hipcc does not know what they are) have written their accumulators before these are read
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    // ---- epilogue: Y = A^T M A per (k half, accumulator row); lane (tile i16, g) holds output channels 16 kh + 4 g .. + 3 of its 4 x 4 pixels
    const int KG = K / 8;
    const float lo = relu ? 0.f : -__builtin_inff();
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
        float yv[4][16];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float tm[4][6];
#pragma unroll
            for (int c = 0; c < 6; ++c)
                at6(A(c, kh)[j], A(6 + c, kh)[j], A(12 + c, kh)[j], A(18 + c, kh)[j], A(24 + c, kh)[j], A(30 + c, kh)[j],
                    tm[0][c], tm[1][c], tm[2][c], tm[3][c]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                at6(tm[i][0], tm[i][1], tm[i][2], tm[i][3], tm[i][4], tm[i][5], yv[j][4 * i], yv[j][4 * i + 1], yv[j][4 * i + 2],
                    yv[j][4 * i + 3]);
        }
        if (live) {
            const float4 bv = (FWD && !(chain & 2)) ? *reinterpret_cast<const float4*>(bias_s + 16 * kh + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
            const size_t obase = (((size_t)bimg * KG + kb * 4 + kh * 2 + hf) * H + oy0 + 4 * ty) * W * 8 + (size_t)(ox0 + 4 * tx) * 8 + sub * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int jx = 0; jx < 4; ++jx) {
                    const int p = 4 * i + jx;
                    float4 v4 = make_float4(yv[0][p] + bv.x, yv[1][p] + bv.y, yv[2][p] + bv.z, yv[3][p] + bv.w);
                    const size_t o = obase + ((size_t)i * W + jx) * 8;
                    if (chain & 1) {
                        const float4 part = *reinterpret_cast<const float4*>(y + o);
                        v4.x += part.x; v4.y += part.y; v4.z += part.z; v4.w += part.w;
                    }
                    if (chain & 2) {
                    } else if (FWD) {
                        v4.x = fmaxf(v4.x, lo); v4.y = fmaxf(v4.y, lo); v4.z = fmaxf(v4.z, lo); v4.w = fmaxf(v4.w, lo);
                    } else {
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
}


// ---- the paired form: TWO waves per SIMD that split the 36 positions, in opposite phases ----------------------------------------------
// The one-wave form above adds its transform (2,000 cycles per group) and its DMA issue to 4,608 cycles of matrix instructions: nothing
// else is resident on the SIMD to feed the matrix pipe while the wave does vector work.  Here a 512-thread workgroup puts waves w and
// w + 4 on one SIMD; both own the SAME 16 x 16-pixel block and all 32 output channels, wave w ("role 0") the transform-domain rows
// 0 .. 2 (positions 0 .. 17) and wave w + 4 ("role 1") the rows 3 .. 5: 18 positions x 2 channel halves x 4 = 144 accumulators each, 36
// transformed values per lane instead of 72, and HALF a transform each (B^T d for three of the six rows: 8 packed operations per column
// instead of 14, then three full row transforms: 90 packed operations against 168) - no vector work is duplicated.  Time runs in
// half-intervals separated by workgroup barriers: role 0 transforms group g in half 2 g and multiplies in half 2 g + 1, role 1 transforms
// in 2 g + 1 and multiplies in 2 g + 2, so on every SIMD one wave feeds the matrix pipe while the other one does its vector work.
// Buffers: patch g is read in halves 2 g and 2 g + 1 -> two slots per block; patch g + 1 is requested by the role-0 wave during its
// transform half 2 g (the vector half has the slack; the matrix halves carry no requests) and awaited at the end of its matrix half
// 2 g + 1.  Filter slice g is read in halves 2 g + 1 and 2 g + 2 -> two slots; slice g + 1 is requested by the role-1 waves during their
// transform half 2 g + 1 and awaited at the end of their matrix half 2 g + 2.  Every request has more than a whole half-interval (~1 us) to land.  The output transform Y = A^T M A is a sum over the rows of M: each
// wave transforms its three rows, the two partial 4 x 4 tiles meet through LDS (role r finishes channel half r).
constexpr int PSLOT = PCHUNKS * 4;                  // floats of one patch slot (664 chunks; the tail of DMA run 10 is masked)
constexpr size_t W43P_SMEM = (size_t)(NUBUF * UF + 8 * PSLOT + KB) * sizeof(float);      // 72 + 83 KiB
static_assert(W43P_SMEM <= 160 * 1024, "LDS of the paired form");
static_assert(8 * 64 * 64 * sizeof(float) <= (NUBUF * UF + 8 * PSLOT) * sizeof(float), "the partial-tile exchange reuses the operand buffers");
#ifndef W43P_SKEW
#define W43P_SKEW 0
#endif
#ifndef W43P_PAIR
#define W43P_PAIR 4      // partner of wave w on its SIMD: w + 4 (the waves of a workgroup go to the four SIMDs round-robin)
#endif

template <bool FWD>
__global__ __launch_bounds__(512, 1) void winograd43_pair_kernel(const float* __restrict__ x, const float* __restrict__ upack,
                                                                 const float* __restrict__ bias, int relu,
                                                                 const float* __restrict__ out_mask,
                                                                 const float* __restrict__ out_addend, float* __restrict__ y, int H,
                                                                 int W, int C, int K, int nblk, int xcd_group, int cb0, int ncb,
                                                                 int chain) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wb = W43P_PAIR == 4 ? (w & 3) : (w >> 1), role = W43P_PAIR == 4 ? (w >> 2) : (w & 1);
    const int i16 = lane & 15, g = lane >> 4;
    const int ty = i16 >> 2, tx = i16 & 3, hf = g >> 1, sub = g & 1;
    float* const us = smem;                                        // filter ring [2][UF]
    float* const pw = smem + NUBUF * UF + wb * 2 * PSLOT;          // this block's two patch slots
    float* const bias_s = smem + NUBUF * UF + 8 * PSLOT;
    const int KBn = K / KB, CBn = C / CC;
    int lid = blockIdx.x;
    if (xcd_group) lid = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int kb = lid % KBn;
    int blk = (lid / KBn) * 4 + wb;
    const bool live = blk < nblk;
    if (!live) blk = nblk - 1;
    const int bx_n = W / 16, by_n = H / 16;
    const int bimg = blk / (bx_n * by_n);
    const int by = (blk / bx_n) % by_n, bx = blk % bx_n;
    const int oy0 = by * 16, ox0 = bx * 16;
    const size_t plane = (size_t)H * W * 8;

    {   // zero both patch slots of the block (its two waves share the work)
        f32x4* z = reinterpret_cast<f32x4*>(pw);
        for (int i = role * 64 + lane; i < 2 * PSLOT / 4; i += 128) z[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (FWD && t < KB) bias_s[t] = bias ? bias[kb * KB + t] : 0.f;

    // patch DMA (role 0): run r moves LDS chunks 64 r .. 64 r + 63 of a slot
    int poff[PRUNS];
#pragma unroll
    for (int r = 0; r < PRUNS; ++r) {
        const int c = 64 * r + lane;
        const int ph = c / PLANE, rr = c - ph * PLANE;
        const int k4 = rr / 73;
        const int q = rr - 73 * k4;
        const int row = 4 * k4 + q / 18, px = q % 18;
        const int iy = oy0 - 1 + row, ix = ox0 - 1 + px;
        const bool ok = ph < 2 && q < 72 && row < 18 && iy >= 0 && iy < H && ix >= 0 && ix < W;
        poff[r] = ok ? (iy * W + ix) * 8 + ph * 4 : -1;
    }
    const float* const xbase = x + ((size_t)bimg * CBn + cb0) * plane;
    const float* const ubase = upack + ((size_t)kb * CBn + cb0) * UF + lane * 4;
    auto dma_patch_run = [&](int cb, int r) {
        if (W43_ABL & 2) return;
        if (poff[r] >= 0) dma16(xbase + (size_t)cb * plane + poff[r], pw + (cb & 1) * PSLOT + 256 * r);
    };
    auto dma_u_run = [&](int cb, int r) {                          // (role 1) run wb + 4 r of slice cb -> ring slot cb & 1
        if (W43_ABL & 2) return;
        const int run = wb + 4 * r;
        dma16(ubase + (size_t)cb * UF + 256 * run, us + (cb & 1) * UF + 256 * run);
    };

    const float* const pread0 = pw + (hf * PLANE + 73 * ty + 4 * tx) * 4 + 2 * sub;
    // 18 positions x 2 channel halves: hipcc splits the 256 registers of a wave 128 + 128, so local positions 0 .. 15 sit in the AGPR half,
    // 16 and 17 in VGPRs
    constexpr int NPL = 18, NXA = 16;
    f32x4 acc[NXA][2], accv[NPL - NXA][2];
#pragma unroll
    for (int p = 0; p < NXA; ++p) { acc[p][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[p][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int p = 0; p < NPL - NXA; ++p) { accv[p][0] = f32x4{0.f, 0.f, 0.f, 0.f}; accv[p][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    auto A = [&](int p, int kh) -> f32x4& { return p < NXA ? acc[p][kh] : accv[p - NXA][kh]; };

    // ---- prologue: patch 0 (role 0), filter slice 0 (role 1)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();                                               // the zero fill of both waves of the block is in LDS before a DMA lane lands
    if (!role) {
#pragma unroll
        for (int r = 0; r < PRUNS; ++r) dma_patch_run(0, r);
    } else {
#pragma unroll
        for (int r = 0; r < URUNS / 4; ++r) dma_u_run(0, r);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (W43P_SKEW && role) __syncthreads();                        // role 1 runs one half-interval behind (role 0 pays this barrier back after its loop)

    // Both roles run the SAME instruction stream; what differs is data (wave-uniform selects):
    //   B^T d, the role's three rows:  X = A2B2 x0 - SAB x2 + x4 over window rows (0, 2, 4) [role 0: point 0] or (1, 3, 5) [role 1: infinity];
    //   e = d4 - Ce d2, o = Co d3 - Cd d1 with (Ce, Co, Cd) = (B2, PA, AB2) [points +-a] or (A2, PB, A2B) [points +-b];  local rows (X, e + o, e - o)
    //   = transform rows (0, 1, 2) or (5, 3, 4) -> position pairs 3 a + b / 2 of the filter slice
    const float Ce = role ? A2 : B2, Co = role ? PB : PA, Cd = role ? A2B : AB2;
    constexpr int RO[6] = {0, 18 * 4, 36 * 4, 54 * 4, (72 + 1) * 4, (90 + 1) * 4};      // window row a -> float offset (a * 18 + (a >> 2)) * 4
    const float* const px0 = pread0 + (role ? RO[1] : RO[0]);
    const float* const px2 = pread0 + (role ? RO[3] : RO[2]);
    const float* const px4 = pread0 + (role ? RO[5] : RO[4]);
    const float* const ufrag = us + (g * 16 + i16) * 4;
    const float* const uf0 = ufrag + (role ? 15 : 0) * 512;       // filter fragments of local row 0, 1, 2
    const float* const uf1 = ufrag + (role ? 9 : 3) * 512;
    const float* const uf2 = ufrag + (role ? 12 : 6) * 512;
#pragma unroll 1
    for (int gi = 0; gi < ncb; ++gi) {
        // ---- transform half; between its column transforms the requests of the NEXT group's operands (the vector half has the slack, the
        //      matrix half stays free of them): role 0 the eleven runs of patch gi + 1 (its slot was last read in half 2 gi - 1), role 1 its
        //      nine runs of filter slice gi + 1 (its slot was last read in half 2 gi) - each has more than a whole half-interval to land
        f32x2 v[3][6];
        const bool nx = gi + 1 < ncb;
        if (!(W43_ABL & 1)) {
            const int so = (gi & 1) * PSLOT;
#pragma unroll
            for (int b = 0; b < 6; ++b) {
                if (nx) {
                    if (role == 0) { dma_patch_run(gi + 1, 2 * b); if (2 * b + 1 < PRUNS) dma_patch_run(gi + 1, 2 * b + 1); }
                    else { dma_u_run(gi + 1, 2 * b); if (2 * b + 1 < URUNS / 4) dma_u_run(gi + 1, 2 * b + 1); }
                }
                const f32x2 q1 = *reinterpret_cast<const f32x2*>(pread0 + so + RO[1] + 4 * b);
                const f32x2 q2 = *reinterpret_cast<const f32x2*>(pread0 + so + RO[2] + 4 * b);
                const f32x2 q3 = *reinterpret_cast<const f32x2*>(pread0 + so + RO[3] + 4 * b);
                const f32x2 q4 = *reinterpret_cast<const f32x2*>(pread0 + so + RO[4] + 4 * b);
                const f32x2 x0 = *reinterpret_cast<const f32x2*>(px0 + so + 4 * b);
                const f32x2 x2 = *reinterpret_cast<const f32x2*>(px2 + so + 4 * b);
                const f32x2 x4 = *reinterpret_cast<const f32x2*>(px4 + so + 4 * b);
                const f32x2 e = q4 - Ce * q2, o = Co * q3 - Cd * q1;
                v[0][b] = A2B2 * x0 - SAB * x2 + x4;
                v[1][b] = e + o;
                v[2][b] = e - o;
            }
#pragma unroll
            for (int a = 0; a < 3; ++a) bt6(v[a][0], v[a][1], v[a][2], v[a][3], v[a][4], v[a][5]);
#pragma unroll
            for (int a = 0; a < 3; ++a)
                asm volatile("" : "+v"(v[a][0]), "+v"(v[a][1]), "+v"(v[a][2]), "+v"(v[a][3]), "+v"(v[a][4]), "+v"(v[a][5]));
        } else {
            if (nx) {
                if (role == 0) {
#pragma unroll
                    for (int r = 0; r < PRUNS; ++r) dma_patch_run(gi + 1, r);
                } else {
#pragma unroll
                    for (int r = 0; r < URUNS / 4; ++r) dma_u_run(gi + 1, r);
                }
            }
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 6; ++b) v[a][b] = f32x2{1.f, 1.f};
        }
        __syncthreads();
        // ---- matrix half: 72 MFMAs on this role's half of filter slice gi (fragments one pair-step ahead)
        const int uo = (gi & 1) * UF;
        f32x4 alo = *reinterpret_cast<const f32x4*>(uf0 + uo), ahi = *reinterpret_cast<const f32x4*>(uf0 + uo + 256);
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            f32x4 nlo = alo, nhi = ahi;
            if (j + 1 < 9) {
                const float* up = ((j + 1) / 3 == 0 ? uf0 : (j + 1) / 3 == 1 ? uf1 : uf2) + uo + ((j + 1) % 3) * 512;
                nlo = *reinterpret_cast<const f32x4*>(up); nhi = *reinterpret_cast<const f32x4*>(up + 256);
            }
            const int p = 2 * j;
            const f32x2 v0 = v[p / 6][p % 6], v1 = v[(p + 1) / 6][(p + 1) % 6];
            __builtin_amdgcn_sched_barrier(0);
            if (!(W43_ABL & 4)) {
                if (p < NXA) {
                    mfma_acc_a(acc[p][0], alo[0], v0[0]); mfma_acc_a(acc[p][1], ahi[0], v0[0]);
                    mfma_acc_a(acc[p + 1][0], alo[2], v1[0]); mfma_acc_a(acc[p + 1][1], ahi[2], v1[0]);
                } else {
                    mfma_acc_v(accv[p - NXA][0], alo[0], v0[0]); mfma_acc_v(accv[p - NXA][1], ahi[0], v0[0]);
                    mfma_acc_v(accv[p + 1 - NXA][0], alo[2], v1[0]); mfma_acc_v(accv[p + 1 - NXA][1], ahi[2], v1[0]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (!(W43_ABL & 4)) {
                if (p < NXA) {
                    mfma_acc_a(acc[p][0], alo[1], v0[1]); mfma_acc_a(acc[p][1], ahi[1], v0[1]);
                    mfma_acc_a(acc[p + 1][0], alo[3], v1[1]); mfma_acc_a(acc[p + 1][1], ahi[3], v1[1]);
                } else {
                    mfma_acc_v(accv[p - NXA][0], alo[1], v0[1]); mfma_acc_v(accv[p - NXA][1], ahi[1], v0[1]);
                    mfma_acc_v(accv[p + 1 - NXA][0], alo[3], v1[1]); mfma_acc_v(accv[p + 1 - NXA][1], ahi[3], v1[1]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            alo = nlo; ahi = nhi;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the requests of my transform half have landed (long ago)
        __syncthreads();
    }
    if (W43P_SKEW && !role) __syncthreads();

    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    // ---- epilogue: Y = A^T M A = sum over the rows a of M of (A^T)[:, a] (M[a, :] A): this wave's three rows give a partial 4 x 4 tile per
    //      (channel half, accumulator register); the half 1 - role goes to the partner through LDS, the half role is finished here.
    //      A^T: y0 = m0 + m1 + m2 + m3 + m4;  y1 = PA (m1 - m2) + PB (m3 - m4);  y2 = A2 (m1 + m2) + B2 (m3 + m4);  y3 = A3 (m1 - m2) + B3 (m3 - m4) + m5
    //      local rows (l0, l1, l2) = (m0, m1, m2) or (m5, m3, m4):  s = l1 + l2, d = l1 - l2;  y0 = s + k0 l0, y1 = c1 d, y2 = c2 s, y3 = c3 d + k3 l0
    const float c1 = role ? PB : PA, c2 = role ? B2 : A2, c3 = role ? B3 : A3, k0 = role ? 0.f : 1.f, k3 = role ? 1.f : 0.f;
    float* const xw = smem + (size_t)w * 4096 + lane * 4;          // [wave][16 f32x4][64 lanes]
    float* const xr = smem + (size_t)(w ^ W43P_PAIR) * 4096 + lane * 4;
    float yv[4][16];
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        float part[4][16];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float tm[4][6];
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                // pass 0: the half that goes to the partner (1 - role), pass 1: the half finished here (role)
                const float l0 = ((pass != 0) == (role != 0) ? A(c, 1) : A(c, 0))[j];
                const float l1 = ((pass != 0) == (role != 0) ? A(6 + c, 1) : A(6 + c, 0))[j];
                const float l2 = ((pass != 0) == (role != 0) ? A(12 + c, 1) : A(12 + c, 0))[j];
                const float sm = l1 + l2, df = l1 - l2;
                tm[0][c] = sm + k0 * l0;
                tm[1][c] = c1 * df;
                tm[2][c] = c2 * sm;
                tm[3][c] = c3 * df + k3 * l0;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
                at6(tm[i][0], tm[i][1], tm[i][2], tm[i][3], tm[i][4], tm[i][5], part[j][4 * i], part[j][4 * i + 1], part[j][4 * i + 2],
                    part[j][4 * i + 3]);
        }
        if (pass == 0) {
#pragma unroll
            for (int p = 0; p < 16; ++p)
                *reinterpret_cast<f32x4*>(xw + p * 256) = f32x4{part[0][p], part[1][p], part[2][p], part[3][p]};
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int p = 0; p < 16; ++p) yv[j][p] = part[j][p];
        }
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 16; ++p) {
        const f32x4 o = *reinterpret_cast<const f32x4*>(xr + p * 256);
        yv[0][p] += o[0]; yv[1][p] += o[1]; yv[2][p] += o[2]; yv[3][p] += o[3];
    }
    const int KG = K / 8;
    const float lo = relu ? 0.f : -__builtin_inff();
    if (live) {
        const int kh = role;
        const float4 bv = (FWD && !(chain & 2)) ? *reinterpret_cast<const float4*>(bias_s + 16 * kh + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
        const size_t obase = (((size_t)bimg * KG + kb * 4 + kh * 2 + hf) * H + oy0 + 4 * ty) * W * 8 + (size_t)(ox0 + 4 * tx) * 8 + sub * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jx = 0; jx < 4; ++jx) {
                const int p = 4 * i + jx;
                float4 v4 = make_float4(yv[0][p] + bv.x, yv[1][p] + bv.y, yv[2][p] + bv.z, yv[3][p] + bv.w);
                const size_t o = obase + ((size_t)i * W + jx) * 8;
                if (chain & 1) {
                    const float4 part = *reinterpret_cast<const float4*>(y + o);
                    v4.x += part.x; v4.y += part.y; v4.z += part.z; v4.w += part.w;
                }
                if (chain & 2) {
                } else if (FWD) {
                    v4.x = fmaxf(v4.x, lo); v4.y = fmaxf(v4.y, lo); v4.z = fmaxf(v4.z, lo); v4.w = fmaxf(v4.w, lo);
                } else {
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

// U = G g G^T for every (k, c) in double, rounded once; packed [k / 32][c / 8][xi / 2][(k % 32) / 16][(c % 8) / 2][k % 16][xi % 2][c % 2] (the kernel's
// LDS order: a slice is copied verbatim).  transposed_rot: the backward-data filters g'[c][k][i][j] = g[k][c][2 - i][2 - j].
__global__ void winograd43_prepack_kernel(const float* __restrict__ wgt, float* __restrict__ upack, int Kout, int Cin,
                                          int transposed_rot) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= Kout * Cin) return;
    const int k = e / Cin, c = e % Cin;
    const double pts[5] = {0.0, (double)PA, -(double)PA, (double)PB, -(double)PB};
    double G[6][3];
    for (int i = 0; i < 5; ++i) {
        double f = 1.0;
        for (int j = 0; j < 5; ++j)
            if (j != i) f *= pts[i] - pts[j];
        G[i][0] = 1.0 / f; G[i][1] = pts[i] / f; G[i][2] = pts[i] * pts[i] / f;
    }
    G[5][0] = 0.0; G[5][1] = 0.0; G[5][2] = 1.0;
    double gk[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            gk[i][j] = transposed_rot ? (double)wgt[((size_t)c * Kout + k) * 9 + (2 - i) * 3 + (2 - j)]
                                      : (double)wgt[((size_t)k * Cin + c) * 9 + i * 3 + j];
    double tg[6][3];
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 3; ++j) tg[i][j] = G[i][0] * gk[0][j] + G[i][1] * gk[1][j] + G[i][2] * gk[2][j];
    const int kk = k % KB, cc = c % CC;
    float* base = upack + ((size_t)(k / KB) * (Cin / CC) + c / CC) * UF;
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
            const int xi = 6 * i + j;
            const double u = tg[i][0] * G[j][0] + tg[i][1] * G[j][1] + tg[i][2] * G[j][2];
            base[(xi >> 1) * 512 + (kk >> 4) * 256 + ((cc >> 1) * 16 + (kk & 15)) * 4 + (xi & 1) * 2 + (cc & 1)] = (float)u;
        }
}

}  // namespace

extern "C" int dhz_winograd43_prepack(const float* weight, float* upack, int Kout, int Cin, int transposed_rot, void* stream) {
    DHZ_REQUIRE(weight && upack && Kout % KB == 0 && Cin % 16 == 0, "dhz_winograd43_prepack: Kout=%d Cin=%d", Kout, Cin);
    const int n = Kout * Cin;
    hipLaunchKernelGGL(winograd43_prepack_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, weight, upack, Kout, Cin,
                       transposed_rot);
    DHZ_CHECK_LAUNCH("dhz_winograd43_prepack");
    return DHZ_OK;
}

extern "C" int dhz_winograd43_conv3x3(const float* x, const float* upack, const float* bias, int relu, const float* out_mask,
                                      const float* out_addend, float* y, int B, int H, int W, int C, int K, void* stream) {
    DHZ_REQUIRE(x && upack && y, "dhz_winograd43_conv3x3: null pointer");
    DHZ_REQUIRE(B > 0 && H % 16 == 0 && W % 16 == 0 && H >= 16 && W >= 16 && C % 16 == 0 && K % KB == 0,
                "dhz_winograd43_conv3x3: unsupported shape B=%d H=%d W=%d C=%d K=%d", B, H, W, C, K);
    const bool fwd = !(out_mask || out_addend);
    DHZ_REQUIRE(fwd || !(bias || relu), "dhz_winograd43_conv3x3: bias/relu and out_mask/out_addend are exclusive");
    DHZ_REQUIRE((long long)H * W * 8 * (C / 8) < (1ll << 31), "dhz_winograd43_conv3x3: image too large");
    const int nblk = B * (H / 16) * (W / 16);
    const int grid = ((nblk + 3) / 4) * (K / KB);
    const int xcd_group = (grid % 8 == 0) ? 1 : 0;
    hipStream_t s = (hipStream_t)stream;
    // Accumulation chains: the fp32 rounding of the transform-domain accumulation walks with the square root of the channel count (one
    // chain of 512 channels sits at 0.9 - 1.0 of the kernel tolerance, 256 at 0.55 - 0.7: profiles/r05_winograd_f43_error_table.txt), and no
    // second accumulator fits the register file of this tile.  So more than CHAIN_GROUPS channel groups run as several launches on the
    // stream: all but the last leave raw partial sums in y, all but the first add what they find there (the same lane, the same
    // addresses), the last applies the epilogue.  One extra read + write of the output per extra chain.
    const int CBn = C / CC;
#if W43_FORM == 3
#define W43_KERNEL winograd43_pair_kernel
#define W43_THREADS 512
#define W43_LDS W43P_SMEM
#else
#define W43_KERNEL winograd43_conv3x3_kernel
#define W43_THREADS 256
#define W43_LDS W43_SMEM
#endif
#define GO(F)                                                                                                          \
    do {                                                                                                               \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&W43_KERNEL<F>),                                       \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)W43_LDS);                           \
        for (int cb0 = 0; cb0 < CBn; cb0 += CHAIN_GROUPS) {                                                            \
            const int ncb = CBn - cb0 < CHAIN_GROUPS ? CBn - cb0 : CHAIN_GROUPS;                                       \
            const int chain = (cb0 > 0 ? 1 : 0) | (cb0 + ncb < CBn ? 2 : 0);                                           \
            hipLaunchKernelGGL((W43_KERNEL<F>), dim3(grid), dim3(W43_THREADS), W43_LDS, s, x, upack, bias,             \
                               relu, out_mask, out_addend, y, H, W, C, K, nblk, xcd_group, cb0, ncb, chain);           \
        }                                                                                                              \
    } while (0)
    if (fwd) GO(true); else GO(false);
#undef GO
    DHZ_CHECK_LAUNCH("dhz_winograd43_conv3x3");
    return DHZ_OK;
}
