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
//   (1) transform phase: B^T d B of the lane's 6 x 6 x 2-channel window as 12 six-point transforms of 12 packed FMAs, in place (the
//       window was read from LDS - 36 ds_read_b64 - behind the last matrix instructions of the group before, into registers whose
//       positions were already consumed);
//   (2) matrix phase: 144 MFMAs (A fragments: one ds_read_b128 per position pair and output-channel half); between them the LDS-DMA
//       requests of the NEXT group, one per MFMA quad (issued back to back they queue in the address unit: 100 cycles each), each with
//       its base in SGPRs and a 32-bit lane offset: no vector instruction sits between two MFMAs (one costs 13 - 17 cycles of matrix-pipe
//       time there: tools/ubench/interleave.hip).
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

#ifndef W43_ABL
#define W43_ABL 0       // timing diagnostics: 1 no input transform, 2 no DMA, 4 no MFMAs, 16 window reads but no transform arithmetic
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
// is reused four MFMAs (128 pipe cycles) later, and the epilogue waits 2 x s_nop 15 before it reads them.  The statements are VOLATILE:
// volatile asm statements keep their program order among themselves, so no matrix instruction can sink behind the s_nop pair that ends
// the loop nest (same generated loop as the non-volatile form, instruction for instruction).  What the compiler could still do - copy or
// split the live range of a "v"-class accumulator next to its MFMA - is excluded per code object by tests/test_isa.py: no instruction of the
// channel loop other than the MFMAs touches an accumulator register, and the two s_nop 15 sit between the last MFMA and the first
// accumulator read.
__device__ __forceinline__ void mfma_acc_a(f32x4& acc, float a, float b) {
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_acc_v(f32x4& acc, float a, float b) {
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
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
                                                                    int chain, float* __restrict__ ypool) {
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

    // the last matrix instructions (inline asm: hipcc does not know what they are) have written their accumulators before these are read.
    // The fence CARRIES the accumulators of the last two pair-steps (positions 32 .. 35: the VGPR-class ones, sixteen matrix instructions) as
    // in / out operands: no read or copy of them can be placed in front of the s_nop pair.  The AGPR-class accumulators are at least those
    // sixteen matrix instructions (512 pipe cycles) old at this point - the compiler may, and does, start shuffling them before the fence.
    static_assert(NXI - NXA == 4, "the fence below names the four VGPR-class accumulator positions");
    asm volatile("s_nop 15\n\ts_nop 15"
                 : "+v"(accv[0][0]), "+v"(accv[0][1]), "+v"(accv[1][0]), "+v"(accv[1][1]), "+v"(accv[2][0]), "+v"(accv[2][1]), "+v"(accv[3][0]),
                   "+v"(accv[3][1])
                 :: "memory");
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
            if (FWD && ypool && !(chain & 2)) {
                // the layer feeds a 2 x 2 max pooling and nothing else (VGG19 conv1_2 / 2_2 / 3_4 / 4_4 in a no-gradient pass): the lane's
                // 4 x 4 pixels give 2 x 2 pooled ones - a quarter of the stores, no pooling launch, no full-resolution map
                const int Hp = H >> 1, Wp = W >> 1;
                const size_t pbase = (((size_t)bimg * KG + kb * 4 + kh * 2 + hf) * Hp + (oy0 >> 1) + 2 * ty) * Wp * 8 + (size_t)((ox0 >> 1) + 2 * tx) * 8 + sub * 4;
#pragma unroll
                for (int pi = 0; pi < 2; ++pi)
#pragma unroll
                    for (int pj = 0; pj < 2; ++pj) {
                        float4 m = make_float4(lo, lo, lo, lo);
#pragma unroll
                        for (int a = 0; a < 2; ++a)
#pragma unroll
                            for (int b = 0; b < 2; ++b) {
                                const int p = 4 * (2 * pi + a) + 2 * pj + b;
                                float4 v4 = make_float4(yv[0][p] + bv.x, yv[1][p] + bv.y, yv[2][p] + bv.z, yv[3][p] + bv.w);
                                if (chain & 1) {
                                    const float4 part = *reinterpret_cast<const float4*>(y + obase + ((size_t)(2 * pi + a) * W + 2 * pj + b) * 8);
                                    v4.x += part.x; v4.y += part.y; v4.z += part.z; v4.w += part.w;
                                }
                                m.x = fmaxf(m.x, v4.x); m.y = fmaxf(m.y, v4.y); m.z = fmaxf(m.z, v4.z); m.w = fmaxf(m.w, v4.w);
                            }
                        *reinterpret_cast<float4*>(ypool + pbase + ((size_t)pi * Wp + pj) * 8) = m;
                    }
                continue;
            }
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

// (A third form - 512-thread workgroups whose waves w and w + 4 share a block, split the 36 positions and run their transform / matrix halves
// in opposite phases - was built, correct, and 3 % slower than this one: vector and matrix instructions of different waves of a SIMD
// serialise on this part, tools/ubench/overlap.hip.  DESIGN 4e; the kernel is in the history at commit 7323fe2.)

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

static int winograd43_launch(const char* who, const float* x, const float* upack, const float* bias, int relu, const float* out_mask,
                             const float* out_addend, float* y, float* ypool, int B, int H, int W, int C, int K, void* stream) {
    DHZ_REQUIRE(x && upack && (y || ypool), "%s: null pointer", who);
    DHZ_REQUIRE(B > 0 && H % 16 == 0 && W % 16 == 0 && H >= 16 && W >= 16 && C % 16 == 0 && K % KB == 0,
                "%s: unsupported shape B=%d H=%d W=%d C=%d K=%d", who, B, H, W, C, K);
    const bool fwd = !(out_mask || out_addend);
    DHZ_REQUIRE(fwd || !(bias || relu), "%s: bias/relu and out_mask/out_addend are exclusive", who);
    DHZ_REQUIRE((long long)H * W * 8 * (C / 8) < (1ll << 31), "%s: image too large", who);
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
    DHZ_REQUIRE(y || CBn <= CHAIN_GROUPS, "%s: %d input channels run as accumulation chains and need the full-resolution scratch y", who, C);
#define GO(F)                                                                                                          \
    do {                                                                                                               \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&winograd43_conv3x3_kernel<F>),                        \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)W43_SMEM);                          \
        for (int cb0 = 0; cb0 < CBn; cb0 += CHAIN_GROUPS) {                                                            \
            const int ncb = CBn - cb0 < CHAIN_GROUPS ? CBn - cb0 : CHAIN_GROUPS;                                       \
            const int chain = (cb0 > 0 ? 1 : 0) | (cb0 + ncb < CBn ? 2 : 0);                                           \
            hipLaunchKernelGGL((winograd43_conv3x3_kernel<F>), dim3(grid), dim3(256), W43_SMEM, s, x, upack, bias,     \
                               relu, out_mask, out_addend, y, H, W, C, K, nblk, xcd_group, cb0, ncb, chain, ypool);    \
        }                                                                                                              \
    } while (0)
    if (fwd) GO(true); else GO(false);
#undef GO
    DHZ_CHECK_LAUNCH(who);
    return DHZ_OK;
}

extern "C" int dhz_winograd43_conv3x3(const float* x, const float* upack, const float* bias, int relu, const float* out_mask,
                                      const float* out_addend, float* y, int B, int H, int W, int C, int K, void* stream) {
    DHZ_REQUIRE(y, "dhz_winograd43_conv3x3: null pointer");
    return winograd43_launch("dhz_winograd43_conv3x3", x, upack, bias, relu, out_mask, out_addend, y, nullptr, B, H, W, C, K, stream);
}

// convolution + bias + ReLU + 2 x 2 max pooling in one launch: ypool [B][K/8][H/2][W/2][8].  scratch (full resolution, [B][K/8][H][W][8]) is
// needed only when C > 256 (the partial sums of the accumulation chains travel through it); may be null otherwise.
extern "C" int dhz_winograd43_conv3x3_pool(const float* x, const float* upack, const float* bias, float* ypool, float* scratch, int B, int H,
                                           int W, int C, int K, void* stream) {
    DHZ_REQUIRE(ypool, "dhz_winograd43_conv3x3_pool: null pointer");
    return winograd43_launch("dhz_winograd43_conv3x3_pool", x, upack, bias, 1, nullptr, nullptr, scratch, ypool, B, H, W, C, K, stream);
}
