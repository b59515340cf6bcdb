// The product's fp32 token-Linear GEMMs (forward, backward-data) on the bf16 matrix pipe - six-term operand splitting.
//     forward        y[T,N]  = x[T,K]  . W[N,K]^T + b
//     backward-data  dx[T,K] = dy[T,N] . W[N,K]
// Arithmetic: every fp32 operand value is cut by TRUNCATION into three bf16 pieces,
//     x = hi + mid + lo        (each piece takes the next eight significant bits: all 24, exactly)
// and a product is taken as  a.b ~ hh + (hm + mh) + (hl + lh + mm):  six v_mfma_f32_16x16x32_bf16 with fp32 accumulation, small
// terms first.  What is dropped (ml, lm, ll) is <= 2^-24 relative per product - the size of ONE fp32 rounding - so a result is in
// the error class of an fp32 FMA chain (not bit-identical to one: other grouping of the sums) at 6 / 16 of the fp32 pipe's
// matrix time.  Operands, results, bias and accumulation stay fp32.
//
// Round 4 structure (csrc/linear_split.hip is the round-3 form: both operands split inside the kernel, one LDS stage, two
// barriers per stage with the split between them):
//   * the WEIGHTS arrive pre-split: three bf16 planes that mirror the flat fp32 parameter buffer (written by one dhz_split3_planes
//     launch after every optimizer step, and again after an outside write).  They go global -> LDS by
//     LDS-DMA (global_load_lds_dwordx4: no VGPR, no VALU, no ds_write); the bank swizzle is carried by the per-lane SOURCE
//     address.  Backward-data reads the same planes as they lie ([N][K], the contraction index is the row) through the
//     hardware transpose ds_read_b64_tr_b16.
//   * only the ACTIVATION operand is split in the kernel: global -> registers one stage ahead, 4.5 VALU instructions per
//     element, three ds_write_b128 per 8 elements - placed INSIDE the MFMA stream of the previous stage (a bf16 MFMA holds the
//     vector issue for 8 of its 16 cycles: a wave's own vector instructions ride in the other 8).
//   * 512-thread workgroups, one per CU, 32-deep contraction stages in a three-slot LDS ring, ONE raw s_barrier per stage; the
//     persistent workgroup treats its (tile, stage) pairs as one stream, so a tile's stores drain behind the next tile's
//     matrix work (counted vmcnt waits).
//   * LDS images: [row][32 k] bf16 = 64-byte rows, the four 16-byte chunks of a row XOR-ed with P[(row >> 2) & 3],
//     P = {0, 2, 3, 1}: a lane's 8 consecutive k are ONE conflict-free ds_read_b128 (checked against the b128 lane groups of
//     MI355X_MICROARCH.md).  Transposed form: [32 k][BN] bf16 with the 16-byte-chunk swizzles of csrc/linear_bf16.hip.
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "tok_epilogue.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

constexpr int BK = 32;          // contraction elements per stage
constexpr int NT = 512;         // threads per workgroup (8 waves)

#ifndef DHZ_S6_ABL
#define DHZ_S6_ABL 0            // timing diagnostics: 1 = no stores, 2 = no MFMAs, 4 = no activation split, 8 = no weight DMA, 16 = no fragment reads
#endif

__device__ __forceinline__ f32x4 mfma_bf16(s16x8 a, s16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// (Non-temporal result stores were tried on the 256 x 128 kernel for results of 192 MB and more: in isolation T = 524288, 64 -> 256:
// 170 -> 146 us, backward-data 256 <- 64: 180 -> 154 - but the training step did not move (35.11 / 35.26 ms without, 35.17 / 35.30 with):
// what the store saves, the kernel that reads the result next pays.  Not used.)
__device__ __forceinline__ void dma16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((glb_void*)g, (lds_void*)l, 16, 0, 0);
}
// byte offset of 16-byte chunk ch (0..3) of row `row` in a 64-byte-row image
__device__ __forceinline__ int swz64(int row) { return (0x78 >> (2 * ((row >> 2) & 3))) & 3; }
__device__ __forceinline__ int off64(int row, int ch) { return row * 64 + 16 * (ch ^ swz64(row)); }
// transposed images (rows = contraction index, F output features per row): swizzles of csrc/linear_bf16.hip
template <int F>
__device__ __forceinline__ int swz_tr(int row) {
    if (F == 128) return ((row & 3) << 2) | ((row >> 2) & 3);
    return (row & 2) | ((row & 8) >> 1);
}
template <int F>
__device__ __forceinline__ int off_tr(int row, int ch) { return row * (2 * F) + 16 * (ch ^ swz_tr<F>(row)); }
template <int F>
__device__ __forceinline__ s16x8 tr_frag(const unsigned char* img, int r0, int cb, int lane) {
    const int m = lane & 15, q = m >> 2, p = m & 3;
    typedef s16x4 __attribute__((address_space(3))) * lds_ptr;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(img + off_tr<F>(r0 + q, 2 * cb + (p >> 1)) + 8 * (p & 1)));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(img + off_tr<F>(r0 + 4 + q, 2 * cb + (p >> 1)) + 8 * (p & 1)));
    return s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// eight fp32 values -> three bf16 pieces each, by truncation: a bf16 is the top 16 bits of the fp32 pattern, so each piece
// takes the next eight significant bits of what is left (x - hi and r - mid are exact).  Per pair of elements: two masks and two
// subtractions per level, one v_perm_b32 per piece to pack the two top halves.
__device__ __forceinline__ uint32_t pack_top(float x1, float x0) {              // (x1 & 0xffff0000) | (x0 >> 16)
    return __builtin_amdgcn_perm(__float_as_uint(x1), __float_as_uint(x0), 0x07060302u);
}
// Between two bf16 MFMAs of a wave the first two plain vector instructions cost nothing and further ones 4 cycles each, a PACKED fp32
// instruction 7 - 17 cycles of matrix-pipe time (tools/ubench/interleave.hip): the subtractions are therefore issued per lane - 11
// instructions per pair of elements instead of 9 - through inline asm (left as C, hipcc's SLP vectoriser re-packs them).
__device__ __forceinline__ float sub_np(float a, float b) {
    float r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ void split8x3(const f32x4 a, const f32x4 b, u32x4& hi, u32x4& mid, u32x4& lo) {
    const float x[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float x0 = x[2 * i], x1 = x[2 * i + 1];
        const float r0 = sub_np(x0, __uint_as_float(__float_as_uint(x0) & 0xffff0000u)), r1 = sub_np(x1, __uint_as_float(__float_as_uint(x1) & 0xffff0000u));
        const float q0 = sub_np(r0, __uint_as_float(__float_as_uint(r0) & 0xffff0000u)), q1 = sub_np(r1, __uint_as_float(__float_as_uint(r1) & 0xffff0000u));
        hi[i] = pack_top(x1, x0);
        mid[i] = pack_top(r1, r0);
        lo[i] = pack_top(q1, q0);
    }
}

// Activation loads as inline assembly: hipcc's wait-count pass does not see them, so it cannot answer a use of their registers
// with vmcnt(0) (which would also wait for the LDS-DMA issued since).  The wait is explicit and counted - wait_regs<N> leaves the
// N youngest vector-memory operations in flight - and takes the registers as in/out operands, so no use can move above it.
__device__ __forceinline__ void gload32(const float* p, f32x4& v0, f32x4& v1) {
    asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:16" : "=&v"(v0), "=&v"(v1) : "v"(p) : "memory");
}
template <int N> __device__ __forceinline__ void wait_regs(f32x4& a, f32x4& b) {
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory");
}
template <int N> __device__ __forceinline__ void wait_regs(f32x4& a, f32x4& b, f32x4& c, f32x4& d) {
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory"); }

// C[M,NF] = A[M,KC] . op(B) (+ bias).  BTR = false: B planes are [NF][KC] (forward); true: [KC][NF] (backward-data).
// Wave tile 16 WM tokens x 16 WN features; the eight waves as WAVES_M x (8 / WAVES_M).
//
// Software pipeline (stage q = 32 contraction elements of one tile; the workgroup's (tile, stage) pairs form one stream).
// LDS: three-slot rings for both operands' piece images (stage q in slot q % 3); raw activations of two stages in registers
// (stage q in set q & 1); two fragment sets (q & 1).  Iteration p, between barrier(p-1) and barrier(p):
//     two instruction streams of MFMAs (lower / upper tile rows of stage p) with everything else of the iteration riding BETWEEN the
//     MFMAs (a bf16 MFMA holds the vector issue for 8 of its 16 cycles; a wave's own vector / LDS / memory instructions go into
//     the other 8 - a partner wave's get one slot per ~20 cycles; stamps: 1500 cycles for 60 instructions):
//         stream 1: DMA of the weight pieces of stage p+3 -> slot p % 3 (stage p's weight fragments are in registers since
//                   iteration p-1), ds_read of stage p's upper-row activation fragments (complete since barrier(p-2)), split of
//                   stage p+2's raw activations (registers loaded in iteration p-2)
//         between : load of the raw activations of stage p+4 -> the register set just consumed
//         stream 2: ds_write of stage p+2's pieces -> slot (p+2) % 3, ds_read of stage p+1's weight fragments and lower-row
//                   activation fragments (complete since barrier(p-1))
//     stores   of a finished tile
//     wait     the DMA of stage p+2 (issued in iteration p-1) has landed; my LDS writes / reads are done
//     barrier(p)
// so every global access has two stages of matrix time to land, no fragment read waits on a barrier, and the barrier is only
// the hand-over of ring slots.  The loop is unrolled six times (ring slots, register sets and fragment sets are compile-time).
// EPI: the epilogue is the block's residual step (csrc/tok_epilogue.h): C[dst(m)] = res[dst(m)] + scale[img(m)] (acc + bias).
template <int WM, int WN, int WAVES_M, bool BTR, bool EPI>
__global__ __launch_bounds__(NT, 1) void split6_gemm_kernel(const float* __restrict__ A, int lda, const uint16_t* __restrict__ Bh,
                                                            const uint16_t* __restrict__ Bm, const uint16_t* __restrict__ Bl,
                                                            int ldb, const float* __restrict__ bias, float* __restrict__ C, int ldc,
                                                            int M, int NF, int KC, int tiles_n, int ntiles, const TokEpi epi) {
    constexpr int WAVES_N = 8 / WAVES_M;
    constexpr int BM = 16 * WM * WAVES_M, BN = 16 * WN * WAVES_N;
    constexpr int A_BYTES = BM * 64, B_BYTES = BN * 64;            // one piece of one stage
    constexpr int A_SLOT = 3 * A_BYTES, B_SLOT = 3 * B_BYTES;      // one ring slot: hi, mid, lo images
    constexpr int NA = BM / 128;                                   // 8-element chunks per thread and stage
    constexpr int B_KIB = B_BYTES / 1024;                          // DMA wave-instructions per piece
    constexpr int NDMA = 3 * B_KIB;                                // ... per stage and workgroup
    constexpr int DPW = (NDMA + 7) / 8;                            // ... per wave
    constexpr int HALF = WM > 1 ? WM / 2 : WM;                     // activation fragment rows [0, HALF) are the "lower" ones
    constexpr int abl = DHZ_S6_ABL;
    static_assert(BM % 128 == 0 && (BN == 32 || BN == 64 || BN == 128), "tile shape");
    static_assert(!BTR || BN >= 64, "transposed weight images: 64 or 128 features per tile");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const Aring = smem;
    unsigned char* const Bring = smem + 3 * A_SLOT;
    float* const bsm = reinterpret_cast<float*>(smem + 3 * A_SLOT + 3 * B_SLOT);     // the bias vector (zeros without one)

    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int i16 = lane & 15, g = lane >> 4;
    const int wm = w / WAVES_N, wn = w % WAVES_N;
    const int nst = KC / BK;
    const int grid = gridDim.x;
    auto tile_of = [&](int i) -> int {
        const int lin = blockIdx.x + i * grid;
        if (lin >= ntiles) return -1;
        if ((grid & 7) == 0 && (ntiles & 7) == 0) return (lin & 7) * (ntiles >> 3) + (lin >> 3);
        return lin;
    };

    // ---- activation operand: thread -> (row, chunk) of the [BM][32 k] stage
    int a_row[NA], a_kc[NA], a_lds[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int c = t + NT * i;
        a_row[i] = c >> 2;
        a_kc[i] = c & 3;
        a_lds[i] = off64(a_row[i], a_kc[i]);
    }
    // ---- weight operand: this wave's DMA instructions q = w + 8 j -> (piece, KiB of the piece image); per-lane source offset
    int b_off[DPW];                 // elements, relative to (n0, k0) of the stage
    int b_dst[DPW];                 // byte offset of the KiB inside a ring slot
    int b_piece[DPW];
#pragma unroll
    for (int j = 0; j < DPW; ++j) {
        const int q = (w + 8 * j) % NDMA;               // (waves past the end repeat an earlier piece: same bytes, same place)
        const int piece = q / B_KIB, sub = q % B_KIB;
        b_piece[j] = piece;
        b_dst[j] = piece * B_BYTES + sub * 1024;
        if (BTR) {
            constexpr int CPR = BN / 8;                            // 16-byte chunks per image row
            const int row = sub * (64 / CPR) + lane / CPR, pch = lane % CPR;
            b_off[j] = row * ldb + 8 * (pch ^ swz_tr<BN>(row));
        } else {
            const int row = 16 * sub + (lane >> 2), pch = lane & 3;
            b_off[j] = row * ldb + 8 * (pch ^ swz64(row));
        }
    }
    const uint16_t* const planes[3] = {Bh, Bm, Bl};

    // (tile, stage) position of the stream; the tile's row / column origin is derived once per tile (an integer division per stage
    // and use costs ~20 scalar instructions of an in-order wave)
    struct Pos { int tile, st, ti, m0, n0; };
    auto at_tile = [&](int ti) -> Pos {
        const int tile = tile_of(ti);
        const int tm = tile < 0 ? 0 : tile / tiles_n;
        return Pos{tile, 0, ti, tm * BM, (tile - tm * tiles_n) * BN};
    };
    auto next = [&](const Pos& p) -> Pos {
        if (p.tile < 0) return p;
        if (p.st + 1 < nst) return Pos{p.tile, p.st + 1, p.ti, p.m0, p.n0};
        return at_tile(p.ti + 1);
    };
    const Pos pz = at_tile(0);
    if (pz.tile < 0) return;
    // past the end of the stream every load / DMA re-fetches the workgroup's first stage (into slots and registers nobody reads
    // any more): the loop body carries no tests around its memory operations
    auto valid = [&](const Pos& p) -> Pos { return p.tile >= 0 ? p : pz; };

    f32x4 ra[2][NA][2];                                            // raw activations: stage q in set q & 1
    auto a_addr = [&](const Pos& q, int i) -> const float* {
        const int m0 = q.m0, k0 = q.st * BK;
        return A + (size_t)min(m0 + a_row[i], M - 1) * lda + k0 + 8 * a_kc[i];
    };
    auto a_load = [&](auto set, const Pos& q) {
        constexpr int S = decltype(set)::value;
#pragma unroll
        for (int i = 0; i < NA; ++i) gload32(a_addr(q, i), ra[S][i][0], ra[S][i][1]);
    };
    auto a_wait = [&](auto set, auto tag) {                        // set's loads are complete; the tag's count of younger operations stays in flight
        constexpr int S = decltype(set)::value, N = decltype(tag)::value;
        if constexpr (NA == 1) wait_regs<N>(ra[S][0][0], ra[S][0][1]);
        else wait_regs<N>(ra[S][0][0], ra[S][0][1], ra[S][1][0], ra[S][1][1]);
    };
    auto a_split_write = [&](auto set, int slot) {
        constexpr int S = decltype(set)::value;
        if (abl & 4) return;
        unsigned char* As = Aring + slot * A_SLOT;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            u32x4 hi, mid, lo;
            split8x3(ra[S][i][0], ra[S][i][1], hi, mid, lo);
            *reinterpret_cast<u32x4*>(As + a_lds[i]) = hi;
            *reinterpret_cast<u32x4*>(As + A_BYTES + a_lds[i]) = mid;
            *reinterpret_cast<u32x4*>(As + 2 * A_BYTES + a_lds[i]) = lo;
        }
    };
    auto b_dma = [&](const Pos& q, int slot) {
        if (abl & 8) return;
        const int n0 = q.n0, k0 = q.st * BK;
        const size_t base = BTR ? (size_t)k0 * ldb + n0 : (size_t)n0 * ldb + k0;
        unsigned char* Bs = Bring + slot * B_SLOT;
#pragma unroll
        for (int j = 0; j < DPW; ++j) dma16(planes[b_piece[j]] + base + b_off[j], Bs + b_dst[j]);
    };

    f32x4 acc[WM][WN];
#pragma unroll
    for (int a = 0; a < WM; ++a)
#pragma unroll
        for (int b = 0; b < WN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = t; i < NF; i += NT) bsm[i] = bias ? bias[i] : 0.f;     // visible after the first barrier below

    // fragment sets: [set][piece][tile row / column]
    s16x8 af[2][3][WM] = {}, bf[2][3][WN] = {};
    auto read_a = [&](auto set, int slot, int a0, int a1) {
        constexpr int F = decltype(set)::value;
        if (abl & 16) return;
        const unsigned char* As = Aring + slot * A_SLOT;
#pragma unroll
        for (int pc = 0; pc < 3; ++pc)
#pragma unroll
            for (int a = a0; a < a1; ++a)
                af[F][pc][a] = *reinterpret_cast<const s16x8*>(As + pc * A_BYTES + off64((wm * WM + a) * 16 + i16, g));
    };
    auto read_b = [&](auto set, int slot) {
        constexpr int F = decltype(set)::value;
        if (abl & 16) return;
        const unsigned char* Bs = Bring + slot * B_SLOT;
#pragma unroll
        for (int pc = 0; pc < 3; ++pc)
#pragma unroll
            for (int b = 0; b < WN; ++b) {
                if (BTR) bf[F][pc][b] = tr_frag<BN>(Bs + pc * B_BYTES, 8 * g, wn * WN + b, lane);
                else bf[F][pc][b] = *reinterpret_cast<const s16x8*>(Bs + pc * B_BYTES + off64((wn * WN + b) * 16 + i16, g));
            }
    };
    // the six products of every tile of rows [a0, a1), small terms first; term-major, so that consecutive MFMAs never wait for one
    // another's accumulator
    auto mma_rows = [&](auto set, int a0, int a1) {
        constexpr int F = decltype(set)::value;
        constexpr int TA[6] = {0, 2, 1, 0, 1, 0}, TB[6] = {2, 0, 1, 1, 0, 0};      // (activation piece, weight piece): lh hl mm hm mh hh
#pragma unroll
        for (int term = 0; term < 6; ++term)
#pragma unroll
            for (int a = a0; a < a1; ++a)
#pragma unroll
                for (int b = 0; b < WN; ++b) {
                    if (abl & 2) { acc[a][b][0] += (float)(af[F][TA[term]][a][0] + bf[F][TB[term]][b][0]); continue; }
                    acc[a][b] = mfma_bf16(bf[F][TB[term]][b], af[F][TA[term]][a], acc[a][b]);
                }
    };
    // acc[a][b][j] = C[token 16 a + i16][feature 16 b + 4 g + j]  (D = C^T block: the weight fragment is the first MFMA operand)
    auto epilogue = [&](int tile) -> bool {
        const int tn = tile % tiles_n, tm = tile / tiles_n;
        const int m0 = tm * BM + wm * WM * 16 + i16, n0 = tn * BN + wn * WN * 16 + 4 * g;
        float* c0 = C + (size_t)m0 * ldc + n0;
        f32x4 bv[WN];
#pragma unroll
        for (int b = 0; b < WN; ++b) bv[b] = *reinterpret_cast<const f32x4*>(bsm + n0 + 16 * b);
        if constexpr (EPI) {
            // the shortcut values are requested here, behind this iteration's operand requests: the wait in front of the stores
            // drains those too (vmcnt retires in order) - about one memory latency per tile, a few per cent of its matrix time
            const int mbw = tm * BM + wm * WM * 16;                  // wave-uniform
            int dst[WM];
            float sc;
            tok_epi_rows<WM>(epi, mbw < M ? mbw : 0, i16, dst, sc);
            const int nn = tn * BN + wn * WN * 16 + 4 * g;
            f32x4 rv[WM][WN];
#pragma unroll
            for (int a = 0; a < WM; ++a)
#pragma unroll
                for (int b = 0; b < WN; ++b)
                    rv[a][b] = epi.res ? *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(epi.res) + (size_t)min(dst[a], M - 1) * ldc + nn + 16 * b)
                                       : f32x4{0.f, 0.f, 0.f, 0.f};
            const bool efull = tm * BM + BM <= M;                    // wave-uniform; WM * WN stores when true (the caller counts them)
#pragma unroll
            for (int a = 0; a < WM; ++a)
                if (efull || m0 + 16 * a < M) {
#pragma unroll
                    for (int b = 0; b < WN; ++b)
                        *reinterpret_cast<f32x4*>(C + (size_t)dst[a] * ldc + nn + 16 * b) = rv[a][b] + sc * (acc[a][b] + bv[b]);
                }
#pragma unroll
            for (int a = 0; a < WM; ++a)
#pragma unroll
                for (int b = 0; b < WN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
            return efull;
        }
        const bool full = tm * BM + BM <= M;                      // wave-uniform
        if (full) {
#pragma unroll
            for (int a = 0; a < WM; ++a)
#pragma unroll
                for (int b = 0; b < WN; ++b)
                    if (!(abl & 1) || acc[a][b][0] == 12345.678f)
                        *reinterpret_cast<f32x4*>(c0 + (size_t)(16 * a) * ldc + 16 * b) = acc[a][b] + bv[b];
        } else {
#pragma unroll
            for (int a = 0; a < WM; ++a)
                if (m0 + 16 * a < M) {
#pragma unroll
                    for (int b = 0; b < WN; ++b) *reinterpret_cast<f32x4*>(c0 + (size_t)(16 * a) * ldc + 16 * b) = acc[a][b] + bv[b];
                }
        }
#pragma unroll
        for (int a = 0; a < WM; ++a)
#pragma unroll
            for (int b = 0; b < WN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        return full;
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    // ---- prologue: stages 0 and 1 complete in LDS, stage 2's weight DMA and the raw activations of stages 2 and 3 in flight,
    //      stage 0's weight fragments and lower-row activation fragments in registers
    Pos p0 = pz;
    Pos p1 = next(p0), p2 = next(p1), p3 = next(p2);
    b_dma(p0, 0);
    b_dma(valid(p1), 1);
    a_load(I0{}, p0);
    a_load(I1{}, valid(p1));
    a_wait(I0{}, I0{});
    a_wait(I1{}, I0{});
    a_split_write(I0{}, 0);
    a_split_write(I1{}, 1);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    a_load(I0{}, valid(p2));                                      // issue order as in the loop (the counted waits rely on it):
    b_dma(valid(p2), 2);                                          // loads of stage q, the DMA of stage q, loads of stage q + 1
    a_load(I1{}, valid(p3));
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    read_b(I0{}, 0);
    read_a(I0{}, 0, 0, HALF);
    // (in the loop a stage's weight fragments are read one iteration ahead, i.e. in front of a barrier; here every wave must
    // hold them before the first iteration's DMA overwrites slot 0)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    Pos p4 = next(p3);

    bool done = false;
    auto body = [&](auto rtag) {
        constexpr int R = decltype(rtag)::value;                   // p % 6
        using F = std::integral_constant<int, R & 1>;              // fragment set of stage p; raw register set of stage p + 2
        using FN = std::integral_constant<int, (R + 1) & 1>;       // fragment set of stage p + 1
        constexpr int S0 = R % 3, S1 = (R + 1) % 3, S2 = (R + 2) % 3;
        const bool have1 = p1.tile >= 0;
        a_wait(F{}, std::integral_constant<int, DPW + 2 * NA>{});  // stage p + 2's raw activations (set p & 1): older than one DMA set + 2 NA loads
        __builtin_amdgcn_sched_barrier(0);
        // ---- first stream: the lower rows' MFMAs; between them the DMA of stage p + 3 -> slot p % 3, the upper-row fragment
        //      reads of stage p, the split of stage p + 2 and the address arithmetic of the loads below
        b_dma(valid(p3), S0);
        if (HALF < WM) read_a(F{}, S0, HALF, WM);
        const float* pa[NA];
#pragma unroll
        for (int i = 0; i < NA; ++i) pa[i] = a_addr(valid(p4), i);
        u32x4 sh[NA], sm[NA], sl[NA];
#pragma unroll
        for (int i = 0; i < NA; ++i) split8x3(ra[R & 1][i][0], ra[R & 1][i][1], sh[i], sm[i], sl[i]);
        mma_rows(F{}, 0, HALF);
#pragma unroll
        for (int i = 0; i < HALF * WN * 6; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // one MFMA
            __builtin_amdgcn_sched_group_barrier(0x3b2, 2, 0);     // then up to two vector / LDS / vector-memory instructions
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NA; ++i) gload32(pa[i], ra[R & 1][i][0], ra[R & 1][i][1]);      // stage p + 4 -> the set just consumed
        __builtin_amdgcn_sched_barrier(0);
        // ---- second stream: the upper rows' MFMAs; between them the LDS writes of stage p + 2 -> slot (p + 2) % 3 and the reads of
        //      stage p + 1's weight fragments and lower-row activation fragments
        if (!(abl & 4)) {
            unsigned char* As = Aring + S2 * A_SLOT;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                *reinterpret_cast<u32x4*>(As + a_lds[i]) = sh[i];
                *reinterpret_cast<u32x4*>(As + A_BYTES + a_lds[i]) = sm[i];
                *reinterpret_cast<u32x4*>(As + 2 * A_BYTES + a_lds[i]) = sl[i];
            }
        }
        read_b(FN{}, S1);
        read_a(FN{}, S1, 0, HALF);
        if (HALF < WM) mma_rows(F{}, HALF, WM);
#pragma unroll
        for (int i = 0; i < (WM - HALF) * WN * 6; ++i) {         // the LDS instructions early: their latency under the rest of the stream
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x382, 2, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        const bool stored = p0.st == nst - 1;
        bool full = true;
        if (stored) full = epilogue(p0.tile);
        if (!have1) { done = true; return; }
        __builtin_amdgcn_sched_barrier(0);
        // the DMA of stage p + 2 (issued one iteration ago) has landed: it is older than 2 NA loads + one DMA set + 2 NA loads and
        // this iteration's stores (vmcnt retires in order); my LDS writes and reads are complete; then everybody's
        if (!stored) wait_vm<4 * NA + DPW>();
        else if (full) wait_vm<4 * NA + DPW + WM * WN>();
        else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        p0 = p1; p1 = p2; p2 = p3; p3 = p4; p4 = next(p4);
    };
    while (true) {
        body(std::integral_constant<int, 0>{}); if (done) break;
        body(std::integral_constant<int, 1>{}); if (done) break;
        body(std::integral_constant<int, 2>{}); if (done) break;
        body(std::integral_constant<int, 3>{}); if (done) break;
        body(std::integral_constant<int, 4>{}); if (done) break;
        body(std::integral_constant<int, 5>{}); if (done) break;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // no LDS-DMA may be in flight when the workgroup's LDS is released
}

// ---- the wide-tile form: 256 tokens x 128 features per workgroup, 64 x 64 per wave (eight waves as 4 x 2).
// Measured on the 128 x 128 kernel above (phase ablation, in-kernel stamps, tools/micro/lds_read_rate.hip): with three piece images per
// operand the LDS carries 18 fragment reads per 48 MFMAs and wave - its time (reads ~710 + writes ~400 cycles per stage and CU)
// is close to the matrix time (1536), the L2 delivers the operand stages at ~32 B/clk/CU (1250 cycles per stage) and these
// add up to the matrix time instead of hiding under it.  A 64 x 64 wave tile needs 24 reads per 96 MFMAs (2/3 of the LDS reads
// per product) and a 256 x 128 workgroup tile 0.7 x the L2 bytes per product; a stage then carries 96 MFMAs per wave (two
// waves per SIMD: ~3000 cycles), long enough for a plain two-slot ring: the DMA of stage p+1 and the split of its activations
// go into the other slot during stage p and have the whole stage to land.
template <bool BTR, bool EPI>
__global__ __launch_bounds__(NT, 1) void split6_wide_kernel(const float* __restrict__ A, int lda, const uint16_t* __restrict__ Bh,
                                                            const uint16_t* __restrict__ Bm, const uint16_t* __restrict__ Bl,
                                                            int ldb, const float* __restrict__ bias, float* __restrict__ C, int ldc,
                                                            int M, int NF, int KC, int tiles_n, int ntiles, const TokEpi epi) {
    constexpr int WM = 4, WN = 4, WAVES_N = 2;
    constexpr int BM = 256, BN = 128;
    constexpr int A_BYTES = BM * 64, B_BYTES = BN * 64;            // one piece of one stage
    constexpr int SLOT = 3 * (A_BYTES + B_BYTES);                  // hi, mid, lo images of both operands
    constexpr int NA = BM / 128;                                   // 8-element chunks per thread and stage
    constexpr int B_KIB = B_BYTES / 1024, NDMA = 3 * B_KIB, DPW = NDMA / 8;
    static_assert(NDMA % 8 == 0, "whole DMA instructions per wave");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* const bsm = reinterpret_cast<float*>(smem + 2 * SLOT);

    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int i16 = lane & 15, g = lane >> 4;
    const int wm = w / WAVES_N, wn = w % WAVES_N;
    const int nst = KC / BK;
    const int grid = gridDim.x;
    auto tile_of = [&](int i) -> int {
        const int lin = blockIdx.x + i * grid;
        if (lin >= ntiles) return -1;
        if ((grid & 7) == 0 && (ntiles & 7) == 0) return (lin & 7) * (ntiles >> 3) + (lin >> 3);
        return lin;
    };
    int a_row[NA], a_kc[NA], a_lds[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int c = t + NT * i;
        a_row[i] = c >> 2;
        a_kc[i] = c & 3;
        a_lds[i] = off64(a_row[i], a_kc[i]);
    }
    int b_off[DPW], b_dst[DPW], b_piece[DPW];
#pragma unroll
    for (int j = 0; j < DPW; ++j) {
        const int q = w + 8 * j;
        const int piece = q / B_KIB, sub = q % B_KIB;
        b_piece[j] = piece;
        b_dst[j] = 3 * A_BYTES + piece * B_BYTES + sub * 1024;
        if (BTR) {
            constexpr int CPR = BN / 8;
            const int row = sub * (64 / CPR) + lane / CPR, pch = lane % CPR;
            b_off[j] = row * ldb + 8 * (pch ^ swz_tr<BN>(row));
        } else {
            const int row = 16 * sub + (lane >> 2), pch = lane & 3;
            b_off[j] = row * ldb + 8 * (pch ^ swz64(row));
        }
    }
    const uint16_t* const planes[3] = {Bh, Bm, Bl};
    // (tile, stage) position of the stream; the tile's row / column origin is derived once per tile (an integer division per stage
    // and use costs ~20 scalar instructions of an in-order wave)
    struct Pos { int tile, st, ti, m0, n0; };
    auto at_tile = [&](int ti) -> Pos {
        const int tile = tile_of(ti);
        const int tm = tile < 0 ? 0 : tile / tiles_n;
        return Pos{tile, 0, ti, tm * BM, (tile - tm * tiles_n) * BN};
    };
    auto next = [&](const Pos& p) -> Pos {
        if (p.tile < 0) return p;
        if (p.st + 1 < nst) return Pos{p.tile, p.st + 1, p.ti, p.m0, p.n0};
        return at_tile(p.ti + 1);
    };
    const Pos pz = at_tile(0);
    if (pz.tile < 0) return;
    auto valid = [&](const Pos& p) -> Pos { return p.tile >= 0 ? p : pz; };

    f32x4 ra[NA][2];
    auto a_addr = [&](const Pos& q, int i) -> const float* {
        const int m0 = q.m0, k0 = q.st * BK;
        return A + (size_t)min(m0 + a_row[i], M - 1) * lda + k0 + 8 * a_kc[i];
    };
    auto b_dma = [&](const Pos& q, int buf) {
        const int n0 = q.n0, k0 = q.st * BK;
        const size_t base = BTR ? (size_t)k0 * ldb + n0 : (size_t)n0 * ldb + k0;
        unsigned char* Bs = smem + buf * SLOT;
#pragma unroll
        for (int j = 0; j < DPW; ++j) dma16(planes[b_piece[j]] + base + b_off[j], Bs + b_dst[j]);
    };
    auto a_write = [&](int buf, const u32x4 (&sh)[NA], const u32x4 (&sm)[NA], const u32x4 (&sl)[NA]) {
        unsigned char* As = smem + buf * SLOT;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            *reinterpret_cast<u32x4*>(As + a_lds[i]) = sh[i];
            *reinterpret_cast<u32x4*>(As + A_BYTES + a_lds[i]) = sm[i];
            *reinterpret_cast<u32x4*>(As + 2 * A_BYTES + a_lds[i]) = sl[i];
        }
    };
    f32x4 acc[WM][WN];
#pragma unroll
    for (int a = 0; a < WM; ++a)
#pragma unroll
        for (int b = 0; b < WN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = t; i < NF; i += NT) bsm[i] = bias ? bias[i] : 0.f;

    s16x8 af[3][WM], bf[3][WN];
    auto read_a = [&](int buf, int a0, int a1) {
        const unsigned char* As = smem + buf * SLOT;
#pragma unroll
        for (int pc = 0; pc < 3; ++pc)
#pragma unroll
            for (int a = a0; a < a1; ++a)
                af[pc][a] = *reinterpret_cast<const s16x8*>(As + pc * A_BYTES + off64((wm * WM + a) * 16 + i16, g));
    };
    auto read_b = [&](int buf) {
        const unsigned char* Bs = smem + buf * SLOT + 3 * A_BYTES;
#pragma unroll
        for (int pc = 0; pc < 3; ++pc)
#pragma unroll
            for (int b = 0; b < WN; ++b) {
                if (BTR) bf[pc][b] = tr_frag<BN>(Bs + pc * B_BYTES, 8 * g, wn * WN + b, lane);
                else bf[pc][b] = *reinterpret_cast<const s16x8*>(Bs + pc * B_BYTES + off64((wn * WN + b) * 16 + i16, g));
            }
    };
    auto mma_rows = [&](int a0, int a1) {
        constexpr int TA[6] = {0, 2, 1, 0, 1, 0}, TB[6] = {2, 0, 1, 1, 0, 0};      // (activation piece, weight piece): lh hl mm hm mh hh
#pragma unroll
        for (int a = a0; a < a1; ++a)
#pragma unroll
            for (int term = 0; term < 6; ++term)
#pragma unroll
                for (int b = 0; b < WN; ++b) acc[a][b] = mfma_bf16(bf[TB[term]][b], af[TA[term]][a], acc[a][b]);
    };
    auto epilogue = [&](int tile) -> bool {
        const int tn = tile % tiles_n, tm = tile / tiles_n;
        const int m0 = tm * BM + wm * WM * 16 + i16, n0 = tn * BN + wn * WN * 16 + 4 * g;
        float* c0 = C + (size_t)m0 * ldc + n0;
        f32x4 bv[WN];
#pragma unroll
        for (int b = 0; b < WN; ++b) bv[b] = *reinterpret_cast<const f32x4*>(bsm + n0 + 16 * b);
        if constexpr (EPI) {
            // the shortcut values are requested here, behind this iteration's operand requests: the wait in front of the stores
            // drains those too (vmcnt retires in order) - about one memory latency per tile, a few per cent of its matrix time
            const int mbw = tm * BM + wm * WM * 16;                  // wave-uniform
            int dst[WM];
            float sc;
            tok_epi_rows<WM>(epi, mbw < M ? mbw : 0, i16, dst, sc);
            const int nn = tn * BN + wn * WN * 16 + 4 * g;
            f32x4 rv[WM][WN];
#pragma unroll
            for (int a = 0; a < WM; ++a)
#pragma unroll
                for (int b = 0; b < WN; ++b)
                    rv[a][b] = epi.res ? *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(epi.res) + (size_t)min(dst[a], M - 1) * ldc + nn + 16 * b)
                                       : f32x4{0.f, 0.f, 0.f, 0.f};
            const bool efull = tm * BM + BM <= M;                    // wave-uniform; WM * WN stores when true (the caller counts them)
#pragma unroll
            for (int a = 0; a < WM; ++a)
                if (efull || m0 + 16 * a < M) {
#pragma unroll
                    for (int b = 0; b < WN; ++b)
                        *reinterpret_cast<f32x4*>(C + (size_t)dst[a] * ldc + nn + 16 * b) = rv[a][b] + sc * (acc[a][b] + bv[b]);
                }
#pragma unroll
            for (int a = 0; a < WM; ++a)
#pragma unroll
                for (int b = 0; b < WN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
            return efull;
        }
        const bool full = tm * BM + BM <= M;
        if (full) {
#pragma unroll
            for (int a = 0; a < WM; ++a)
#pragma unroll
                for (int b = 0; b < WN; ++b) *reinterpret_cast<f32x4*>(c0 + (size_t)(16 * a) * ldc + 16 * b) = acc[a][b] + bv[b];
        } else {
#pragma unroll
            for (int a = 0; a < WM; ++a)
                if (m0 + 16 * a < M) {
#pragma unroll
                    for (int b = 0; b < WN; ++b) *reinterpret_cast<f32x4*>(c0 + (size_t)(16 * a) * ldc + 16 * b) = acc[a][b] + bv[b];
                }
        }
#pragma unroll
        for (int a = 0; a < WM; ++a)
#pragma unroll
            for (int b = 0; b < WN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        return full;
    };
    auto a_wait0 = [&]() { wait_regs<0>(ra[0][0], ra[0][1], ra[1][0], ra[1][1]); };
    auto a_waitN = [&]() { wait_regs<DPW>(ra[0][0], ra[0][1], ra[1][0], ra[1][1]); };

    // ---- prologue: stage 0 complete in slot 0, the raw activations of stage 1 in flight
    Pos p0 = pz, p1 = next(p0), p2 = next(p1);
    b_dma(p0, 0);
#pragma unroll
    for (int i = 0; i < NA; ++i) gload32(a_addr(p0, i), ra[i][0], ra[i][1]);
    a_wait0();
    {
        u32x4 sh[NA], sm[NA], sl[NA];
#pragma unroll
        for (int i = 0; i < NA; ++i) split8x3(ra[i][0], ra[i][1], sh[i], sm[i], sl[i]);
        a_write(0, sh, sm, sl);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < NA; ++i) gload32(a_addr(valid(p1), i), ra[i][0], ra[i][1]);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    for (int p = 0;; ++p) {
        const int buf = p & 1;
        const bool have1 = p1.tile >= 0;
        // ---- stream 1: rows 0, 1 (48 MFMAs); between them the DMA of stage p+1 -> the other slot, the fragment reads of rows 2, 3,
        //      the split of stage p+1's raw activations (loaded during stage p-1) and the address arithmetic of the loads below
        b_dma(valid(p1), buf ^ 1);
        read_b(buf);
        read_a(buf, 0, 2);
        a_waitN();                                                 // older than this stage's DMA (and the previous tile's stores)
        __builtin_amdgcn_sched_barrier(0);
        read_a(buf, 2, 4);
        u32x4 sh[NA], sm[NA], sl[NA];
#pragma unroll
        for (int i = 0; i < NA; ++i) split8x3(ra[i][0], ra[i][1], sh[i], sm[i], sl[i]);
        const float* pa[NA];
#pragma unroll
        for (int i = 0; i < NA; ++i) pa[i] = a_addr(valid(p2), i);
        mma_rows(0, 2);
#pragma unroll
        for (int i = 0; i < 48; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x382, 2, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NA; ++i) gload32(pa[i], ra[i][0], ra[i][1]);      // stage p + 2
        __builtin_amdgcn_sched_barrier(0);
        // ---- stream 2: rows 2, 3; between them the LDS writes of stage p+1's pieces -> the other slot
        a_write(buf ^ 1, sh, sm, sl);
        mma_rows(2, 4);
#pragma unroll
        for (int i = 0; i < 48; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x382, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        const bool stored = p0.st == nst - 1;
        bool full = true;
        if (stored) full = epilogue(p0.tile);
        if (!have1) break;
        __builtin_amdgcn_sched_barrier(0);
        // the DMA of stage p+1 has landed: it is older than the 2 NA loads and this iteration's stores
        if (!stored) wait_vm<2 * NA>();
        else if (full) wait_vm<2 * NA + WM * WN>();
        else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        p0 = p1; p1 = p2; p2 = next(p2);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <bool BTR, bool EPI>
void launch_wide(const float* A, int lda, const uint16_t* Bh, const uint16_t* Bm, const uint16_t* Bl, int ldb, const float* bias, float* C,
                 int ldc, int M, int NF, int KC, const TokEpi& epi, hipStream_t s) {
    constexpr int BM = 256, BN = 128;
    const size_t smem = 2 * 3 * (size_t)(BM * 64 + BN * 64) + (size_t)NF * sizeof(float);
    const int tiles_n = NF / BN, tiles_m = (M + BM - 1) / BM;
    const int ntiles = tiles_n * tiles_m;
    const int slots = dhz_num_cus();
    const int grid = ntiles < slots ? ntiles : slots;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&split6_wide_kernel<BTR, EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL((split6_wide_kernel<BTR, EPI>), dim3(grid), dim3(NT), smem, s, A, lda, Bh, Bm, Bl, ldb, bias, C, ldc, M, NF, KC, tiles_n,
                       ntiles, epi);
}

template <int WM, int WN, int WAVES_M, bool BTR, bool EPI>
void launch(const float* A, int lda, const uint16_t* Bh, const uint16_t* Bm, const uint16_t* Bl, int ldb, const float* bias, float* C,
            int ldc, int M, int NF, int KC, const TokEpi& epi, hipStream_t s) {
    constexpr int BM = 16 * WM * WAVES_M, BN = 16 * WN * (8 / WAVES_M);
    const size_t smem = 3 * 3 * (size_t)(BM * 64 + BN * 64) + (size_t)NF * sizeof(float);     // three ring slots per operand + the bias vector
    const int tiles_n = NF / BN, tiles_m = (M + BM - 1) / BM;
    const int ntiles = tiles_n * tiles_m;
    const int slots = dhz_num_cus();
    const int grid = ntiles < slots ? ntiles : slots;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&split6_gemm_kernel<WM, WN, WAVES_M, BTR, EPI>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL((split6_gemm_kernel<WM, WN, WAVES_M, BTR, EPI>), dim3(grid), dim3(NT), smem, s, A, lda, Bh, Bm, Bl, ldb, bias, C, ldc,
                       M, NF, KC, tiles_n, ntiles, epi);
}

template <bool BTR, bool EPI = false>
int dispatch(const char* who, const float* A, int lda, const uint16_t* Bh, const uint16_t* Bm, const uint16_t* Bl, int ldb,
             const float* bias, float* C, int ldc, int M, int NF, int KC, hipStream_t s, const TokEpi& epi = TokEpi{}) {
    DHZ_REQUIRE(A && Bh && Bm && Bl && C, "%s: null pointer", who);
    DHZ_REQUIRE(M > 0 && NF > 0 && KC > 0 && NF % (BTR ? 64 : 32) == 0 && KC % 32 == 0,
                "%s: T=%d features=%d contraction=%d (contraction a multiple of 32, features of %d)", who, M, NF, KC, BTR ? 64 : 32);
    DHZ_REQUIRE(lda % 4 == 0 && ldb % 8 == 0 && ldc % 4 == 0 && ldc >= NF && lda >= KC, "%s: bad leading dimensions", who);
    DHZ_REQUIRE((((uintptr_t)A | (uintptr_t)Bh | (uintptr_t)Bm | (uintptr_t)Bl | (uintptr_t)C | (uintptr_t)bias) & 15) == 0,
                "%s: operands must be 16-byte aligned", who);
    DHZ_REQUIRE((long)ldb * (BTR ? KC : NF) < (1L << 31), "%s: weight matrix too large", who);
    DHZ_REQUIRE(NF <= 2048, "%s: %d output features (at most 2048: the bias vector is staged in LDS)", who, NF);
    // tile: the widest feature block that divides NF (the pre-split weight side costs no vector work: wide tiles re-split the
    // activations fewer times); 128 tokens (three ring slots of both operands fit the LDS)
    const int cus = dhz_num_cus();
    int bn = NF % 128 == 0 ? 128 : NF % 64 == 0 ? 64 : 32;
    if (bn == 128 && (long)((M + 127) / 128) * (NF / 128) < cus && (long)((M + 127) / 128) * (NF / 64) >= cus / 2) bn = 64;
    if (EPI) {
        const char* bad = tok_epi_check(epi, M);
        DHZ_REQUIRE(!bad, "%s: %s", who, bad);
        DHZ_REQUIRE(((uintptr_t)epi.res & 15) == 0, "%s: the shortcut must be 16-byte aligned", who);
    }
#define GO(WM_, WN_, WV_) launch<WM_, WN_, WV_, BTR, EPI>(A, lda, Bh, Bm, Bl, ldb, bias, C, ldc, M, NF, KC, epi, s)
    static const int force = getenv("DHZ_S6_TILE") ? atoi(getenv("DHZ_S6_TILE")) : 0;          // diagnostics: 1 = never wide, 2 = always wide
    const bool wide = bn == 128 && force != 1 && (force == 2 || (long)((M + 255) / 256) * (NF / 128) >= cus);
    if (wide) launch_wide<BTR, EPI>(A, lda, Bh, Bm, Bl, ldb, bias, C, ldc, M, NF, KC, epi, s);        // 256 x 128
    else if (bn == 128) GO(4, 2, 2);                              // 128 x 128
    else if (bn == 64) GO(2, 2, 4);                               // 128 x 64
    else if constexpr (!BTR) GO(1, 2, 8);                         // 128 x 32
#undef GO
    DHZ_CHECK_LAUNCH(who);
    return DHZ_OK;
}

// ---- the three bf16 planes of an fp32 buffer (the split the GEMMs would otherwise do per use), n a multiple of 8
__global__ __launch_bounds__(256) void split3_planes_kernel(const float* __restrict__ src, long n8, uint16_t* __restrict__ hi,
                                                            uint16_t* __restrict__ mid, uint16_t* __restrict__ lo) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(src + 8 * i), b = *reinterpret_cast<const f32x4*>(src + 8 * i + 4);
        u32x4 h, m, l;
        split8x3(a, b, h, m, l);
        *reinterpret_cast<u32x4*>(hi + 8 * i) = h;
        *reinterpret_cast<u32x4*>(mid + 8 * i) = m;
        *reinterpret_cast<u32x4*>(lo + 8 * i) = l;
    }
}

// ---- the planes of the TRANSPOSES of a set of matrices that live in one fp32 buffer (the flat parameter buffer): matrix m occupies
// src[off, off + R C) as [R][C]; its transpose goes to the SAME offsets of the three planes as [C][R].  Backward-data then runs the
// forward kernel on them (dx = dy . W = dy . (W^T)^T): no transposed fragment reads - in front of those the compiler waits for every
// LDS-DMA in flight - and measured 12 % faster over the step's shapes, bit-identical results.  desc[m] = {off, R, C, first tile};
// 32 x 32 tiles through LDS; R, C multiples of 32.
__global__ __launch_bounds__(256) void split3_planes_t_kernel(const float* __restrict__ src, uint16_t* __restrict__ hi,
                                                              uint16_t* __restrict__ mid, uint16_t* __restrict__ lo,
                                                              const int* __restrict__ desc, int nmat, int ntiles) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 32 x 8
    for (int tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        int m = 0;
        while (m + 1 < nmat && desc[4 * (m + 1) + 3] <= tl) ++m;   // a few dozen matrices
        const int off = desc[4 * m], R = desc[4 * m + 1], Cc = desc[4 * m + 2], t0 = desc[4 * m + 3];
        const int tpr = Cc / 32, ti = (tl - t0) / tpr, tj = (tl - t0) % tpr;
        const float* s0 = src + off + (size_t)(32 * ti) * Cc + 32 * tj;
#pragma unroll
        for (int r = ty; r < 32; r += 8) tile[r][tx] = s0[(size_t)r * Cc + tx];
        __syncthreads();
        const size_t d0 = (size_t)off + (size_t)(32 * tj) * R + 32 * ti;
#pragma unroll
        for (int c = ty; c < 32; c += 8) {
            const float x = tile[tx][c];                           // element (row 32 ti + tx, column 32 tj + c) -> its transposed position
            const float h = __uint_as_float(__float_as_uint(x) & 0xffff0000u);
            const float r1 = x - h;
            const float mdl = __uint_as_float(__float_as_uint(r1) & 0xffff0000u);
            const float r2 = r1 - mdl;
            const size_t o = d0 + (size_t)c * R + tx;
            hi[o] = (uint16_t)(__float_as_uint(x) >> 16);
            mid[o] = (uint16_t)(__float_as_uint(r1) >> 16);
            lo[o] = (uint16_t)(__float_as_uint(r2) >> 16);
        }
        __syncthreads();
    }
}

}  // namespace


extern "C" int dhz_split3_planes_t(const float* src, void* hi, void* mid, void* lo, const int* desc, int nmat, int ntiles, void* stream) {
    const char* who = "dhz_split3_planes_t";
    DHZ_REQUIRE(src && hi && mid && lo && desc && nmat > 0 && ntiles > 0, "%s: null pointer or empty table", who);
    const int cap = 8 * dhz_num_cus();
    hipLaunchKernelGGL(split3_planes_t_kernel, dim3(ntiles < cap ? ntiles : cap), dim3(256), 0, (hipStream_t)stream, src, (uint16_t*)hi,
                       (uint16_t*)mid, (uint16_t*)lo, desc, nmat, ntiles);
    DHZ_CHECK_LAUNCH(who);
    return DHZ_OK;
}

extern "C" int dhz_split3_planes(const float* src, int64_t n, void* hi, void* mid, void* lo, void* stream) {
    const char* who = "dhz_split3_planes";
    DHZ_REQUIRE(src && hi && mid && lo && n > 0 && n % 8 == 0, "%s: null pointer or n=%lld not a positive multiple of 8", who, (long long)n);
    DHZ_REQUIRE((((uintptr_t)src | (uintptr_t)hi | (uintptr_t)mid | (uintptr_t)lo) & 15) == 0, "%s: 16-byte alignment", who);
    const long n8 = n / 8;
    long blocks = (n8 + 255) / 256;
    const long cap = 8L * dhz_num_cus();
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(split3_planes_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, n8, (uint16_t*)hi,
                       (uint16_t*)mid, (uint16_t*)lo);
    DHZ_CHECK_LAUNCH(who);
    return DHZ_OK;
}

extern "C" int dhz_linear_fwd_split6(const float* x, int ldx, const void* w_hi, const void* w_mid, const void* w_lo, const float* bias,
                                     float* y, int ldy, int T, int N, int K, void* stream) {
    return dispatch<false>("dhz_linear_fwd_split6", x, ldx, (const uint16_t*)w_hi, (const uint16_t*)w_mid, (const uint16_t*)w_lo, K, bias,
                           y, ldy, T, N, K, (hipStream_t)stream);
}

extern "C" int dhz_linear_dgrad_split6(const float* dy, int ldy, const void* w_hi, const void* w_mid, const void* w_lo, float* dx,
                                       int ldx, int T, int N, int K, void* stream) {
    // dx[T,K] = dy[T,N] w[N,K]: the contraction runs over the N rows of w (transposed reads), the output features are its K columns
    return dispatch<true>("dhz_linear_dgrad_split6", dy, ldy, (const uint16_t*)w_hi, (const uint16_t*)w_mid, (const uint16_t*)w_lo, K,
                          nullptr, dx, ldx, T, K, N, (hipStream_t)stream);
}

extern "C" int dhz_linear_fwd_split6_res(const float* x, int ldx, const void* w_hi, const void* w_mid, const void* w_lo, const float* bias,
                                         const float* res, const float* scale, float* out, int ldo, int T, int N, int K, int tokens_per_image,
                                         int Hres, int Wres, int shift, int windowed, void* stream) {
    const TokEpi epi{res, scale, tokens_per_image, Hres, Wres, shift, windowed};
    return dispatch<false, true>("dhz_linear_fwd_split6_res", x, ldx, (const uint16_t*)w_hi, (const uint16_t*)w_mid, (const uint16_t*)w_lo, K,
                                 bias, out, ldo, T, N, K, (hipStream_t)stream, epi);
}
