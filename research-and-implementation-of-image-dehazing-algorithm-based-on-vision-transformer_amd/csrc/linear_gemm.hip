// Forward and backward-data GEMMs of every token-major nn.Linear on the path, on the fp32 matrix pipe:
//     forward        y[T,N]  = x[T,K]  . W[N,K]^T + b        (WT = true : W rows are output features, K contiguous)
//     backward-data  dx[T,K] = dy[T,N] . W[N,K]              (WT = false: W rows are the CONTRACTION index)
// Both are "tall" GEMMs: T = B*H*W tokens (2k .. 524k) against 32..2048 features, so the block tile is 32*WM tokens x
// 32*WN features with four waves as 2 x 2, the contraction streamed in 32-deep stages through two LDS stage buffers.
//
// Round 3 structure.  In-kernel s_memtime stamps of the round-2 kernel (tools/micro/stamp_gemm.py) showed that a wave which is
// NOT issuing MFMAs - global-load address arithmetic, the vmcnt waits + ds_write_b128 burst of the register staging, the
// barrier - crawls at the pace of its SIMD partner's fp32 MFMA stream (600-3000 cycles per stage for ~40 instructions), and
// the four waves of a workgroup meet at the barrier with whatever skew their partners gave them: matrix pipe 0.63-0.67 busy.
//   * operands go global -> LDS by LDS-DMA (global_load_lds_dwordx4: no VGPR staging, no ds_write, no wait for load DATA in
//     the instruction stream); the images are lane-linear per wave instruction (1 KiB = 8 rows x 128 B) and the bank swizzle
//     is carried by the per-lane SOURCE address (MI355X_MICROARCH.md, LDS-DMA);
//   * the DMA of stage s+1 is issued at the start of stage s (right behind the barrier that freed its buffer) and, at a tile
//     boundary, BEFORE the epilogue stores of the finished tile: vmcnt retires in order, so the end-of-stage wait is the
//     counted s_waitcnt vmcnt(#stores issued since) and the stores drain behind the next tile's matrix work instead of in
//     front of it;
//   * one raw s_barrier per stage (a __syncthreads() would drain vmcnt); fragment reads of the second half of a stage are
//     issued inside the MFMA stream of the first half (sched_barrier-pinned): a wave's own non-matrix instructions ride in
//     its own MFMA shadow;
//   * the bias is the initial value of the accumulators, read from an LDS copy of the bias vector: no global load result is
//     waited for anywhere in the tile loop.
// LDS images (floats):
//   * token operand, and W when WT: [row][32 k]; the 16-byte k-quads of a row XOR-swizzled by (row >> 1) & 7, so that a lane's
//     four consecutive k are ONE conflict-free ds_read_b128; MFMA j of a 16-deep step contracts over k = 16 s + 4 g + j
//     (g = lane >> 4) - any order of k is valid as long as both operands use the same one.
//   * W when !WT: [32 k][BN] as it lies in memory (rows of W are the contraction index).  A lane owns WN CONSECUTIVE output
//     features (column WN * i16 + b of the wave's 16 WN): its B values of one k are one ds_read of WN dwords, and its
//     accumulator columns are one WN-dword store in the epilogue.  Conflict-free as is for WN = 4 (b128 lane groups mix two
//     rows whose 16-byte slots interleave); for WN = 2 / 1 rows with k & 4 set have their two halves swapped (source side).
// Workgroups are persistent (two per CU) and treat their (tile, stage) pairs as one stream.
#include <stdio.h>
#include <stdlib.h>
#include "common.h"

namespace {

constexpr int BK = 32;
#ifndef DHZ_GEMM_ABL
#define DHZ_GEMM_ABL 0           // timing diagnostics (tools/gemm_phases.sh): 1 = no epilogue stores, 2 = no MFMAs, 4 = no global operand loads
#endif

#ifndef DHZ_GEMM_PRIO
#define DHZ_GEMM_PRIO 0          // experiment switch: 1 = s_setprio 1 in the first half of a stage (no DMA), 0 in the second; 2 = the reverse
#endif
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
// 16 bytes per lane, global -> LDS: the LDS address is wave-uniform (M0) + 16 * lane, the global address is per lane
__device__ __forceinline__ void dma16(const float* g, float* l) {
    __builtin_amdgcn_global_load_lds((glb_void*)g, (lds_void*)l, 16, 0, 0);
}

#ifdef DHZ_GEMM_STAMP      // timing diagnostics only (tools/micro/stamp_gemm.py builds its own copy): s_memtime at the phase boundaries,
                           // kept in LDS (a global store per stamp would sit in vmcnt and be waited for) and dumped at the end
__device__ long long* g_stamp = nullptr;
constexpr int NSTAMP = 24;
#define STAMP(slot)                                                                                          \
    do {                                                                                                     \
        if (lane == 0 && nstamp < NSTAMP)                                                                    \
            stamp_lds[(w * NSTAMP + nstamp) * 8 + (slot)] = (unsigned)__builtin_amdgcn_s_memtime();          \
    } while (0)
#else
#define STAMP(slot)
#endif

#ifdef DHZ_GEMM_CLOCK      // timing diagnostics only: in-kernel shader clock = d(s_memtime) / d(s_memrealtime) x 100 MHz, per workgroup
__device__ long long* g_clock = nullptr;
#endif

template <int WN> struct VecN;
template <> struct VecN<1> { typedef float type; };
template <> struct VecN<2> { typedef float type __attribute__((ext_vector_type(2))); };
template <> struct VecN<4> { typedef float type __attribute__((ext_vector_type(4))); };

template <int WM, int WN, bool WT, bool RAG>
__global__ __launch_bounds__(256) void linear_gemm_kernel(const float* __restrict__ A, int lda,
                                                          const float* __restrict__ W, int ldw,
                                                          const float* __restrict__ bias, float* __restrict__ Y, int ldy,
                                                          int M, int N, int K, int tiles_n, int ntiles) {
    constexpr int BM = 32 * WM, BN = 32 * WN;
    constexpr int A_FLOATS = BM * BK, B_FLOATS = BN * BK;
    constexpr int STAGE = A_FLOATS + B_FLOATS;
    constexpr int abl = DHZ_GEMM_ABL;
    static_assert(WN == 1 || WN == 2 || WN == 4, "1, 2 or 4 column tiles per wave");
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);          // wave-uniform (scalar LDS addresses for the DMA's M0)
    const int i16 = lane & 15, g = lane >> 4;
    const int wm = w >> 1, wn = w & 1;
    const int nst = K / BK;
#ifdef DHZ_GEMM_CLOCK
    const long long ck0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
    long long ck_wait = 0, ck_bar = 0, ck_ep = 0, ck_t = 0;
#define CK_BEGIN() ck_t = __builtin_amdgcn_s_memtime()
#define CK_END(acc) do { const long long n_ = __builtin_amdgcn_s_memtime(); acc += n_ - ck_t; ck_t = n_; } while (0)
#else
#define CK_BEGIN()
#define CK_END(acc)
#endif

    // Persistent workgroups: a workgroup walks the tiles bid, bid + grid, ...  XCD-aware numbering: workgroups b and b + 8
    // share an XCD (round-robin dispatch), and consecutive LOGICAL tiles share their token rows across the tn sweep: map the
    // tiles so that an XCD owns a contiguous range of them (the re-read of A then comes from that XCD's L2).
    const int grid = gridDim.x;
    auto tile_of = [&](int i) -> int {                            // i-th tile of this workgroup, or -1
        const int lin = blockIdx.x + i * grid;
        if (lin >= ntiles) return -1;
        if ((grid & 7) == 0 && (ntiles & 7) == 0) return (lin & 7) * (ntiles >> 3) + (lin >> 3);
        return lin;
    };

    // ---- per-lane source offsets of this wave's DMA instructions (floats, relative to the stage's first element).
    // A / W(WT): instruction ia covers image rows 8 ia .. 8 ia + 7; lane -> row 8 ia + (lane >> 3), and its 16-byte slot lane & 7
    // receives k-quad slot ^ ((row >> 1) & 7).
    const int lrow = lane >> 3, lslot = lane & 7;
    int offA[WM], offB[WN];
#pragma unroll
    for (int i = 0; i < WM; ++i) {
        const int row = 8 * (w * WM + i) + lrow;
        offA[i] = row * lda + 4 * (lslot ^ ((row >> 1) & 7));
    }
#pragma unroll
    for (int i = 0; i < WN; ++i) {
        if (WT) {
            // image row R (of the 16 WN rows of wave column block R / (16 WN)) holds output feature WN * (R & 15) + (R >> 4) of
            // that block: fragment reads stay the conflict-free "row 16 b + i16", and a lane's WN accumulator columns are WN
            // CONSECUTIVE features (one WN-dword store in the epilogue instead of WN dword stores)
            const int row = 8 * (w * WN + i) + lrow;
            const int blk = row / (16 * WN), R = row % (16 * WN);
            const int feat = blk * 16 * WN + WN * (R & 15) + (R >> 4);
            offB[i] = feat * ldw + 4 * (lslot ^ ((row >> 1) & 7));
        } else {
            const int e = ((w * WN + i) * 64 + lane) * 4;         // float index in the [32][BN] image
            const int kk = e / BN, c = e % BN;
            const int cs = WN == 4 ? c : (c ^ (((kk >> 2) & 1) * (BN / 2)));
            offB[i] = kk * ldw + cs;
        }
    }

    float* bsm = smem + 2 * STAGE;
#ifdef DHZ_GEMM_STAMP
    unsigned* stamp_lds = reinterpret_cast<unsigned*>(bsm + N);
    for (int i = t; i < 4 * NSTAMP * 8; i += 256) stamp_lds[i] = 0;
#endif
    for (int i = t; i < N; i += 256) bsm[i] = bias ? bias[i] : 0.f;      // visible after the first barrier below

    // One operand stage = WM + WN DMA instructions per wave (first the token operand's, then W's), issued one by one inside
    // the MFMA stream.  StageSrc: the wave-uniform part of their source addresses, computed once per stage.
    // RAG (T not a multiple of the tile height; a template flag so that the common case carries no per-instruction test):
    // rows past the end re-read the last valid row (never stored).
    struct StageSrc { const float* a; const float* b; int m0; };
    auto stage_src = [&](int tile, int st) -> StageSrc {
        const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN, k0 = st * BK;
        return StageSrc{A + (size_t)m0 * lda + k0, WT ? W + (size_t)n0 * ldw + k0 : W + (size_t)k0 * ldw + n0, m0};
    };
    auto dma_one = [&](int idx, const StageSrc& src, int buf) {
        if (abl & 4) return;
        float* As = smem + buf * STAGE;
        if (idx < WM) {
            int off = offA[idx < WM ? idx : 0];
            if constexpr (RAG) {
                const int row = 8 * (w * WM + idx) + lrow;
                off += (min(src.m0 + row, M - 1) - src.m0 - row) * lda;
            }
            dma16(src.a + off, As + (w * WM + idx) * 256);
        } else {
            const int i = idx - WM;
            dma16(src.b + offB[i < WN ? i : 0], As + A_FLOATS + (w * WN + i) * 256);
        }
    };

    f32x4 acc[WM][WN];
    auto acc_init = [&](int tile) {
        const float* bp = bsm + (tile % tiles_n) * BN + wn * WN * 16;
#pragma unroll
        for (int b = 0; b < WN; ++b) {
            const float bv = bp[WN * i16 + b];
#pragma unroll
            for (int a = 0; a < WM; ++a) acc[a][b] = f32x4{bv, bv, bv, bv};
        }
    };

    // fragments of half-stage s (16 k) from stage buffer `buf`
    const int sw = (i16 >> 1) & 7;
    auto frag_a = [&](int buf, int s, f32x4 (&af)[WM]) {
        const float* As = smem + buf * STAGE + (wm * WM * 16 + i16) * BK + 4 * ((4 * s + g) ^ sw);
#pragma unroll
        for (int a = 0; a < WM; ++a) af[a] = *reinterpret_cast<const f32x4*>(As + a * 16 * BK);
    };
    // bf[b][j]: B value of column tile b for MFMA j
    auto frag_b = [&](int buf, int s, f32x4 (&bf)[WN]) {
        const float* Bs = smem + buf * STAGE + A_FLOATS;
        if constexpr (WT) {
#pragma unroll
            for (int b = 0; b < WN; ++b)
                bf[b] = *reinterpret_cast<const f32x4*>(&Bs[((wn * WN + b) * 16 + i16) * BK + 4 * ((4 * s + g) ^ sw)]);
        } else {
            typedef typename VecN<WN>::type vec;
            // row kk = 16 s + 4 g + j; rows with kk & 4 (g odd) hold their halves swapped for WN < 4
            const int c = wn * WN * 16 + WN * i16;
            const int cs = WN == 4 ? c : (c ^ ((g & 1) * (BN / 2)));
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const vec v = *reinterpret_cast<const vec*>(&Bs[(16 * s + 4 * g + j) * BN + cs]);
#pragma unroll
                for (int b = 0; b < WN; ++b) {
                    if constexpr (WN == 1) bf[b][j] = v; else bf[b][j] = v[b];
                }
            }
        }
    };
    auto mma_j = [&](const f32x4 (&af)[WM], const f32x4 (&bf)[WN], int j) {
#pragma unroll
        for (int a = 0; a < WM; ++a)
#pragma unroll
            for (int b = 0; b < WN; ++b) {
                if (abl & 2) acc[a][b][j] += af[a][j] + bf[b][j];
                else acc[a][b] = mfma16(af[a][j], bf[b][j], acc[a][b]);
            }
    };

    // ---- epilogue of a finished tile, straight from the accumulators.  acc[a][b][j] = C[16 a + 4 g + j][column(b, i16)].
    //      Full tiles (every tile unless T is ragged) store without per-row guards: behind a divergent guard hipcc waits for
    //      vmcnt(0) - i.e. for the acknowledgement of ALL earlier stores - before every guarded group.
    //      Returns the number of store instructions issued (for the counted wait), or -1 for "unknown: wait for everything".
    auto epilogue = [&](int tile) -> int {
        const int tn = tile % tiles_n, tm = tile / tiles_n;
        const bool full = !RAG || tm * BM + BM <= M;                  // wave-uniform
        const int m0 = tm * BM + wm * WM * 16 + 4 * g;
        {
            typedef typename VecN<WN>::type vec;
            float* y0 = Y + (size_t)m0 * ldy + tn * BN + wn * WN * 16 + WN * i16;
            auto pack = [&](int a, int j) -> vec {
                if constexpr (WN == 1) return acc[a][0][j];
                else if constexpr (WN == 2) return vec{acc[a][0][j], acc[a][1][j]};
                else return vec{acc[a][0][j], acc[a][1][j], acc[a][2][j], acc[a][3][j]};
            };
            if (full) {
#pragma unroll
                for (int a = 0; a < WM; ++a)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (!(abl & 1) || acc[a][0][0] == 12345.678f)
                            *reinterpret_cast<vec*>(y0 + (size_t)(16 * a + j) * ldy) = pack(a, j);
                return WM * 4;
            }
#pragma unroll
            for (int a = 0; a < WM; ++a)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (m0 + 16 * a + j < M) *reinterpret_cast<vec*>(y0 + (size_t)(16 * a + j) * ldy) = pack(a, j);
            return -1;
        }
    };

    // ---- the stream of (tile, stage) pairs of this workgroup
    struct Pos { int tile, st, ti; };
    auto next = [&](const Pos& p) -> Pos {
        if (p.tile < 0) return p;
        if (p.st + 1 < nst) return Pos{p.tile, p.st + 1, p.ti};
        return Pos{tile_of(p.ti + 1), 0, p.ti + 1};
    };
    Pos p0{tile_of(0), 0, 0};
    if (p0.tile < 0) return;
    Pos p1 = next(p0);
    {
        const StageSrc s0 = stage_src(p0.tile, p0.st);
#pragma unroll
        for (int i = 0; i < WM + WN; ++i) dma_one(i, s0, 0);
        if (p1.tile >= 0) {
            const StageSrc s1 = stage_src(p1.tile, p1.st);
#pragma unroll
            for (int i = 0; i < WM + WN; ++i) dma_one(i, s1, 1);
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    acc_init(p0.tile);

    // The loop is rotated by half a stage: iteration p runs the SECOND half (k 16..31) of stage p and the FIRST half of stage
    // p + 1, with the barrier between two iterations.  At that barrier every wave holds both fragment sets of stage p in
    // registers, so buffer p & 1 is free: the DMA of stage p + 2 goes into it, one instruction per 8-MFMA group of the second
    // half; it is waited for one iteration later, i.e. it has a whole stage of matrix time to land.  A finished tile's stores
    // are issued between the two halves, BEHIND that DMA (vmcnt retires in order: the wait at the next barrier is the counted
    // vmcnt(#stores) and leaves them in flight); they have to be complete one barrier later.
    f32x4 af0[WM], bf0[WN], af1[WM], bf1[WN];
    frag_a(0, 0, af0);
    frag_b(0, 0, bf0);
    frag_a(0, 1, af1);
    frag_b(0, 1, bf1);
#pragma unroll
    for (int j = 0; j < 4; ++j) mma_j(af0, bf0, j);
    int nstore = 0;
    int nstamp = 0;
    (void)nstamp;
    for (int p = 0;; ++p) {
        const int buf = p & 1;
        const Pos p2 = next(p1);
        STAMP(0);
        if (p1.tile >= 0) {
            CK_BEGIN();
            // my DMA of stage p + 1 has landed and my reads of buffer `buf` are complete; then: everybody's
            if (nstore == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            else if (nstore == 16) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
            else if (nstore == 8) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
            else if (nstore >= 63) asm volatile("s_waitcnt vmcnt(63) lgkmcnt(0)" ::: "memory");
            else if (nstore >= 32) asm volatile("s_waitcnt vmcnt(32) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            STAMP(1);
            CK_END(ck_wait);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            CK_END(ck_bar);
        }
        STAMP(2);
        if (DHZ_GEMM_PRIO == 1) __builtin_amdgcn_s_setprio(0);
        if (DHZ_GEMM_PRIO == 2) __builtin_amdgcn_s_setprio(1);
        const bool have1 = p1.tile >= 0;
        // past the end of the stream the DMA re-loads this stage (into the buffer nobody reads any more) and the fragment reads
        // fetch stale bytes: the loop body carries no tests around them
        const StageSrc src2 = p2.tile >= 0 ? stage_src(p2.tile, p2.st) : stage_src(p0.tile, p0.st);
        // ---- second half of stage p (fragment set 1); inside it: DMA of stage p + 2, first-half fragments of stage p + 1
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int a = hf * (WM / 2); a < (hf + 1) * (WM / 2); ++a)
#pragma unroll
                    for (int b = 0; b < WN; ++b) {
                        if (abl & 2) acc[a][b][j] += af1[a][j] + bf1[b][j];
                        else acc[a][b] = mfma16(af1[a][j], bf1[b][j], acc[a][b]);
                    }
                __builtin_amdgcn_sched_barrier(0);
                const int slot = 2 * j + hf;
                if (slot < WM + WN) dma_one(slot, src2, buf);
                if (slot == 0) frag_a(buf ^ 1, 0, af0);
                if (slot == 1) frag_b(buf ^ 1, 0, bf0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        STAMP(3);
        nstore = 0;
        if (p0.st == nst - 1) {
            CK_BEGIN();
            nstore = epilogue(p0.tile);
            if (have1) acc_init(p1.tile);
            CK_END(ck_ep);
        }
        STAMP(4);
        if (!have1) break;
        if (DHZ_GEMM_PRIO == 1) __builtin_amdgcn_s_setprio(1);
        if (DHZ_GEMM_PRIO == 2) __builtin_amdgcn_s_setprio(0);
        // ---- first half of stage p + 1 (fragment set 0); inside it: its second-half fragments
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            __builtin_amdgcn_sched_barrier(0);
            mma_j(af0, bf0, j);
            __builtin_amdgcn_sched_barrier(0);
            if (j == 0) frag_a(buf ^ 1, 1, af1);
            if (j == 1) frag_b(buf ^ 1, 1, bf1);
        }
        __builtin_amdgcn_sched_barrier(0);
        STAMP(5);
        ++nstamp;
        p0 = p1;
        p1 = p2;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // no LDS-DMA may be in flight when the workgroup's LDS is released
#ifdef DHZ_GEMM_CLOCK
    if (g_clock && t == 0) {
        g_clock[8 * blockIdx.x] = __builtin_amdgcn_s_memtime() - ck0;
        g_clock[8 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - rt0;
        g_clock[8 * blockIdx.x + 2] = ck_wait;
        g_clock[8 * blockIdx.x + 3] = ck_bar;
        g_clock[8 * blockIdx.x + 4] = ck_ep;
    }
#endif
#ifdef DHZ_GEMM_STAMP
    __syncthreads();
    if (g_stamp && blockIdx.x == DHZ_GEMM_STAMP)
        for (int i = t; i < 4 * NSTAMP * 8; i += 256) g_stamp[i] = stamp_lds[i];
#endif
}

template <int WM, int WN, bool WT, bool RAG>
void launch(const float* A, int lda, const float* W, int ldw, const float* bias, float* Y, int ldy, int M, int N, int K,
            hipStream_t s) {
    constexpr int BM = 32 * WM, BN = 32 * WN;
    constexpr size_t stage = (size_t)(BM + BN) * BK * sizeof(float);
#ifdef DHZ_GEMM_STAMP
    const size_t smem = 2 * stage + (size_t)N * sizeof(float) + 4 * NSTAMP * 8 * 4;
#else
    const size_t smem = 2 * stage + (size_t)N * sizeof(float);       // two operand stages + the bias vector
#endif
    const int tiles_n = N / BN, tiles_m = (M + BM - 1) / BM;
    const int ntiles = tiles_n * tiles_m;
    // two workgroups per CU when the LDS allows (it does for every tile shape: <= 64 KiB + bias per workgroup)
#ifdef DHZ_DIAG
    const int slots = getenv("DHZ_GEMM_SLOTS") ? atoi(getenv("DHZ_GEMM_SLOTS")) : 2 * dhz_num_cus();
#else
    const int slots = 2 * dhz_num_cus();
#endif
    const int grid = ntiles < slots ? ntiles : slots;
    if (smem > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_gemm_kernel<WM, WN, WT, RAG>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL((linear_gemm_kernel<WM, WN, WT, RAG>), dim3(grid), dim3(256), smem, s, A, lda, W, ldw, bias, Y, ldy, M, N,
                       K, tiles_n, ntiles);
}

// Narrow shapes (output features or contraction a multiple of 16 but not of 32: the embed_dim = 16 model, utils/model_utils.py:96-98
// "Uformer16"; its C = 16 stage has Linears 16 -> 48 / 16 / 64 and 64 -> 16).  One wave = 16 token rows x all output features, operands
// straight from global memory / L1 (these tensors are 16 .. 64 floats wide: nothing to tile); lane (i16, g) of MFMA k-step s takes
// contraction index 16 kc + 4 g + s for BOTH operands, so the A fragment of a 16-deep chunk is one float4 load.
template <bool WT>
__global__ __launch_bounds__(256) void narrow_gemm_kernel(const float* __restrict__ A, int lda, const float* __restrict__ W, int ldw,
                                                          const float* __restrict__ bias, float* __restrict__ Y, int ldy, int M, int N,
                                                          int K) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, i16 = lane & 15, g = lane >> 4;
    const long row0 = ((long)blockIdx.x * 4 + w) * 16;
    if (row0 >= M) return;
    const long arow = row0 + i16 < M ? row0 + i16 : M - 1;                  // ragged last tile: clamp the read, guard the store
    for (int nt = 0; nt < N / 16; ++nt) {
        const float b0 = bias ? bias[16 * nt + i16] : 0.f;
        f32x4 acc = {b0, b0, b0, b0};
        for (int kc = 0; kc < K / 16; ++kc) {
            const f32x4 a4 = *reinterpret_cast<const f32x4*>(A + arow * lda + 16 * kc + 4 * g);
            f32x4 b4;
            if (WT) b4 = *reinterpret_cast<const f32x4*>(W + (long)(16 * nt + i16) * ldw + 16 * kc + 4 * g);
            else {
#pragma unroll
                for (int s = 0; s < 4; ++s) b4[s] = W[(long)(16 * kc + 4 * g + s) * ldw + 16 * nt + i16];
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = mfma16(a4[s], b4[s], acc);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (row0 + 4 * g + j < M) Y[(row0 + 4 * g + j) * ldy + 16 * nt + i16] = acc[j];
    }
}

template <bool WT>
int dispatch(const char* who, const float* A, int lda, const float* W, int ldw, const float* bias, float* Y, int ldy, int M,
             int N, int K, hipStream_t s) {
    DHZ_REQUIRE(A && W && Y, "%s: null pointer", who);
    if (M > 0 && N > 0 && K > 0 && (N % 32 || K % 32) && N % 16 == 0 && K % 16 == 0) {        // narrow shapes (embed_dim 16)
        DHZ_REQUIRE(lda % 4 == 0 && ldw % 4 == 0 && lda >= K && ldy >= N && (((uintptr_t)A | (uintptr_t)W) & 15) == 0,
                    "%s: bad leading dimensions / alignment (narrow form)", who);
        hipLaunchKernelGGL((narrow_gemm_kernel<WT>), dim3((M + 63) / 64), dim3(256), 0, s, A, lda, W, ldw, bias, Y, ldy, M, N, K);
        DHZ_CHECK_LAUNCH(who);
        return DHZ_OK;
    }
    DHZ_REQUIRE(M > 0 && N > 0 && K > 0 && N % 32 == 0 && K % 32 == 0, "%s: T=%d N=%d K=%d (N, K must be multiples of 16)", who,
                M, N, K);
    DHZ_REQUIRE(N <= 16384, "%s: N=%d (at most 16384 output features: the bias vector is staged in LDS)", who, N);
    DHZ_REQUIRE(lda % 4 == 0 && ldy % 4 == 0 && ldw % 4 == 0 && lda >= K && ldy >= N, "%s: bad leading dimensions", who);
    DHZ_REQUIRE((((uintptr_t)A | (uintptr_t)Y | (uintptr_t)W) & 15) == 0 && ((uintptr_t)bias & 3) == 0,
                "%s: activations and weights must be 16-byte aligned", who);
    DHZ_REQUIRE((long)128 * (lda > ldw ? lda : ldw) < (1L << 30), "%s: row stride too large", who);
    // largest tile that still gives every CU two blocks: 128 x 128 down to 64 x 32 (tile width must divide N)
    int wm = 2, wn = 1;
    {
        static const int cand[6][2] = {{4, 4}, {4, 2}, {2, 4}, {2, 2}, {4, 1}, {2, 1}};
        long best_blocks = -1;
        const long slots = 2 * dhz_num_cus();
        for (int i = 0; i < 6; ++i) {
            const int a = cand[i][0], b = cand[i][1];
            if (N % (32 * b)) continue;
            const long blocks = (long)((M + 32 * a - 1) / (32 * a)) * (N / (32 * b));
            if (blocks >= slots) { wm = a; wn = b; break; }    // two resident workgroups per CU
            if (blocks > best_blocks) { best_blocks = blocks; wm = a; wn = b; }
        }
    }
#ifdef DHZ_DIAG
    if (const char* e = getenv("DHZ_GEMM_TILE")) {               // "wm,wn" (ignored where wn does not divide N / 32)
        int a = 0, b = 0;
        if (sscanf(e, "%d,%d", &a, &b) == 2 && N % (32 * b) == 0) { wm = a; wn = b; }
    }
#endif
#define CASE(a, b)                                                                       \
    if (wm == a && wn == b) {                                                            \
        if (M % (32 * a)) launch<a, b, WT, true>(A, lda, W, ldw, bias, Y, ldy, M, N, K, s);  \
        else launch<a, b, WT, false>(A, lda, W, ldw, bias, Y, ldy, M, N, K, s);              \
    }
    CASE(4, 1) CASE(4, 2) CASE(4, 4) CASE(2, 1) CASE(2, 2) CASE(2, 4)
#undef CASE
    DHZ_CHECK_LAUNCH(who);
    return DHZ_OK;
}

}  // namespace

#ifdef DHZ_GEMM_CLOCK
extern "C" int dhz_debug_clock(void* p) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_clock), &p, sizeof(p));
}
#endif
#ifdef DHZ_GEMM_STAMP
extern "C" int dhz_debug_stamp(void* p) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_stamp), &p, sizeof(p));
}
#endif

extern "C" int dhz_linear_fwd(const float* x, int ldx, const float* w, const float* bias, float* y, int ldy, int T, int N, int K,
                              void* stream) {
    return dispatch<true>("dhz_linear_fwd", x, ldx, w, K, bias, y, ldy, T, N, K, (hipStream_t)stream);
}

extern "C" int dhz_linear_dgrad(const float* dy, int ldy, const float* w, float* dx, int ldx, int T, int N, int K,
                                void* stream) {
    // dx[T,K] = dy[T,N] w[N,K]: the contraction runs over the N rows of w, the output features are its K columns
    return dispatch<false>("dhz_linear_dgrad", dy, ldy, w, K, nullptr, dx, ldx, T, K, N, (hipStream_t)stream);
}
