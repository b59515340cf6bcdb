// Token-Linear GEMM for bf16 activations and bf16 weight copies (BASELINE config 4), software-pipelined:
//
//     y[T,N] = x[T,K] . W[N,K]^T (+ b)         both operands contraction-contiguous, fp32 accumulation, bf16 result
//
// and, called on the bf16 copy of W^T that the optimizer keeps (dhz_bf16_transpose_batched), the backward-data product
// dx[T,K] = dy[T,N] . (W^T)[K,N]^T.  The round-2 kernel of csrc/linear_bf16.hip stages both operands through registers into a two-slot
// LDS buffer behind a __syncthreads() per 64-element stage (0.19 of the bf16 MFMA peak on the config-4 step).  Here, as in
// csrc/split6_gemm.hip without the split:
//   * BOTH operands go global -> LDS by DMA (global_load_lds_dwordx4, 16 bytes per lane, no registers), into a ring of three stage
//     slots, two stages ahead of the multiplies; the XOR swizzle of the 64-byte-row images is applied on the SOURCE address (the DMA
//     writes lane-linear);
//   * 256 tokens x 128 features per workgroup of eight waves (4 x 2), 64 x 64 per wave: 32 MFMAs per wave and 64-element stage, 16
//     ds_read_b128 fragment reads (0.6 of the LDS read rate at the full matrix rate), 48 KB per stage from L2;
//   * persistent workgroups; the stream of (tile, stage) positions runs across tile boundaries, so the short contractions of this
//     model (64 .. 1024: one to sixteen stages per tile) never drain the pipeline;
//   * one raw s_barrier per stage with a COUNTED vmcnt wait in front of it (the stage after next and the tile's stores stay in flight);
//   * the MFMAs take the weight fragment as their first operand (a 16 x 16 block arrives transposed) and the weight rows are read in a
//     permuted order, so a lane owns 8 consecutive features of a token: 16-byte bf16 stores (as csrc/linear_bf16.hip).
#include <stdlib.h>
#include "common.h"
#include "tok_epilogue.h"

#ifndef BFP_ABL
#define BFP_ABL 0        // timing diagnostics (tools/variants.sh): 1 no stores, 2 no MFMAs, 4 no DMA, 8 no fragment reads
#endif

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

constexpr int NCOMP = 8;                               // multiplying waves (4 x 2)
constexpr int BM = 256, BN = 128, BKS = 64;            // tile; contraction elements per stage (two 32-element halves)
constexpr int WM = 4, WN = 4, WAVES_N = 2;             // 16 x 16 blocks per multiplying wave: 64 tokens x 64 features
constexpr int A_HALF = BM * 64, B_HALF = BN * 64;      // bytes of one half-stage image (64-byte rows = 32 bf16)
constexpr int SLOT = 2 * (A_HALF + B_HALF);            // 48 KB
constexpr int RING = 3;
constexpr int NSUB = (BM + BN) / 16;                   // 16-row groups per stage (one DMA instruction per group and half): 24

__device__ __forceinline__ f32x4 mfma_bf16(s16x8 a, s16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ void dma16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((glb_void*)g, (lds_void*)l, 16, 0, 0);
}
// 64-byte-row images: the four 16-byte chunks of a row are XORed with P[key], P = {0, 2, 3, 1}.  Token image: key = (row >> 2) & 3 -
// the 16 consecutive rows of a fragment read land on 16 different 16-byte bank groups.  Weight image: its fragment reads take the
// rows base + 8 q + p (q, p in 0..3; perm_row below), so the key is (row >> 3) & 3.
__device__ __forceinline__ int pxor(int key) { return (0x78 >> (2 * key)) & 3; }
__device__ __forceinline__ int swz_a(int row) { return pxor((row >> 2) & 3); }
__device__ __forceinline__ int swz_b(int row) { return pxor((row >> 3) & 3); }
__device__ __forceinline__ int perm_row(int b, int r) { return 32 * (b >> 1) + 8 * (r >> 2) + 4 * (b & 1) + (r & 3); }
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// Two kinds of waves.  vmcnt retires in issue order and counts loads and stores alike: a wave that both feeds the LDS ring and stores
// results has to sit out its own stores' way to HBM before it may trust a landed stage (measured with all ten duties in every wave,
// T = 131072, K = 256, N = 1024: 168 us, 51 us without the stores, spreading a tile's stores over the following stages 170).  So the
// NLOAD extra waves only issue the DMA (48 / NLOAD instructions per stage), wait for the previous stage with a counted vmcnt and meet
// the others at the barrier; the eight multiplying waves read fragments, multiply and store - and never wait on vmcnt at all.
// Things measured and NOT kept (T = 131072, K = 256, N = 1024; 268 MB of results): the loader waves' DMA alone 39 us, the multiplying
// waves' fragment reads + stores alone 59, + MFMAs 83, everything 144: the phases do not overlap - a CU's vector-memory pipeline is one
// in-order queue, the stores wait for HBM write bandwidth (every CU reaches its epilogue in the same microsecond) and the DMA behind
// them waits with them.  Spreading a tile's stores over the next tile's stages (1, 2 or 4 per wave and stage, results parked in 32
// registers): 144 -> 167 at K = 256, 86 -> 172 at K = 1024 - a store per stage queues behind the 48 DMA instructions of the stage.
// Non-temporal stores (NTS): T = 131072, K = 256, N = 1024: 144 -> 103 us; T = 524288, K = 128, N = 512: 190 -> 144; K = 64, N = 256: 102 -> 76 -
// every result of 200 MB and more gains 15 - 30 %, every smaller one loses 3 - 20 % (tools/bench_bf16_gemm.py with DHZ_BF16_PIPE_NT=0 / 1):
// a result that does not fit the 256 MB memory-side cache anyway should not be written through it.  Used from 192 MB.
template <int NLOAD, bool NTS>
__global__ __launch_bounds__(64 * (NCOMP + NLOAD), 1) void gemm_bf16_pipe_kernel(const uint16_t* __restrict__ A, int lda,
                                                                                 const uint16_t* __restrict__ B, int ldb,
                                                                                 const float* __restrict__ bias, uint16_t* __restrict__ C,
                                                                                 int ldc, int M, int NF, int KC, int tiles_n, int ntiles,
                                                                                 const TokEpi epi, int epi_on) {
    constexpr int NT = 64 * (NCOMP + NLOAD);
    constexpr int SPL = NSUB / NLOAD;                  // 16-row groups per loading wave
    constexpr int DPL = 2 * SPL;                       // its DMA instructions per stage
    static_assert(NSUB % NLOAD == 0 && DPL < 64, "vmcnt is a 6-bit counter");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* const bsm = reinterpret_cast<float*>(smem + RING * SLOT);
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int nst = KC / BKS;
    const int grid = gridDim.x;
    auto tile_of = [&](int i) -> int {
        const int lin = blockIdx.x + i * grid;
        if (lin >= ntiles) return -1;
        if ((grid & 7) == 0 && (ntiles & 7) == 0) return (lin & 7) * (ntiles >> 3) + (lin >> 3);      // XCD-contiguous tile ranges
        return lin;
    };
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
    auto valid = [&](const Pos& p) -> Pos { return p.tile >= 0 ? p : pz; };       // past the end: re-fetch the first stage (never read)
    Pos p0 = pz, p1 = next(p0), p2 = next(p1);

    if (w >= NCOMP) {
        // ------------------------------------------------------------------------------------------------ loading waves
        // group u of this wave: sub = (w - NCOMP) + NLOAD u of the 24; sub < 16: token rows 16 sub .., else weight rows 16 (sub - 16) ..;
        // lane -> row + lane / 4, source chunk (lane & 3) ^ swizzle; the two halves of a stage are 64 bytes apart at the source
        const int lw = w - NCOMP;
        int d_row[SPL], d_col[SPL], d_dst[SPL];
#pragma unroll
        for (int u = 0; u < SPL; ++u) {
            const int sub = lw + NLOAD * u;
            const bool tok = sub < BM / 16;
            const int row = 16 * (tok ? sub : sub - BM / 16) + (lane >> 2);
            d_row[u] = row;
            d_col[u] = 8 * ((lane & 3) ^ (tok ? swz_a(row) : swz_b(row)));
            d_dst[u] = tok ? sub * 1024 : 2 * A_HALF + (sub - BM / 16) * 1024;
        }
        const uint16_t* dp[SPL];
        int dp_ti = -1;
        auto dma_stage = [&](const Pos& q, int slot) {
            if constexpr (BFP_ABL & 4) return;
            if (q.ti != dp_ti) {                                                   // wave-uniform: the stream entered another tile
                dp_ti = q.ti;
#pragma unroll
                for (int u = 0; u < SPL; ++u) {
                    const bool tok = (lw + NLOAD * u) < BM / 16;
                    dp[u] = tok ? A + (size_t)min(q.m0 + d_row[u], M - 1) * lda + d_col[u] : B + (size_t)(q.n0 + d_row[u]) * ldb + d_col[u];
                }
            }
            unsigned char* S = smem + slot * SLOT;
            const int k0 = q.st * BKS;
#pragma unroll
            for (int u = 0; u < SPL; ++u) {
                const bool tok = (lw + NLOAD * u) < BM / 16;
                dma16(dp[u] + k0, S + d_dst[u]);
                dma16(dp[u] + k0 + 32, S + d_dst[u] + (tok ? A_HALF : B_HALF));
            }
        };
        dma_stage(p0, 0);
        dma_stage(valid(p1), 1);
        wait_vm<DPL>();                                 // stage 0 has landed
        __builtin_amdgcn_s_barrier();
        int slot = 0;
        for (;;) {
            int s2 = slot + 2; if (s2 >= RING) s2 -= RING;
            dma_stage(valid(p2), s2);                   // into the slot of stage p - 1: its readers passed the barrier
            if (p1.tile < 0) break;
            wait_vm<DPL>();                             // stage p + 1 (issued one iteration ago) has landed
            __builtin_amdgcn_s_barrier();
            p0 = p1; p1 = p2; p2 = next(p2);
            slot = slot + 1 == RING ? 0 : slot + 1;
        }
        wait_vm<0>();                                   // nothing may still be writing the LDS when the workgroup's allocation is released
        return;
    }

    // ---------------------------------------------------------------------------------------------------- multiplying waves
    const int i16 = lane & 15, g = lane >> 4;
    const int wm = w / WAVES_N, wn = w % WAVES_N;
    f32x4 acc[WM][WN];
#pragma unroll
    for (int a = 0; a < WM; ++a)
#pragma unroll
        for (int b = 0; b < WN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = t; i < NF; i += 64 * NCOMP) bsm[i] = bias ? bias[i] : 0.f;
    int fa[WM], fb[WN];                                 // fragment byte offsets inside a half-stage image
#pragma unroll
    for (int a = 0; a < WM; ++a) {
        const int row = (wm * WM + a) * 16 + i16;
        fa[a] = row * 64 + 16 * (g ^ swz_a(row));
    }
#pragma unroll
    for (int b = 0; b < WN; ++b) {
        const int row = wn * WN * 16 + perm_row(b, i16);
        fb[b] = 2 * A_HALF + row * 64 + 16 * (g ^ swz_b(row));
    }
    auto epilogue = [&](const Pos& q) {
        // acc[a][b][j] = C[token 16 a + i16][feature perm_row(b, 4 g + j)]: blocks 2 h, 2 h + 1 hold the 8 consecutive features 32 h + 8 g ..
        const int m0 = q.m0 + wm * WM * 16 + i16, n0 = q.n0 + wn * WN * 16 + 8 * g;
        uint16_t* c0 = C + (size_t)m0 * ldc + n0;
        const bool full = q.m0 + BM <= M;                                          // wave-uniform
        if (epi_on) {                                                              // the block's residual step (csrc/tok_epilogue.h)
            const int mbw = q.m0 + wm * WM * 16;
            int dst[WM];
            float sc;
            tok_epi_rows<WM>(epi, mbw < M ? mbw : 0, i16, dst, sc);
#pragma unroll
            for (int a = 0; a < WM; ++a)
                if (full || m0 + 16 * a < M) {
#pragma unroll
                    for (int h = 0; h < WN / 2; ++h)
                        tok_epi_store8_bf16(epi, C, (size_t)dst[a] * ldc + n0 + 32 * h, sc,
                                            acc[a][2 * h] + *reinterpret_cast<const f32x4*>(bsm + n0 + 32 * h),
                                            acc[a][2 * h + 1] + *reinterpret_cast<const f32x4*>(bsm + n0 + 32 * h + 4));
                }
        } else
#pragma unroll
        for (int a = 0; a < WM; ++a)
#pragma unroll
            for (int h = 0; h < WN / 2; ++h) {
                const f32x4 v0 = acc[a][2 * h] + *reinterpret_cast<const f32x4*>(bsm + n0 + 32 * h);
                const f32x4 v1 = acc[a][2 * h + 1] + *reinterpret_cast<const f32x4*>(bsm + n0 + 32 * h + 4);
                u32x4 r;
                r[0] = (uint32_t)f32_to_bf16(v0[0]) | ((uint32_t)f32_to_bf16(v0[1]) << 16);
                r[1] = (uint32_t)f32_to_bf16(v0[2]) | ((uint32_t)f32_to_bf16(v0[3]) << 16);
                r[2] = (uint32_t)f32_to_bf16(v1[0]) | ((uint32_t)f32_to_bf16(v1[1]) << 16);
                r[3] = (uint32_t)f32_to_bf16(v1[2]) | ((uint32_t)f32_to_bf16(v1[3]) << 16);
                u32x4* dst = reinterpret_cast<u32x4*>(c0 + (size_t)(16 * a) * ldc + 32 * h);
                if (BFP_ABL & 1) { if (r[0] == 0x12345678u) *dst = r; }            // (keeps the results alive)
                else if (full || m0 + 16 * a < M) {
                    if constexpr (NTS) __builtin_nontemporal_store(r, dst);
                    else *dst = r;
                }
            }
#pragma unroll
        for (int a = 0; a < WM; ++a)
#pragma unroll
            for (int b = 0; b < WN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the bias vector is in LDS
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    int slot = 0;
    for (;;) {
        const unsigned char* S = smem + slot * SLOT;
        // fragments: the weight side of both halves (2 x 4), the token side two at a time (the next one is read while the current one
        // multiplies) - 40 registers instead of 64 for everything at once
        s16x8 bf[2][WN], af[2];
        auto rd_b = [&](int h, int b) {
            if constexpr (BFP_ABL & 8) bf[h][b] = s16x8{(short)lane, 1, 2, 3, 4, 5, 6, 7};
            else bf[h][b] = *reinterpret_cast<const s16x8*>(S + h * B_HALF + fb[b]);
        };
        auto rd_a = [&](int k) {                        // step k = 4 h + a
            if constexpr (BFP_ABL & 8) af[k & 1] = s16x8{(short)lane, 1, 2, 3, 4, 5, 6, 7};
            else af[k & 1] = *reinterpret_cast<const s16x8*>(S + (k >> 2) * A_HALF + fa[k & 3]);
        };
#pragma unroll
        for (int b = 0; b < WN; ++b) rd_b(0, b);
        rd_a(0);
#pragma unroll
        for (int k = 0; k < 2 * WM; ++k) {
            if (k + 1 < 2 * WM) rd_a(k + 1);
            if (k == 0) { rd_b(1, 0); rd_b(1, 1); }
            if (k == 1) { rd_b(1, 2); rd_b(1, 3); }
#pragma unroll
            for (int b = 0; b < WN; ++b) {
                if constexpr (BFP_ABL & 2) asm volatile("" : "+v"(acc[k & 3][b]) : "v"(bf[k >> 2][b]), "v"(af[k & 1]));
                else acc[k & 3][b] = mfma_bf16(bf[k >> 2][b], af[k & 1], acc[k & 3][b]);
            }
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 5, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 3, 0); __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 3, 0); __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
#pragma unroll
        for (int k = 2; k < 7; ++k) { __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 4, 0); }
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (p0.st == nst - 1) epilogue(p0);
        if (p1.tile < 0) break;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        p0 = p1; p1 = p2; p2 = next(p2);
        slot = slot + 1 == RING ? 0 : slot + 1;
    }
}

template <int NLOAD, bool NTS>
void launch_pipe(const TokEpi& epi, int epi_on, const uint16_t* A, int lda, const uint16_t* B, int ldb, const float* bias, uint16_t* C, int ldc, int M, int NF, int KC,
                 int tiles_n, int ntiles, int grid, size_t smem, hipStream_t s) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_pipe_kernel<NLOAD, NTS>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)smem);
    hipLaunchKernelGGL((gemm_bf16_pipe_kernel<NLOAD, NTS>), dim3(grid), dim3(64 * (NCOMP + NLOAD)), smem, s, A, lda, B, ldb, bias, C, ldc, M, NF,
                       KC, tiles_n, ntiles, epi, epi_on);
}

}  // namespace

namespace {
// bf16 (round to nearest even) copies of the TRANSPOSES of a set of matrices that live in one fp32 buffer (the flat parameter buffer):
// matrix m occupies src[off, off + R C) as [R][C]; its transpose goes to the SAME offsets of dst as [C][R].  desc[m] = {off, R, C, first
// tile} (the table of dhz_split3_planes_t); 32 x 32 tiles through LDS; R, C multiples of 32.
__global__ __launch_bounds__(256) void bf16_transpose_batched_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst,
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
        for (int c = ty; c < 32; c += 8) dst[d0 + (size_t)c * R + tx] = f32_to_bf16(tile[tx][c]);
        __syncthreads();
    }
}
}  // namespace

extern "C" int dhz_bf16_transpose_batched(const float* src, void* dst, const int* desc, int nmat, int ntiles, void* stream) {
    const char* who = "dhz_bf16_transpose_batched";
    DHZ_REQUIRE(src && dst && desc && nmat > 0 && ntiles > 0, "%s: null pointer or empty table", who);
    const int cap = 8 * dhz_num_cus();
    hipLaunchKernelGGL(bf16_transpose_batched_kernel, dim3(ntiles < cap ? ntiles : cap), dim3(256), 0, (hipStream_t)stream, src,
                       (uint16_t*)dst, desc, nmat, ntiles);
    DHZ_CHECK_LAUNCH(who);
    return DHZ_OK;
}

// Called by csrc/linear_bf16.hip's forward dispatch for the shapes this kernel takes; returns false when it does not apply.
bool dhz_gemm_bf16_pipe_try(const uint16_t* A, int lda, const uint16_t* B, int ldb, const float* bias, uint16_t* C, int ldc, int M, int NF,
                            int KC, hipStream_t s, const TokEpi* epi) {
    const int epi_on = epi != nullptr;
    const TokEpi e = epi ? *epi : TokEpi{};
    static const int mode = getenv("DHZ_BF16_PIPE") ? atoi(getenv("DHZ_BF16_PIPE")) : 1;     // diagnostics: 0 = never, 2 = whenever legal
    if (!mode || NF % BN || KC % BKS || NF > 2048 || (long)ldb * NF >= (1L << 31)) return false;
    const int tiles_n = NF / BN, tiles_m = (M + BM - 1) / BM;
    const int ntiles = tiles_n * tiles_m;
    const int cus = dhz_num_cus();
    if (mode != 2 && ntiles < cus) return false;                  // fewer 256 x 128 tiles than CUs: the 128 x 128 kernel fills the chip better
    const size_t smem = (size_t)RING * SLOT + (size_t)NF * sizeof(float);
    const int grid = ntiles < cus ? ntiles : cus;
    static const int nt_env = (getenv("DHZ_BF16_PIPE_NT") && *getenv("DHZ_BF16_PIPE_NT")) ? atoi(getenv("DHZ_BF16_PIPE_NT")) : -1;       // diagnostics: 0 / 1 force
    const int nt_store = nt_env >= 0 ? nt_env : ((double)M * NF * 2 >= 192e6);
    // (the residual epilogue's output is the next kernel's input and its shortcut: never a non-temporal store)
    if (nt_store && !epi_on) launch_pipe<2, true>(e, epi_on, A, lda, B, ldb, bias, C, ldc, M, NF, KC, tiles_n, ntiles, grid, smem, s);
    else launch_pipe<2, false>(e, epi_on, A, lda, B, ldb, bias, C, ldc, M, NF, KC, tiles_n, ntiles, grid, smem, s);
    return true;
}

// (The weight gradient was given the same division of labour - two loading waves bring 64-token stages of dy and x [token][feature]
// into a three-slot ring by DMA, eight multiplying waves read them with ds_read_b64_tr_b16, 256 x 128 / 128 x 256 tiles, the bias gradient
// as an MFMA with a fragment of ones - and measured against csrc/linear_bf16.hip's register-staged 128 x 128 kernel on the config-4
// shapes (tools/bench_bf16_wgrad.py): T = 8192, N = 4096, K = 1024: 99.6 -> 89.5 us, T = 131072, N = 1024, K = 256: 113 -> 100, most
// others within 3 %, several small ones 10 % slower; sum over the step's 36 shapes 2515 -> 2483 us.  Removed.)
