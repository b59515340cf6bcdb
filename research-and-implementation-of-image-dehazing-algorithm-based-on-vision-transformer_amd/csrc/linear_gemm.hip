// Forward and backward-data GEMMs of every token-major nn.Linear on the path, on the fp32 matrix pipe:
//     forward        y[T,N]  = x[T,K]  . W[N,K]^T + b        (WT = true : W rows are output features, K contiguous)
//     backward-data  dx[T,K] = dy[T,N] . W[N,K]              (WT = false: W rows are the CONTRACTION index)
// Both are "tall" GEMMs: T = B*H*W tokens (2k .. 524k) against 32..2048 features, so the block tile is 32*WM tokens x
// 32*WN features with four waves as 2 x 2, the contraction streamed in 32-deep stages through double-buffered LDS
// (coalesced float4 global loads -> registers -> LDS while the previous stage feeds v_mfma_f32_16x16x4_f32; one barrier
// per stage).  LDS images:
//   * token operand (and W when WT): [row][32 k] with the 16-byte k-quads of a row XOR-swizzled by (row >> 1) & 7, so
//     that a lane's four consecutive k are ONE conflict-free ds_read_b128; MFMA j of a 16-deep step then contracts over
//     k = 16 s + 4 g + j (g = lane >> 4) - any order of k is valid as long as both operands use the same one.
//   * W when !WT: [32 k][BN + 4] (row stride = 4 mod 8 floats -> the g = 0 / 1 halves of a ds_read_b32 hit disjoint banks).
// Workgroups are persistent (two per CU) and pipeline across their tiles; the epilogue (+ bias) stores straight from the
// accumulators.
#include <stdio.h>
#include <stdlib.h>
#include "common.h"

namespace {

constexpr int BK = 32;
#ifndef DHZ_GEMM_ABL
#define DHZ_GEMM_ABL 0           // timing diagnostics (tools/gemm_phases.sh): 1 = no epilogue stores, 2 = no MFMAs, 4 = no global operand loads
#endif

#ifdef DHZ_GEMM_STAMP      // timing diagnostics only (tools/micro/stamp_gemm.py builds its own copy): s_memtime at the phase boundaries
__device__ long long* g_stamp = nullptr;
#define STAMP(slot)                                                                                  \
    do {                                                                                             \
        if (g_stamp && blockIdx.x == DHZ_GEMM_STAMP && lane == 0 && nstamp < 40)                     \
            g_stamp[(w * 40 + nstamp) * 8 + (slot)] = (long long)__builtin_amdgcn_s_memtime();       \
    } while (0)
#else
#define STAMP(slot)
#endif

template <int WM, int WN, bool WT>
__global__ __launch_bounds__(256) void linear_gemm_kernel(const float* __restrict__ A, int lda,
                                                          const float* __restrict__ W, int ldw,
                                                          const float* __restrict__ bias, float* __restrict__ Y, int ldy,
                                                          int M, int N, int K, int tiles_n, int ntiles) {
    constexpr int BM = 32 * WM, BN = 32 * WN;
    constexpr int SBN = BN + 4;                                   // !WT image row stride
    constexpr int A_FLOATS = BM * BK;
    constexpr int B_FLOATS = WT ? BN * BK : BK * SBN;
    constexpr int STAGE = A_FLOATS + B_FLOATS;
    constexpr int NA = BM * 8 / 256;                              // float4 per thread per stage (= WM)
    constexpr int NB = WN;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int i16 = lane & 15, g = lane >> 4;
    const int wm = w >> 1, wn = w & 1;
    const int nst = K / BK;
    constexpr int abl = DHZ_GEMM_ABL;

    // Persistent workgroups: a workgroup walks the tiles bid, bid + grid, ... and treats their (tile, stage) pairs as ONE
    // stream of stages, so the loads of the next tile's first stage are in flight while the current tile finishes and its
    // epilogue stores drain behind the next tile's matrix work.  XCD-aware numbering: workgroups b and b + 8 share an XCD
    // (round-robin dispatch), and consecutive LOGICAL tiles share their token rows across the tn sweep: map the tiles so that
    // an XCD owns a contiguous range of them (the re-read of A then comes from that XCD's L2).
    const int grid = gridDim.x;
    auto tile_of = [&](int i) -> int {                            // i-th tile of this workgroup, or -1
        const int lin = blockIdx.x + i * grid;
        if (lin >= ntiles) return -1;
        if ((grid & 7) == 0 && (ntiles & 7) == 0) return (lin & 7) * (ntiles >> 3) + (lin >> 3);
        return lin;
    };

    // The bias enters as the INITIAL VALUE of the accumulators (exactly the addend of y = x W^T + b), read from an LDS copy of
    // the whole bias vector: no global load is issued - or waited for - between a tile's epilogue stores and the next tile's
    // matrix work.  (A bias load at the top of a tile made hipcc wait vmcnt(0) there: on gfx9 stores count in vmcnt, so every
    // tile began by waiting for the ACKNOWLEDGEMENT of the previous tile's 64 stores - the "matrix phase and store burst add
    // up" finding of round 2.)
    float* bsm = smem + 2 * STAGE;
    for (int i = t; i < N; i += 256) bsm[i] = bias ? bias[i] : 0.f;      // visible after the first barrier below
    f32x4 acc[WM][WN];
    auto acc_init = [&](int tile) {
        const float* bp = bsm + (tile % tiles_n) * BN + wn * WN * 16 + i16;
#pragma unroll
        for (int b = 0; b < WN; ++b) {
            const float bv = bp[16 * b];
#pragma unroll
            for (int a = 0; a < WM; ++a) acc[a][b] = f32x4{bv, bv, bv, bv};
        }
    };

    f32x4 ra[NA], rb[NB];
    // per-tile operand row pointers (recomputed only when the stream moves on to another tile): a stage adds its k offset
    const float* pa[NA];
    const float* pb[NB];
    auto set_tile = [&](int tile) {
        const int tn = tile % tiles_n, tm = tile / tiles_n;
        const int m0 = tm * BM, n0 = tn * BN;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int e = t + 256 * i, row = e >> 3, q = e & 7;
            pa[i] = A + (size_t)min(m0 + row, M - 1) * lda + 4 * q;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int e = t + 256 * i;
            if (WT) pb[i] = W + (size_t)(n0 + (e >> 3)) * ldw + 4 * (e & 7);
            else pb[i] = W + (size_t)(e / (BN / 4)) * ldw + n0 + 4 * (e % (BN / 4));
        }
    };
    auto gload = [&](int k0) {
        if (abl & 4) return;
#pragma unroll
        for (int i = 0; i < NA; ++i) ra[i] = *reinterpret_cast<const f32x4*>(pa[i] + k0);
#pragma unroll
        for (int i = 0; i < NB; ++i) rb[i] = *reinterpret_cast<const f32x4*>(pb[i] + (WT ? (size_t)k0 : (size_t)k0 * ldw));
    };
    auto swrite = [&](int buf) {
        float* As = smem + buf * STAGE;
        float* Bs = As + A_FLOATS;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int e = t + 256 * i, row = e >> 3, q = e & 7;
            *reinterpret_cast<f32x4*>(&As[row * BK + 4 * (q ^ ((row >> 1) & 7))]) = ra[i];
        }
        if (WT) {
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int e = t + 256 * i, row = e >> 3, q = e & 7;
                *reinterpret_cast<f32x4*>(&Bs[row * BK + 4 * (q ^ ((row >> 1) & 7))]) = rb[i];
            }
        } else {
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int e = t + 256 * i;
                const int row = e / (BN / 4), c4 = e % (BN / 4);
                *reinterpret_cast<f32x4*>(&Bs[row * SBN + 4 * c4]) = rb[i];
            }
        }
    };

    int ti = 0, tile = tile_of(0);
    if (tile < 0) return;
    set_tile(tile);
    gload(0);
    swrite(0);
    __syncthreads();
    acc_init(tile);
    const int sw = (i16 >> 1) & 7;
    int buf = 0;
    int nstamp = 0;
    (void)nstamp;
    while (true) {
        const int ntile = tile_of(ti + 1);
        const int tn = tile % tiles_n, tm = tile / tiles_n;
        for (int st = 0; st < nst; ++st) {
            // next stage of the stream: the same tile, or the first stage of this workgroup's next tile
            const bool last = st + 1 == nst;
            const bool more = !last || ntile >= 0;
            STAMP(0);
            if (!last) gload((st + 1) * BK);
            else if (ntile >= 0) { set_tile(ntile); gload(0); }
            STAMP(1);
            const float* As = smem + buf * STAGE + (wm * WM * 16 + i16) * BK;
            const float* Bs = smem + buf * STAGE + A_FLOATS;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                f32x4 af[WM], bf[WN];
#pragma unroll
                for (int a = 0; a < WM; ++a)
                    af[a] = *reinterpret_cast<const f32x4*>(&As[a * 16 * BK + 4 * ((4 * s + g) ^ sw)]);
                if (WT) {
#pragma unroll
                    for (int b = 0; b < WN; ++b)
                        bf[b] = *reinterpret_cast<const f32x4*>(&Bs[((wn * WN + b) * 16 + i16) * BK + 4 * ((4 * s + g) ^ sw)]);
                } else {
#pragma unroll
                    for (int b = 0; b < WN; ++b)
#pragma unroll
                        for (int j = 0; j < 4; ++j) bf[b][j] = Bs[(16 * s + 4 * g + j) * SBN + (wn * WN + b) * 16 + i16];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int a = 0; a < WM; ++a)
#pragma unroll
                        for (int b = 0; b < WN; ++b)
                            if (abl & 2) acc[a][b][j] += af[a][j] + bf[b][j];
                            else acc[a][b] = mfma16(af[a][j], bf[b][j], acc[a][b]);
            }
            STAMP(2);
            if (more) {
                swrite(buf ^ 1);
                STAMP(3);
                __syncthreads();
                buf ^= 1;
            }
            STAMP(4);
            ++nstamp;
        }
        STAMP(5);
        // ---- tile epilogue straight from the accumulators; the stores drain behind the next tile's matrix work (its first
        //      stage is already in LDS).  acc[a][b][j] = C[16 a + 4 g + j][16 b + i16]: the 16 lanes of a row write 64 contiguous
        //      bytes, the b sweep completes the lines.
        //      Full tiles (every tile unless T is ragged) store without per-row guards: behind a divergent guard hipcc waits for
        //      vmcnt(0) - i.e. for the acknowledgement of ALL earlier stores - before every guarded group, which serialised the
        //      epilogue into 16 memory round trips per tile.
        //      (Swapping the MFMA operands turns a lane's four values into four consecutive features = one 16-byte store: 10-20 %
        //      faster in isolation on outputs of <= 96 features, slower on wide ones, and no gain in the training step - not kept.)
        const bool full = tm * BM + BM <= M;                          // wave-uniform
        {
            const int m0 = tm * BM + wm * WM * 16 + 4 * g, n0 = tn * BN + wn * WN * 16 + i16;
            float* y0 = Y + (size_t)m0 * ldy + n0;
            if (full) {
#pragma unroll
                for (int a = 0; a < WM; ++a)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int b = 0; b < WN; ++b)
                            if (!(abl & 1) || acc[a][0][0] == 12345.678f) y0[(size_t)(16 * a + j) * ldy + 16 * b] = acc[a][b][j];
            } else {
#pragma unroll
                for (int a = 0; a < WM; ++a)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (m0 + 16 * a + j < M) {
#pragma unroll
                            for (int b = 0; b < WN; ++b) y0[(size_t)(16 * a + j) * ldy + 16 * b] = acc[a][b][j];
                        }
            }
        }
        STAMP(6);
        ++nstamp;
        if (ntile < 0) break;
        tile = ntile;
        acc_init(tile);
        ++ti;
    }
}

template <int WM, int WN, bool WT>
void launch(const float* A, int lda, const float* W, int ldw, const float* bias, float* Y, int ldy, int M, int N, int K,
            hipStream_t s) {
    constexpr int BM = 32 * WM, BN = 32 * WN;
    constexpr size_t stage = (size_t)(BM * BK + (WT ? BN * BK : BK * (BN + 4))) * sizeof(float);
    const size_t smem = 2 * stage + (size_t)N * sizeof(float);       // two operand stages + the bias vector
    const int tiles_n = N / BN, tiles_m = (M + BM - 1) / BM;
    const int ntiles = tiles_n * tiles_m;
    // two workgroups per CU when the LDS allows (it does for every tile shape: <= 66.5 KiB per workgroup)
#ifdef DHZ_DIAG
    const int slots = getenv("DHZ_GEMM_SLOTS") ? atoi(getenv("DHZ_GEMM_SLOTS")) : 2 * dhz_num_cus();
#else
    const int slots = 2 * dhz_num_cus();
#endif
    const int grid = ntiles < slots ? ntiles : slots;
    if (smem > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_gemm_kernel<WM, WN, WT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL((linear_gemm_kernel<WM, WN, WT>), dim3(grid), dim3(256), smem, s, A, lda, W, ldw, bias, Y, ldy, M, N,
                       K, tiles_n, ntiles);
}

template <bool WT>
int dispatch(const char* who, const float* A, int lda, const float* W, int ldw, const float* bias, float* Y, int ldy, int M,
             int N, int K, hipStream_t s) {
    DHZ_REQUIRE(A && W && Y, "%s: null pointer", who);
    DHZ_REQUIRE(M > 0 && N > 0 && K > 0 && N % 32 == 0 && K % 32 == 0, "%s: T=%d N=%d K=%d (N, K must be multiples of 32)", who,
                M, N, K);
    DHZ_REQUIRE(N <= 16384, "%s: N=%d (at most 16384 output features: the bias vector is staged in LDS)", who, N);
    DHZ_REQUIRE(lda % 4 == 0 && ldy % 4 == 0 && ldw % 4 == 0 && lda >= K && ldy >= N, "%s: bad leading dimensions", who);
    DHZ_REQUIRE((((uintptr_t)A | (uintptr_t)Y) & 15) == 0 && (((uintptr_t)W | (uintptr_t)bias) & 3) == 0,
                "%s: activations must be 16-byte aligned (weights: 4-byte)", who);
    // largest tile that still gives every CU two blocks: 128 x 128 down to 64 x 32 (tile width must divide N)
    int wm = 2, wn = 1;
    {
        static const int cand[8][2] = {{4, 4}, {4, 3}, {4, 2}, {2, 4}, {2, 3}, {2, 2}, {4, 1}, {2, 1}};
        long best_blocks = -1;
        const long slots = 2 * dhz_num_cus();
        for (int i = 0; i < 8; ++i) {
            const int a = cand[i][0], b = cand[i][1];
            if (N % (32 * b)) continue;
            const long blocks = (long)((M + 32 * a - 1) / (32 * a)) * (N / (32 * b));
            if (blocks >= slots) { wm = a; wn = b; break; }    // two resident workgroups per CU (256 / 384 / 512: 1838 / 1822 / 1809 us over the deep stages)
            if (blocks > best_blocks) { best_blocks = blocks; wm = a; wn = b; }
        }
    }
#ifdef DHZ_DIAG
    if (const char* e = getenv("DHZ_GEMM_TILE")) {               // "wm,wn" (ignored where wn does not divide N / 32)
        int a = 0, b = 0;
        if (sscanf(e, "%d,%d", &a, &b) == 2 && N % (32 * b) == 0) { wm = a; wn = b; }
    }
#endif
#define CASE(a, b) \
    if (wm == a && wn == b) launch<a, b, WT>(A, lda, W, ldw, bias, Y, ldy, M, N, K, s);
    CASE(4, 1) CASE(4, 2) CASE(4, 3) CASE(4, 4) CASE(2, 1) CASE(2, 2) CASE(2, 3) CASE(2, 4)
#undef CASE
    DHZ_CHECK_LAUNCH(who);
    return DHZ_OK;
}

}  // namespace

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
