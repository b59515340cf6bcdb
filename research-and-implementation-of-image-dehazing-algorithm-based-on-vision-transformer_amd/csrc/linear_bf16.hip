// Token-Linear GEMMs for bf16 activations / bf16 weight copies with fp32 accumulation (BASELINE config 4) on
// v_mfma_f32_16x16x32_bf16:
//     forward        y[T,N]  = x[T,K]  . W[N,K]^T + b      both operands K-contiguous: fragments by ds_read_b128
//     backward-data  dx[T,K] = dy[T,N] . W[N,K]            W's rows are the CONTRACTION index: B fragments by the
//                                                          hardware transpose read ds_read_b64_tr_b16 (no transposed copy)
//     weight grad    dW[N,K] += dy^T[N,T] . x[T,K]         the contraction runs over token ROWS of both operands: both
//                    db[N]   += sum_t dy[t,:]              fragments by transpose reads; split over T, fp32 atomics
// Operand fragment of the 16x16x32 MFMA: lane (i = lane & 15, g = lane >> 4) holds the 8 bf16 k = 8 g .. 8 g + 7 of row
// (A) / column (B) i.  LDS images (64 contraction elements per stage):
//   * K-contiguous operand: [row][64 bf16 = 128 B], the eight 16-byte chunks of a row XOR-swizzled by (row >> 1) & 7
//     -> conflict-free ds_read_b128;
//   * contraction-major operand: [64 contraction rows][F features] with F = 128 (256-B rows, chunk ^ (((row & 3) << 2) |
//     ((row >> 2) & 3))) or F = 64 (128-B rows, chunk ^ ((row & 2) | ((row & 8) >> 1))) -> conflict-free transpose reads:
//     lane 4 q + p of a 16-lane group addresses row r0 + q, columns 4 p .. 4 p + 3 of a 4 x 16 block and receives column
//     (lane & 15), rows r0 .. r0 + 3.
// Forward / backward-data: persistent workgroups pipelined across tiles (as csrc/linear_gemm.hip); outputs leave as bf16.
#include "common.h"
#include "tok_epilogue.h"

#ifndef BF_ABL
#define BF_ABL 0        // timing diagnostics (tools/variants.sh): 1 no epilogue stores, 2 no MFMAs, 4 no global operand loads, 8 weight gradient: no atomics
#endif

bool dhz_gemm_bf16_pipe_try(const uint16_t* A, int lda, const uint16_t* B, int ldb, const float* bias, uint16_t* C, int ldc, int M, int NF,
                            int KC, hipStream_t s, const TokEpi* epi);          // csrc/gemm_bf16_pipe.hip (epi: residual epilogue or null)

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 64;
#ifndef BF_WABL
#define BF_WABL 0       // weight gradient diagnostics: 1 no MFMAs (and no fragment reads), 2 no global loads after the first stage
#endif
#ifndef BF_WPAIR
#define BF_WPAIR 1      // weight gradient: two slabs per workgroup of 512 threads (half the atomics) where a tile has two slabs or more
#endif
#ifndef BF_WPAIR_MAX_STAGES
#define BF_WPAIR_MAX_STAGES 32
#endif
#ifndef BF_WSPLIT
#define BF_WSPLIT 2     // workgroups per CU the token split aims at
#endif

__device__ __forceinline__ f32x4 mfma_bf16(s16x8 a, s16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// byte offset of 16-byte chunk ch of row `row` in a K-contiguous image (128-B rows)
__device__ __forceinline__ int off_row(int row, int ch) { return row * 128 + 16 * (ch ^ ((row >> 1) & 7)); }
// The same for the feature-major B image of the forward / backward-data kernels, whose rows are read in the order
//     fragment block b, fragment row r  <->  tile feature 32 (b >> 1) + 8 (r >> 2) + 4 (b & 1) + (r & 3)
// (a lane then owns 8 CONSECUTIVE features per pair of blocks and the four lanes of a token 32: 16-byte stores in 64-byte runs
// instead of 8-byte stores in 32-byte runs; any assignment of features to MFMA columns is as good as another).  The 16 rows of a
// fragment are base + 8 q + p (q, p in 0..3): the key (2 q + (p >> 1)) keeps them on distinct chunk positions in pairs, as
// (row >> 1) & 7 does for 16 consecutive rows.
__device__ __forceinline__ int off_rowp(int row, int ch) { return row * 128 + 16 * (ch ^ ((((row >> 3) & 3) << 1) | ((row >> 1) & 1))); }
__device__ __forceinline__ int perm_row(int b, int r) { return 32 * (b >> 1) + 8 * (r >> 2) + 4 * (b & 1) + (r & 3); }
// byte offset of 16-byte chunk ch of contraction row `row` in a contraction-major image with F features per row
template <int F>
__device__ __forceinline__ int off_tr(int row, int ch) {
    if (F == 128) return row * 256 + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3)));
    return row * 128 + 16 * (ch ^ ((row & 2) | ((row & 8) >> 1)));
}
// fragment (8 contraction rows r0 .. r0 + 7 of feature column 16 cb + (lane & 15)) from a contraction-major image
template <int F>
__device__ __forceinline__ s16x8 tr_frag(const unsigned char* img, int r0, int cb, int lane) {
    const int m = lane & 15, q = m >> 2, p = m & 3;
    typedef s16x4 __attribute__((address_space(3))) * lds_ptr;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(img + off_tr<F>(r0 + q, 2 * cb + (p >> 1)) + 8 * (p & 1)));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(img + off_tr<F>(r0 + 4 + q, 2 * cb + (p >> 1)) + 8 * (p & 1)));
    return s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// ------------------------------------------------------------------------------------------------ forward / backward-data
// C[M,N] = A[M,K] . op(B) (+ bias), A K-contiguous.  BTR = false: B is [N][K] (forward); true: B is [K][N] (backward-data).
template <int WM, int WN, bool BTR>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(const uint16_t* __restrict__ A, int lda,
                                                        const uint16_t* __restrict__ B, int ldb,
                                                        const float* __restrict__ bias, uint16_t* __restrict__ C, int ldc,
                                                        int M, int N, int K, int tiles_n, int ntiles, const TokEpi epi, int epi_on) {
    constexpr int BM = 32 * WM, BN = 32 * WN;
    constexpr int A_BYTES = BM * 128;
    constexpr int B_BYTES = BTR ? BK * BN * 2 : BN * 128;
    constexpr int STAGE = A_BYTES + B_BYTES;
    constexpr int NA = WM, NB = WN;                              // 16-byte chunks per thread per stage
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int i16 = lane & 15, g = lane >> 4;
    const int wm = w >> 1, wn = w & 1;
    const int nst = K / BK;
    const int grid = gridDim.x;
    auto tile_of = [&](int i) -> int {
        const int lin = blockIdx.x + i * grid;
        if (lin >= ntiles) return -1;
        if ((grid & 7) == 0 && (ntiles & 7) == 0) return (lin & 7) * (ntiles >> 3) + (lin >> 3);
        return lin;
    };
    f32x4 acc[WM][WN];
#pragma unroll
    for (int a = 0; a < WM; ++a)
#pragma unroll
        for (int b = 0; b < WN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    u32x4 ra[NA], rb[NB];
    const uint16_t* pa[NA];
    const uint16_t* pb[NB];
    auto set_tile = [&](int tile) {
        const int tn = tile % tiles_n, tm = tile / tiles_n;
        const int m0 = tm * BM, n0 = tn * BN;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int e = t + 256 * i;
            pa[i] = A + (size_t)min(m0 + (e >> 3), M - 1) * lda + 8 * (e & 7);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int e = t + 256 * i;
            if (BTR) pb[i] = B + (size_t)(e / (BN / 8)) * ldb + n0 + 8 * (e % (BN / 8));
            else pb[i] = B + (size_t)(n0 + (e >> 3)) * ldb + 8 * (e & 7);
        }
    };
    auto gload = [&](int k0) {
        if (BF_ABL & 4) return;
#pragma unroll
        for (int i = 0; i < NA; ++i) ra[i] = *reinterpret_cast<const u32x4*>(pa[i] + k0);
#pragma unroll
        for (int i = 0; i < NB; ++i) rb[i] = *reinterpret_cast<const u32x4*>(pb[i] + (BTR ? (size_t)k0 * ldb : (size_t)k0));
    };
    auto swrite = [&](int buf) {
        unsigned char* As = smem + buf * STAGE;
        unsigned char* Bs = As + A_BYTES;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int e = t + 256 * i;
            *reinterpret_cast<u32x4*>(As + off_row(e >> 3, e & 7)) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int e = t + 256 * i;
            if (BTR) {
                // feature chunk fc = 4 B + q of contraction row kr: its two 4-feature halves belong to fragment blocks 2 B and 2 B + 1
                // (physical columns 32 B + 4 q + p and 32 B + 16 + 4 q + p): two 8-byte writes
                const int kr = e / (BN / 8), fc = e % (BN / 8);
                const int B2 = fc >> 2, q = fc & 3;
                unsigned char* d0 = Bs + off_tr<BN>(kr, 4 * B2 + (q >> 1)) + 8 * (q & 1);
                unsigned char* d1 = Bs + off_tr<BN>(kr, 4 * B2 + 2 + (q >> 1)) + 8 * (q & 1);
                *reinterpret_cast<uint2*>(d0) = uint2{rb[i][0], rb[i][1]};
                *reinterpret_cast<uint2*>(d1) = uint2{rb[i][2], rb[i][3]};
            } else *reinterpret_cast<u32x4*>(Bs + off_rowp(e >> 3, e & 7)) = rb[i];
        }
    };

    int ti = 0, tile = tile_of(0);
    if (tile < 0) return;
    set_tile(tile);
    gload(0);
    swrite(0);
    __syncthreads();
    const int sw = (i16 >> 1) & 7;
    int buf = 0;
    while (true) {
        const int ntile = tile_of(ti + 1);
        const int tn = tile % tiles_n, tm = tile / tiles_n;
        f32x4 bv[WN];                                                    // bias of this tile's columns in the epilogue's lane layout
#pragma unroll
        for (int b = 0; b < WN; ++b)
#pragma unroll
            for (int j = 0; j < 4; ++j) bv[b][j] = bias ? bias[tn * BN + wn * WN * 16 + perm_row(b, 4 * g + j)] : 0.f;
        for (int st = 0; st < nst; ++st) {
            const bool last = st + 1 == nst;
            const bool more = !last || ntile >= 0;
            if (!last) gload((st + 1) * BK);
            else if (ntile >= 0) { set_tile(ntile); gload(0); }
            const unsigned char* As = smem + buf * STAGE;
            const unsigned char* Bs = As + A_BYTES;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                s16x8 af[WM], bf[WN];
#pragma unroll
                for (int a = 0; a < WM; ++a)
                    af[a] = *reinterpret_cast<const s16x8*>(As + (wm * WM * 16 + a * 16 + i16) * 128 + 16 * ((4 * s + g) ^ sw));
#pragma unroll
                for (int b = 0; b < WN; ++b) {
                    if (BTR) bf[b] = tr_frag<BN>(Bs, 32 * s + 8 * g, wn * WN + b, lane);
                    else bf[b] = *reinterpret_cast<const s16x8*>(Bs + off_rowp(wn * WN * 16 + perm_row(b, i16), 4 * s + g));
                }
#pragma unroll
                for (int a = 0; a < WM; ++a)
#pragma unroll
                    for (int b = 0; b < WN; ++b) {
                        if (BF_ABL & 2) acc[a][b][0] += (float)(af[a][0] + bf[b][0]);
                        else acc[a][b] = mfma_bf16(bf[b], af[a], acc[a][b]);    // D = C^T block (epilogue)
                    }
            }
            if (more) {
                swrite(buf ^ 1);
                __syncthreads();
                buf ^= 1;
            }
        }
        {   // epilogue from the accumulators.  The MFMAs take the B (feature) fragment as their first operand, so a 16 x 16 block
            // arrives transposed: acc[a][b][j] = C[token 16 a + i16][feature perm_row(b, 4 g + j)] - with the permuted feature order a
            // lane owns the 8 consecutive features 32 h + 8 g .. + 7 in blocks 2 h, 2 h + 1 and stores them as ONE 16-byte vector; the
            // four lanes of a token cover 64 contiguous bytes per store (8-byte stores of 32-byte pieces ran the wide outputs at
            // 3.7 TB/s stores-only, 40 - 50 % behind the tuned library on them).  Full tiles store without per-row guards and the
            // bias was loaded at the top of the tile: behind a divergent guard or a load hipcc waits for vmcnt(0).
            const int m0 = tm * BM + wm * WM * 16 + i16, n0 = tn * BN + wn * WN * 16 + 8 * g;
            uint16_t* c0 = C + (size_t)m0 * ldc + n0;
            auto pack = [&](int a, int h) {
                const f32x4 v0 = acc[a][2 * h] + bv[2 * h], v1 = acc[a][2 * h + 1] + bv[2 * h + 1];
                u32x4 r;
                r[0] = (uint32_t)f32_to_bf16(v0[0]) | ((uint32_t)f32_to_bf16(v0[1]) << 16);
                r[1] = (uint32_t)f32_to_bf16(v0[2]) | ((uint32_t)f32_to_bf16(v0[3]) << 16);
                r[2] = (uint32_t)f32_to_bf16(v1[0]) | ((uint32_t)f32_to_bf16(v1[1]) << 16);
                r[3] = (uint32_t)f32_to_bf16(v1[2]) | ((uint32_t)f32_to_bf16(v1[3]) << 16);
                return r;
            };
            if (epi_on) {                                                // the block's residual step (csrc/tok_epilogue.h)
                const int mbw = __builtin_amdgcn_readfirstlane(tm * BM + wm * WM * 16);
                int dst[WM];
                float sc;
                tok_epi_rows<WM>(epi, mbw < M ? mbw : 0, i16, dst, sc);
#pragma unroll
                for (int a = 0; a < WM; ++a)
                    if (m0 + 16 * a < M) {
#pragma unroll
                        for (int h = 0; h < WN / 2; ++h)
                            tok_epi_store8_bf16(epi, C, (size_t)dst[a] * ldc + n0 + 32 * h, sc, acc[a][2 * h] + bv[2 * h], acc[a][2 * h + 1] + bv[2 * h + 1]);
                    }
            } else if (tm * BM + BM <= M) {                              // wave-uniform
#pragma unroll
                for (int a = 0; a < WM; ++a)
#pragma unroll
                    for (int h = 0; h < WN / 2; ++h)
                        if (!(BF_ABL & 1) || acc[a][0][0] == 12345.678f) *reinterpret_cast<u32x4*>(c0 + (size_t)(16 * a) * ldc + 32 * h) = pack(a, h);
            } else {
#pragma unroll
                for (int a = 0; a < WM; ++a)
                    if (m0 + 16 * a < M) {
#pragma unroll
                        for (int h = 0; h < WN / 2; ++h) *reinterpret_cast<u32x4*>(c0 + (size_t)(16 * a) * ldc + 32 * h) = pack(a, h);
                    }
            }
#pragma unroll
            for (int a = 0; a < WM; ++a)
#pragma unroll
                for (int b = 0; b < WN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (ntile < 0) break;
        tile = ntile;
        ++ti;
    }
}

template <int WM, int WN, bool BTR>
void launch_gemm(const uint16_t* A, int lda, const uint16_t* B, int ldb, const float* bias, uint16_t* C, int ldc, int M, int N,
                 int K, const TokEpi& epi, int epi_on, hipStream_t s) {
    constexpr int BM = 32 * WM, BN = 32 * WN;
    constexpr size_t smem = 2 * (size_t)(BM * 128 + (BTR ? BK * BN * 2 : BN * 128));
    const int tiles_n = N / BN, tiles_m = (M + BM - 1) / BM;
    const int ntiles = tiles_n * tiles_m;
    const int slots = 2 * dhz_num_cus();                          // two resident workgroups per CU
    const int grid = ntiles < slots ? ntiles : slots;
    if (smem > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_kernel<WM, WN, BTR>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL((gemm_bf16_kernel<WM, WN, BTR>), dim3(grid), dim3(256), smem, s, A, lda, B, ldb, bias, C, ldc, M, N, K,
                       tiles_n, ntiles, epi, epi_on);
}

template <bool BTR>
int dispatch_gemm(const char* who, const uint16_t* A, int lda, const uint16_t* B, int ldb, const float* bias, uint16_t* C, int ldc,
                  int M, int N, int K, hipStream_t s, const TokEpi* epi = nullptr) {
    const int epi_on = epi != nullptr;
    const TokEpi e = epi ? *epi : TokEpi{};
    if (epi_on) {
        const char* bad = tok_epi_check(e, M);
        DHZ_REQUIRE(!bad, "%s: %s", who, bad);
        DHZ_REQUIRE(((uintptr_t)e.res & 15) == 0, "%s: the shortcut must be 16-byte aligned", who);
    }
    DHZ_REQUIRE(A && B && C, "%s: null pointer", who);
    DHZ_REQUIRE(M > 0 && N > 0 && K > 0 && N % 64 == 0 && K % 64 == 0, "%s: T=%d N=%d K=%d (N, K must be multiples of 64)", who, M, N,
                K);
    DHZ_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && ldc % 8 == 0 && ldc >= N && lda >= K, "%s: bad leading dimensions", who);
    DHZ_REQUIRE((((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) & 15) == 0, "%s: operands must be 16-byte aligned", who);
    if constexpr (!BTR) {
        // the software-pipelined 256 x 128 kernel (csrc/gemm_bf16_pipe.hip) where the problem has a tile per CU
        if (dhz_gemm_bf16_pipe_try(A, lda, B, ldb, bias, C, ldc, M, N, K, s, epi)) {
            DHZ_CHECK_LAUNCH(who);
            return DHZ_OK;
        }
    }
    const int wn = N % 128 == 0 ? 4 : 2;
    const long blocks128 = (long)((M + 127) / 128) * (N / (32 * wn));
    const int wm = blocks128 >= 256 ? 4 : 2;
#define CASE(a, b) \
    if (wm == a && wn == b) launch_gemm<a, b, BTR>(A, lda, B, ldb, bias, C, ldc, M, N, K, e, epi_on, s);
    CASE(4, 4) CASE(4, 2) CASE(2, 4) CASE(2, 2)
#undef CASE
    DHZ_CHECK_LAUNCH(who);
    return DHZ_OK;
}

// ------------------------------------------------------------------------------------------------ weight gradient
constexpr int MAXMAT = 4;
struct WgradOut {
    float* dw[MAXMAT];
    float* db[MAXMAT];
    int nper;
};

// dW tile [32 WM rows of N] x [32 WN cols of K] per workgroup for one slab of tokens; FM = 32 WM, FN = 32 WN in {64, 128}
// G = 2: a workgroup is TWO groups of four waves, each with its own token slab and its own pair of stage buffers; their accumulators meet
// in LDS before the atomics.  Same waves, loads in flight and LDS per CU as two workgroups of the one-group form, HALF the atomics: the
// fp32 atomics of a launch (2 workgroups per CU x 256 CUs x 16 K floats = 8.4 M, ~2.7 ps each whatever the shape) were 23 us of the 33 - 58 us
// that the config-4 shapes below 20 GFLOP take (tools/variants.sh, BF_ABL = 8: sum over the step's 36 shapes 2517 -> 1839 us without them).
template <int WM, int WN, int G>
__global__ __launch_bounds__(256 * G) void wgrad_bf16_kernel(const uint16_t* __restrict__ dy, int ldy,
                                                             const uint16_t* __restrict__ x, int ldx, int T, int N, int K,
                                                             WgradOut out, int nsplit) {
    constexpr int FM = 32 * WM, FN = 32 * WN, NT = 256 * G;
    constexpr int A_BYTES = BK * FM * 2, B_BYTES = BK * FN * 2;
    constexpr int STAGE = A_BYTES + B_BYTES;
    constexpr int NA = BK * (FM / 8) / 256, NB = BK * (FN / 8) / 256;       // chunks per thread per stage (2 or 4)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
    const int grp = G == 1 ? 0 : (int)(threadIdx.x >> 8);                   // wave-uniform
    unsigned char* const smem = smem_all + grp * 2 * STAGE;
    const int t = threadIdx.x & 255, lane = t & 63, w = t >> 6;
    const int i16 = lane & 15, g = lane >> 4;
    const int wm = w >> 1, wn = w & 1;
    const int tiles_n = K / FN;
    int bid = blockIdx.x;
    const int split = bid % nsplit; bid /= nsplit;
    const int tn = bid % tiles_n, tm = bid / tiles_n;
    const int n0 = tm * FM, k0 = tn * FN;
    const int mat = n0 / out.nper, nloc = n0 - mat * out.nper;
    float* __restrict__ const dw = out.dw[mat];
    float* __restrict__ const db = out.db[mat];
    const int nst = T / BK, nslab = nsplit * G, slab = split * G + grp;
    const int st0 = (int)((long long)nst * slab / nslab), st1 = (int)((long long)nst * (slab + 1) / nslab);
    int nit = st1 - st0;                                                    // the groups share the barriers: common trip count
    if (G == 2) {
        const int o = split * G + (grp ^ 1);
        const int no = (int)((long long)nst * (o + 1) / nslab) - (int)((long long)nst * o / nslab);
        nit = nit > no ? nit : no;
    }

    f32x4 acc[WM][WN];
#pragma unroll
    for (int a = 0; a < WM; ++a)
#pragma unroll
        for (int b = 0; b < WN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x4 ra[NA], rb[NB];
    float dbacc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};     // column sums of this thread's 8 dy columns (chunk t % (FM/8))
    const bool do_db = (db != nullptr) && (tn == 0);

    auto gload = [&](int st) {
        const size_t tok0 = (size_t)st * BK;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int e = t + 256 * i;
            ra[i] = *reinterpret_cast<const u32x4*>(dy + (tok0 + e / (FM / 8)) * ldy + n0 + 8 * (e % (FM / 8)));
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int e = t + 256 * i;
            rb[i] = *reinterpret_cast<const u32x4*>(x + (tok0 + e / (FN / 8)) * ldx + k0 + 8 * (e % (FN / 8)));
        }
    };
    auto swrite = [&](int buf) {
        unsigned char* As = smem + buf * STAGE;
        unsigned char* Bs = As + A_BYTES;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int e = t + 256 * i;
            *reinterpret_cast<u32x4*>(As + off_tr<FM>(e / (FM / 8), e % (FM / 8))) = ra[i];
            if (do_db) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    dbacc[2 * c] += __uint_as_float(ra[i][c] << 16);
                    dbacc[2 * c + 1] += __uint_as_float(ra[i][c] & 0xffff0000u);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int e = t + 256 * i;
            *reinterpret_cast<u32x4*>(Bs + off_tr<FN>(e / (FN / 8), e % (FN / 8))) = rb[i];
        }
    };

    if (st0 < st1) {
        gload(st0);
        swrite(0);
    }
    __syncthreads();
    for (int it = 0; it < nit; ++it) {
        const int st = st0 + it;
        const int buf = it & 1;
        const bool have = st < st1, more = st + 1 < st1;
        if (more && !(BF_WABL & 2)) gload(st + 1);
        const unsigned char* As = smem + buf * STAGE;
        const unsigned char* Bs = As + A_BYTES;
        if (have) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                s16x8 af[WM], bf[WN];
#pragma unroll
                for (int a = 0; a < WM; ++a) af[a] = tr_frag<FM>(As, 32 * s + 8 * g, wm * WM + a, lane);
#pragma unroll
                for (int b = 0; b < WN; ++b) bf[b] = tr_frag<FN>(Bs, 32 * s + 8 * g, wn * WN + b, lane);
#pragma unroll
                for (int a = 0; a < WM; ++a)
#pragma unroll
                    for (int b = 0; b < WN; ++b) { if (!(BF_WABL & 1)) acc[a][b] = mfma_bf16(af[a], bf[b], acc[a][b]); }
            }
        }
        if (more) swrite(buf ^ 1);
        __syncthreads();
    }
    // ---- epilogue: tile -> LDS (row-major FM x FN fp32; G = 2: the second group's tile first, the first group adds its own) ->
    // full-line fp32 atomics
    float* Cs = reinterpret_cast<float*>(smem_all);
    if (grp == G - 1) {
#pragma unroll
        for (int a = 0; a < WM; ++a)
#pragma unroll
            for (int b = 0; b < WN; ++b)
#pragma unroll
                for (int j = 0; j < 4; ++j) Cs[(wm * WM * 16 + a * 16 + 4 * g + j) * FN + wn * WN * 16 + b * 16 + i16] = acc[a][b][j];
    }
    __syncthreads();
    if (G == 2) {
        if (grp == 0) {
#pragma unroll
            for (int a = 0; a < WM; ++a)
#pragma unroll
                for (int b = 0; b < WN; ++b)
#pragma unroll
                    for (int j = 0; j < 4; ++j) Cs[(wm * WM * 16 + a * 16 + 4 * g + j) * FN + wn * WN * 16 + b * 16 + i16] += acc[a][b][j];
        }
        __syncthreads();
    }
    // (each of a tile's nsplit workgroups starts its walk over the FM x FN addresses at another row: they finish together, and
    // queueing on the same cache lines was a quarter of the kernel)
    if (!(BF_ABL & 8)) {
        const int rot = (int)(((unsigned)split * 2654435761u) % (unsigned)FM) * FN;
        for (int i = threadIdx.x; i < FM * FN; i += NT) {
            int e = i + rot;
            if (e >= FM * FN) e -= FM * FN;
            atomicAdd(dw + (size_t)(nloc + e / FN) * K + k0 + e % FN, Cs[e]);
        }
    }
    if (do_db) {
        // thread t always staged chunk (t % (FM/8)) of rows (t / (FM/8)) + k * 256 / (FM/8): fold the row groups of all threads
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem_all);      // [NT / (FM/8)][FM]
        constexpr int CPR = FM / 8;
#pragma unroll
        for (int c = 0; c < 8; ++c) red[((grp * 256 + t) / CPR) * FM + 8 * (t % CPR) + c] = dbacc[c];
        __syncthreads();
        if (threadIdx.x < FM) {
            float tot = 0.f;
            for (int r = 0; r < NT / CPR; ++r) tot += red[r * FM + threadIdx.x];
            atomicAdd(db + nloc + threadIdx.x, tot);
        }
    }
}

template <int WM, int WN, int G>
void launch_wgrad_g(const uint16_t* dy, int ldy, const uint16_t* x, int ldx, int T, int N, int K, const WgradOut& out, int tiles, int nwg,
                    hipStream_t s) {
    constexpr int FM = 32 * WM, FN = 32 * WN;
    constexpr size_t stage = (size_t)BK * (FM + FN) * 2;
    constexpr size_t smem = 2 * G * stage > (size_t)FM * FN * 4 ? 2 * G * stage : (size_t)FM * FN * 4;
    if (smem > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_bf16_kernel<WM, WN, G>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)smem);
    hipLaunchKernelGGL((wgrad_bf16_kernel<WM, WN, G>), dim3(tiles * nwg), dim3(256 * G), smem, s, dy, ldy, x, ldx, T, N, K, out, nwg);
}

template <int WM, int WN>
void launch_wgrad(const uint16_t* dy, int ldy, const uint16_t* x, int ldx, int T, int N, int K, const WgradOut& out, hipStream_t s) {
    constexpr int FM = 32 * WM, FN = 32 * WN;
    const int tiles = (N / FM) * (K / FN);
    int nsplit = BF_WSPLIT * dhz_num_cus() / tiles;                      // token slabs per tile
    const int max_split = T / (BK * 4) > 0 ? T / (BK * 4) : 1;           // at least 4 stages per slab
    if (nsplit > max_split) nsplit = max_split;
    if (nsplit < 1) nsplit = 1;
    // two slabs per workgroup where the slabs are short (the atomics are then a third of the launch); long slabs keep independent
    // workgroups (the two groups of a pair share every barrier: T = 8192, N = 3072, K = 1024 measured 86 -> 104 us paired)
    if (BF_WPAIR && nsplit >= 2 && (nsplit % 2 == 0 || nsplit >= 16) && T / BK / nsplit <= BF_WPAIR_MAX_STAGES) launch_wgrad_g<WM, WN, 2>(dy, ldy, x, ldx, T, N, K, out, tiles, nsplit / 2, s);
    else launch_wgrad_g<WM, WN, 1>(dy, ldy, x, ldx, T, N, K, out, tiles, nsplit, s);
}

int dispatch_wgrad(const char* who, const uint16_t* dy, int ldy, const uint16_t* x, int ldx, int T, int nmat, int nper, int K,
                   const WgradOut& out, hipStream_t s) {
    const int N = nmat * nper;
    DHZ_REQUIRE(T > 0 && T % BK == 0, "%s: T=%d must be a multiple of %d", who, T, BK);
    DHZ_REQUIRE(nper % 64 == 0 && K % 64 == 0 && nper > 0 && K > 0, "%s: N=%d K=%d must be multiples of 64", who, nper, K);
    DHZ_REQUIRE(ldy % 8 == 0 && ldx % 8 == 0 && ldy >= N && ldx >= K, "%s: bad leading dims", who);
    DHZ_REQUIRE((((uintptr_t)dy | (uintptr_t)x) & 15) == 0, "%s: operands must be 16-byte aligned", who);
    const int wm = nper % 128 == 0 ? 4 : 2, wn = K % 128 == 0 ? 4 : 2;
#define CASE(a, b) \
    if (wm == a && wn == b) launch_wgrad<a, b>(dy, ldy, x, ldx, T, N, K, out, s);
    CASE(4, 4) CASE(4, 2) CASE(2, 4) CASE(2, 2)
#undef CASE
    DHZ_CHECK_LAUNCH(who);
    return DHZ_OK;
}

}  // namespace

extern "C" int dhz_linear_fwd_bf16(const void* x, int ldx, const void* w, const float* bias, void* y, int ldy, int T, int N, int K,
                                   void* stream) {
    return dispatch_gemm<false>("dhz_linear_fwd_bf16", (const uint16_t*)x, ldx, (const uint16_t*)w, K, bias, (uint16_t*)y, ldy, T, N, K,
                                (hipStream_t)stream);
}

extern "C" int dhz_linear_fwd_bf16_res(const void* x, int ldx, const void* w, const float* bias, const void* res, const float* scale, void* out,
                                       int ldo, int T, int N, int K, int tokens_per_image, int Hres, int Wres, int shift, int windowed,
                                       void* stream) {
    const TokEpi epi{res, scale, tokens_per_image, Hres, Wres, shift, windowed};
    return dispatch_gemm<false>("dhz_linear_fwd_bf16_res", (const uint16_t*)x, ldx, (const uint16_t*)w, K, bias, (uint16_t*)out, ldo, T, N, K,
                                (hipStream_t)stream, &epi);
}

extern "C" int dhz_linear_dgrad_bf16(const void* dy, int ldy, const void* w, void* dx, int ldx, int T, int N, int K, void* stream) {
    // dx[T,K] = dy[T,N] w[N,K]: contraction over the N rows of w (transpose reads), output features = its K columns
    return dispatch_gemm<true>("dhz_linear_dgrad_bf16", (const uint16_t*)dy, ldy, (const uint16_t*)w, K, nullptr, (uint16_t*)dx, ldx, T, K,
                               N, (hipStream_t)stream);
}

extern "C" int dhz_linear_wgrad_bf16(const void* dy, int ldy, const void* x, int ldx, int T, int nmat, int nper, int K,
                                     float* const* dw, float* const* db, void* stream) {
    DHZ_REQUIRE(dy && x && dw, "dhz_linear_wgrad_bf16: null pointer");
    DHZ_REQUIRE(nmat >= 1 && nmat <= MAXMAT, "dhz_linear_wgrad_bf16: nmat=%d must be 1..%d", nmat, MAXMAT);
    WgradOut out = {};
    for (int i = 0; i < nmat; ++i) {
        DHZ_REQUIRE(dw[i], "dhz_linear_wgrad_bf16: null dw[%d]", i);
        out.dw[i] = dw[i];
        out.db[i] = db ? db[i] : nullptr;
    }
    out.nper = nper;
    return dispatch_wgrad("dhz_linear_wgrad_bf16", (const uint16_t*)dy, ldy, (const uint16_t*)x, ldx, T, nmat, nper, K, out,
                          (hipStream_t)stream);
}
