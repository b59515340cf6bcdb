// EXPERIMENT (off by default, never the headline): the fp32 token-Linear GEMMs with a contraction of 128 or more on the bf16
// matrix pipe by operand splitting.  Every fp32 operand is cut, on its way into LDS, into two bf16 pieces
//     x = hi + lo + r,   hi = bf16(x),  lo = bf16(x - hi),   |r| <= 2^-17 |x|
// and a product is taken as  a.b ~ a_hi b_hi + a_hi b_lo + a_lo b_hi  (three v_mfma_f32_16x16x32_bf16 with fp32 accumulation;
// the dropped a_lo b_lo and the r terms are <= 2^-16 relative per product).  The result is NOT fp32 arithmetic - about 16
// mantissa bits per product instead of 24 - which is why the path sits behind an explicit switch (dehaze_hip.ops.SPLIT_BF16,
// bench.py --split-bf16), is reported as its own bench object with its dtype stated, and BASELINE configs[1] stays on the fp32 pipe
// (csrc/linear_gemm.hip).  What it buys: the bf16 pipe has 16 x the rate of v_mfma_f32_16x16x4_f32, so three passes plus the
// split arithmetic cost about a third of the fp32 matrix time and these GEMMs fall back to their HBM time.
//     forward        y[T,N]  = x[T,K]  . W[N,K]^T + b      both operands contraction-contiguous
//     backward-data  dx[T,K] = dy[T,N] . W[N,K]            W's rows are the contraction: transpose reads of the bf16 images
// SIX = true is the 6-term form: three pieces per operand (hi + mid + lo = all 24 mantissa bits, exactly) and the six products
// down to 2^-16 (hh, hm, mh, hl, lh, mm); the dropped ml, lm, ll terms are <= 2^-24 relative - the size of ONE fp32 rounding - so
// the result is in the error class of an fp32 GEMM (not bit-identical: other grouping of the sums) at 6 / 16 of the fp32 pipe's
// matrix time.
// LDS images and fragment reads are those of csrc/linear_bf16.hip (one hi and one lo image per operand); operands and results stay
// fp32 in HBM.
#include <stdlib.h>
#include "common.h"
#include "tok_epilogue.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 64;

__device__ __forceinline__ f32x4 mfma_bf16(s16x8 a, s16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ int off_row(int row, int ch) { return row * 128 + 16 * (ch ^ ((row >> 1) & 7)); }
template <int F>
__device__ __forceinline__ int off_tr(int row, int ch) {
    if (F >= 128) return row * (2 * F) + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3)));   // 256 / 512-byte rows: XOR on the low four chunk bits
    return row * 128 + 16 * (ch ^ ((row & 2) | ((row & 8) >> 1)));
}
template <int F>
__device__ __forceinline__ s16x8 tr_frag(const unsigned char* img, int r0, int cb, int lane) {
    const int m = lane & 15, q = m >> 2, p = m & 3;
    typedef s16x4 __attribute__((address_space(3))) * lds_ptr;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(img + off_tr<F>(r0 + q, 2 * cb + (p >> 1)) + 8 * (p & 1)));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(img + off_tr<F>(r0 + 4 + q, 2 * cb + (p >> 1)) + 8 * (p & 1)));
    return s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// eight fp32 values -> their bf16 heads and the bf16 of what the heads leave
__device__ __forceinline__ void split8(const f32x4 a, const f32x4 b, u32x4& hi, u32x4& lo) {
    const float x[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    uint32_t h[8], l[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        h[i] = f32_to_bf16(x[i]);
        l[i] = f32_to_bf16(x[i] - __uint_as_float(h[i] << 16));
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        hi[i] = h[2 * i] | (h[2 * i + 1] << 16);
        lo[i] = l[2 * i] | (l[2 * i + 1] << 16);
    }
}

// ... and into three pieces by TRUNCATION: a bf16 is the top 16 bits of the fp32 pattern, so each piece takes the next eight
// significant bits of what is left (x - hi and r - mid are exact) and three pieces hold all 24.  Per pair of elements: two masks
// and one packed subtraction per level, one v_perm_b32 per piece to pack the two top halves - 4.5 VALU instructions per element.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_top(float x1, float x0) {              // (x1 & 0xffff0000) | (x0 >> 16)
    return __builtin_amdgcn_perm(__float_as_uint(x1), __float_as_uint(x0), 0x07060302u);
}
__device__ __forceinline__ f32x2 top16(f32x2 v) {
    return f32x2{__uint_as_float(__float_as_uint(v[0]) & 0xffff0000u), __uint_as_float(__float_as_uint(v[1]) & 0xffff0000u)};
}
__device__ __forceinline__ void split8x3(const f32x4 a, const f32x4 b, u32x4& hi, u32x4& mid, u32x4& lo) {
    const f32x2 x[4] = {{a[0], a[1]}, {a[2], a[3]}, {b[0], b[1]}, {b[2], b[3]}};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x2 r1 = x[i] - top16(x[i]);
        const f32x2 r2 = r1 - top16(r1);
        hi[i] = pack_top(x[i][1], x[i][0]);
        mid[i] = pack_top(r1[1], r1[0]);
        lo[i] = pack_top(r2[1], r2[0]);
    }
}

// The same split with SCALAR-lane subtractions (11 instructions per pair instead of 9), for loops that multiply on the bf16 matrix pipe
// while they split: a packed fp32 instruction between two MFMAs costs 7 - 17 cycles of matrix-pipe time, the first two plain vector
// instructions behind a bf16 MFMA cost nothing and further ones 4 cycles each (tools/ubench/interleave.hip).  Inline asm: left as C,
// hipcc's SLP vectoriser re-packs the subtractions.
__device__ __forceinline__ float sub_np(float a, float b) {
    float r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float mul_np(float a, float b) {
    float r;
    asm("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ void split8x3_np(const f32x4 a, const f32x4 b, u32x4& hi, u32x4& mid, u32x4& lo) {
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

// C[M,NF] = A[M,KC] . op(B) (+ bias).  BTR = false: B is [NF][KC] (forward); true: B is [KC][NF] (backward-data).
template <int WM, int WN, bool BTR, bool SIX>
__global__ __launch_bounds__(256, 2) void gemm_split_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B,
                                                            int ldb, const float* __restrict__ bias, float* __restrict__ C, int ldc,
                                                            int M, int NF, int KC, int tiles_n, int ntiles, const TokEpi epi, int epi_on) {
    constexpr int BM = 32 * WM, BN = 32 * WN;
    constexpr int A_BYTES = BM * 128;
    constexpr int B_BYTES = BTR ? BK * BN * 2 : BN * 128;
    constexpr int NA = WM, NB = WN;                              // 8-element chunks per thread per stage
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NP = SIX ? 3 : 2;                              // pieces (LDS images) per operand; piece NP - 1 is the smallest
    unsigned char* const Ah = smem;                              // A images: hi, (mid,) lo
    unsigned char* const Bh = smem + NP * A_BYTES;               // B images
    unsigned char* const Al = Ah + (NP - 1) * A_BYTES;
    unsigned char* const Bl = Bh + (NP - 1) * B_BYTES;
    unsigned char* const Am = Ah + A_BYTES;                      // (SIX only)
    unsigned char* const Bm = Bh + B_BYTES;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int i16 = lane & 15, g = lane >> 4;
    const int wm = w >> 1, wn = w & 1;
    const int nst = KC / BK;
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

    f32x4 ra[NA][2], rb[NB][2];
    const float* pa[NA];
    const float* pb[NB];
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
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            ra[i][0] = *reinterpret_cast<const f32x4*>(pa[i] + k0);
            ra[i][1] = *reinterpret_cast<const f32x4*>(pa[i] + k0 + 4);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const float* p = pb[i] + (BTR ? (size_t)k0 * ldb : (size_t)k0);
            rb[i][0] = *reinterpret_cast<const f32x4*>(p);
            rb[i][1] = *reinterpret_cast<const f32x4*>(p + 4);
        }
    };
    auto swrite = [&]() {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int e = t + 256 * i;
            u32x4 hi, mid, lo;
            if (SIX) split8x3(ra[i][0], ra[i][1], hi, mid, lo);
            else split8(ra[i][0], ra[i][1], hi, lo);
            const int o = off_row(e >> 3, e & 7);
            *reinterpret_cast<u32x4*>(Ah + o) = hi;
            if (SIX) *reinterpret_cast<u32x4*>(Am + o) = mid;
            *reinterpret_cast<u32x4*>(Al + o) = lo;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int e = t + 256 * i;
            u32x4 hi, mid, lo;
            if (SIX) split8x3(rb[i][0], rb[i][1], hi, mid, lo);
            else split8(rb[i][0], rb[i][1], hi, lo);
            const int o = BTR ? off_tr<BN>(e / (BN / 8), e % (BN / 8)) : off_row(e >> 3, e & 7);
            *reinterpret_cast<u32x4*>(Bh + o) = hi;
            if (SIX) *reinterpret_cast<u32x4*>(Bm + o) = mid;
            *reinterpret_cast<u32x4*>(Bl + o) = lo;
        }
    };

    int ti = 0, tile = tile_of(0);
    if (tile < 0) return;
    set_tile(tile);
    gload(0);
    swrite();
    __syncthreads();
    const int sw = (i16 >> 1) & 7;
    while (true) {
        const int ntile = tile_of(ti + 1);
        const int tn = tile % tiles_n, tm = tile / tiles_n;
        for (int st = 0; st < nst; ++st) {
            const bool last = st + 1 == nst;
            const bool more = !last || ntile >= 0;
            if (!last) gload((st + 1) * BK);
            else if (ntile >= 0) { set_tile(ntile); gload(0); }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                s16x8 ah[WM], am[SIX ? WM : 1], al[WM], bh[WN], bm[SIX ? WN : 1], bl[WN];
#pragma unroll
                for (int a = 0; a < WM; ++a) {
                    const int o = (wm * WM * 16 + a * 16 + i16) * 128 + 16 * ((4 * s + g) ^ sw);
                    ah[a] = *reinterpret_cast<const s16x8*>(Ah + o);
                    if (SIX) am[a] = *reinterpret_cast<const s16x8*>(Am + o);
                    al[a] = *reinterpret_cast<const s16x8*>(Al + o);
                }
#pragma unroll
                for (int b = 0; b < WN; ++b) {
                    if (BTR) {
                        bh[b] = tr_frag<BN>(Bh, 32 * s + 8 * g, wn * WN + b, lane);
                        if (SIX) bm[b] = tr_frag<BN>(Bm, 32 * s + 8 * g, wn * WN + b, lane);
                        bl[b] = tr_frag<BN>(Bl, 32 * s + 8 * g, wn * WN + b, lane);
                    } else {
                        const int o = ((wn * WN + b) * 16 + i16) * 128 + 16 * ((4 * s + g) ^ sw);
                        bh[b] = *reinterpret_cast<const s16x8*>(Bh + o);
                        if (SIX) bm[b] = *reinterpret_cast<const s16x8*>(Bm + o);
                        bl[b] = *reinterpret_cast<const s16x8*>(Bl + o);
                    }
                }
#pragma unroll
                for (int a = 0; a < WM; ++a)
#pragma unroll
                    for (int b = 0; b < WN; ++b) {                           // D = C^T block (epilogue); small terms first
                        acc[a][b] = mfma_bf16(bl[b], ah[a], acc[a][b]);
                        acc[a][b] = mfma_bf16(bh[b], al[a], acc[a][b]);
                        if (SIX) {
                            acc[a][b] = mfma_bf16(bm[b], am[a], acc[a][b]);
                            acc[a][b] = mfma_bf16(bm[b], ah[a], acc[a][b]);
                            acc[a][b] = mfma_bf16(bh[b], am[a], acc[a][b]);
                        }
                        acc[a][b] = mfma_bf16(bh[b], ah[a], acc[a][b]);
                    }
            }
            if (more) {
                __syncthreads();                                         // every wave is done with this stage's images
                swrite();
                __syncthreads();
            }
        }
        {   // acc[a][b][j] = C[token 16 a + i16][feature 16 b + 4 g + j]: one 16-byte store per block and lane
            const int m0 = tm * BM + wm * WM * 16 + i16, n0 = tn * BN + wn * WN * 16 + 4 * g;
            float* c0 = C + (size_t)m0 * ldc + n0;
            if (epi_on) {                                                    // the block's residual step (csrc/tok_epilogue.h)
                const int mbw = __builtin_amdgcn_readfirstlane(tm * BM + wm * WM * 16);
                int dst[WM];
                float sc;
                tok_epi_rows<WM>(epi, mbw < M ? mbw : 0, i16, dst, sc);
#pragma unroll
                for (int a = 0; a < WM; ++a) {
                    if (m0 + 16 * a < M) {
#pragma unroll
                        for (int b = 0; b < WN; ++b) {
                            f32x4 v = acc[a][b];
                            if (bias) v += *reinterpret_cast<const f32x4*>(bias + n0 + 16 * b);
                            v *= sc;
                            if (epi.res) v += *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(epi.res) + (size_t)dst[a] * ldc + n0 + 16 * b);
                            *reinterpret_cast<f32x4*>(C + (size_t)dst[a] * ldc + n0 + 16 * b) = v;
                        }
                    }
                }
            } else
#pragma unroll
            for (int a = 0; a < WM; ++a) {
                if (m0 + 16 * a < M) {
#pragma unroll
                    for (int b = 0; b < WN; ++b) {
                        f32x4 v = acc[a][b];
                        if (bias) v += *reinterpret_cast<const f32x4*>(bias + n0 + 16 * b);
                        *reinterpret_cast<f32x4*>(c0 + (size_t)(16 * a) * ldc + 16 * b) = v;
                    }
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

template <int WM, int WN, bool BTR, bool SIX>
void launch_split(const float* A, int lda, const float* B, int ldb, const float* bias, float* C, int ldc, int M, int NF, int KC,
                  const TokEpi& epi, int epi_on, hipStream_t s) {
    constexpr int BM = 32 * WM, BN = 32 * WN;
    constexpr size_t smem = (SIX ? 3 : 2) * ((size_t)(BM * 128) + (size_t)(BTR ? BK * BN * 2 : BN * 128));
    const int tiles_n = NF / BN, tiles_m = (M + BM - 1) / BM;
    const int ntiles = tiles_n * tiles_m;
    const int slots = 2 * dhz_num_cus();
    const int grid = ntiles < slots ? ntiles : slots;
    if (smem > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_split_kernel<WM, WN, BTR, SIX>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL((gemm_split_kernel<WM, WN, BTR, SIX>), dim3(grid), dim3(256), smem, s, A, lda, B, ldb, bias, C, ldc, M, NF, KC,
                       tiles_n, ntiles, epi, epi_on);
}

template <bool BTR>
int dispatch_split(const char* who, const float* A, int lda, const float* B, int ldb, const float* bias, float* C, int ldc, int M,
                   int NF, int KC, int terms, hipStream_t s, const TokEpi& epi = TokEpi{}, int epi_on = 0) {
    DHZ_REQUIRE(terms == 3 || terms == 6, "%s: terms=%d (3 or 6)", who, terms);
    DHZ_REQUIRE(A && B && C, "%s: null pointer", who);
    DHZ_REQUIRE(M > 0 && NF > 0 && KC > 0 && NF % 64 == 0 && KC % 64 == 0, "%s: T=%d features=%d contraction=%d (multiples of 64)", who,
                M, NF, KC);
    DHZ_REQUIRE(lda % 4 == 0 && ldb % 4 == 0 && ldc % 4 == 0 && ldc >= NF && lda >= KC, "%s: bad leading dimensions", who);
    DHZ_REQUIRE((((uintptr_t)A | (uintptr_t)B | (uintptr_t)C | (uintptr_t)bias) & 15) == 0, "%s: operands must be 16-byte aligned", who);
    // six-term form: three LDS images per operand - 64-column tiles keep two workgroups per CU (72 KB)
    if (epi_on) {
        const char* bad = tok_epi_check(epi, M);
        DHZ_REQUIRE(!bad, "%s: %s", who, bad);
        DHZ_REQUIRE(((uintptr_t)epi.res & 15) == 0, "%s: the shortcut must be 16-byte aligned", who);
    }
    const int wn = (NF % 128 == 0 && terms == 3) ? 4 : 2;      // (128-column tiles at 96 KB / one workgroup per CU: slower, measured)
    const long blocks128 = (long)((M + 127) / 128) * (NF / (32 * wn));
    const int wm = blocks128 >= dhz_num_cus() ? 4 : 2;
#define CASE(a, b) \
    if (wm == a && wn == b) {                                                                              \
        if (terms == 3) launch_split<a, b, BTR, false>(A, lda, B, ldb, bias, C, ldc, M, NF, KC, epi, epi_on, s);        \
        else launch_split<a, b, BTR, true>(A, lda, B, ldb, bias, C, ldc, M, NF, KC, epi, epi_on, s);                    \
    }
    CASE(4, 4) CASE(4, 2) CASE(2, 4) CASE(2, 2)
#undef CASE
    DHZ_CHECK_LAUNCH(who);
    return DHZ_OK;
}

// ------------------------------------------------------------------------------------------------ weight gradient
//     dW[N,K] += dy^T[N,T] . x[T,K],  db[N] += column sums of dy        (contraction over token rows: both operands by transpose
// reads of their hi / lo images; split over T, fp32 atomics; db from the fp32 values, i.e. exact).  Row t of dy may carry a factor
// row_scale[t / rows_per_scale] (the per-image DropPath scale of dhz_linear_wgrad_rs), applied before the split.
constexpr int MAXMAT = 4;
struct WgradOut {
    float* dw[MAXMAT];
    float* db[MAXMAT];
    int nper;
};

template <int WM, int WN, bool SIX>
__global__ __launch_bounds__(256, 2) void wgrad_split_kernel(const float* __restrict__ dy, int ldy, const float* __restrict__ x,
                                                             int ldx, int T, int N, int K, WgradOut out, int nsplit,
                                                             const float* __restrict__ row_scale, int rows_per_scale) {
    constexpr int FM = 32 * WM, FN = 32 * WN;
    constexpr int A_BYTES = BK * FM * 2, B_BYTES = BK * FN * 2;
    constexpr int NA = BK * (FM / 8) / 256, NB = BK * (FN / 8) / 256;       // 8-element chunks per thread per stage (2 or 4)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NP = SIX ? 3 : 2;
    unsigned char* const Ah = smem;
    unsigned char* const Bh = smem + NP * A_BYTES;
    unsigned char* const Al = Ah + (NP - 1) * A_BYTES;
    unsigned char* const Bl = Bh + (NP - 1) * B_BYTES;
    unsigned char* const Am = Ah + A_BYTES;                      // (SIX only)
    unsigned char* const Bm = Bh + B_BYTES;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
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
    const int nst = T / BK;
    const int st0 = (int)((long long)nst * split / nsplit), st1 = (int)((long long)nst * (split + 1) / nsplit);

    f32x4 acc[WM][WN];
#pragma unroll
    for (int a = 0; a < WM; ++a)
#pragma unroll
        for (int b = 0; b < WN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 ra[NA][2], rb[NB][2];
    float rs[NA];
    float dbacc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};     // column sums of this thread's 8 dy columns (chunk t % (FM/8))
    const bool do_db = (db != nullptr) && (tn == 0);

    auto gload = [&](int st) {
        const size_t tok0 = (size_t)st * BK;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int e = t + 256 * i;
            const size_t tok = tok0 + e / (FM / 8);
            const float* p = dy + tok * ldy + n0 + 8 * (e % (FM / 8));
            ra[i][0] = *reinterpret_cast<const f32x4*>(p);
            ra[i][1] = *reinterpret_cast<const f32x4*>(p + 4);
            rs[i] = row_scale ? row_scale[tok / rows_per_scale] : 1.f;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int e = t + 256 * i;
            const float* p = x + (tok0 + e / (FN / 8)) * ldx + k0 + 8 * (e % (FN / 8));
            rb[i][0] = *reinterpret_cast<const f32x4*>(p);
            rb[i][1] = *reinterpret_cast<const f32x4*>(p + 4);
        }
    };
    auto swrite = [&]() {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int e = t + 256 * i;
            const f32x4 v0 = ra[i][0] * rs[i], v1 = ra[i][1] * rs[i];
            u32x4 hi, mid, lo;
            if (SIX) split8x3(v0, v1, hi, mid, lo);
            else split8(v0, v1, hi, lo);
            const int o = off_tr<FM>(e / (FM / 8), e % (FM / 8));
            *reinterpret_cast<u32x4*>(Ah + o) = hi;
            if (SIX) *reinterpret_cast<u32x4*>(Am + o) = mid;
            *reinterpret_cast<u32x4*>(Al + o) = lo;
            if (do_db) {
#pragma unroll
                for (int c = 0; c < 4; ++c) { dbacc[c] += v0[c]; dbacc[4 + c] += v1[c]; }
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int e = t + 256 * i;
            u32x4 hi, mid, lo;
            if (SIX) split8x3(rb[i][0], rb[i][1], hi, mid, lo);
            else split8(rb[i][0], rb[i][1], hi, lo);
            const int o = off_tr<FN>(e / (FN / 8), e % (FN / 8));
            *reinterpret_cast<u32x4*>(Bh + o) = hi;
            if (SIX) *reinterpret_cast<u32x4*>(Bm + o) = mid;
            *reinterpret_cast<u32x4*>(Bl + o) = lo;
        }
    };

    if (st0 < st1) {
        gload(st0);
        swrite();
    }
    __syncthreads();
    for (int st = st0; st < st1; ++st) {
        const bool more = st + 1 < st1;
        if (more) gload(st + 1);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            s16x8 ah[WM], am[SIX ? WM : 1], al[WM], bh[WN], bm[SIX ? WN : 1], bl[WN];
#pragma unroll
            for (int a = 0; a < WM; ++a) {
                ah[a] = tr_frag<FM>(Ah, 32 * s + 8 * g, wm * WM + a, lane);
                if (SIX) am[a] = tr_frag<FM>(Am, 32 * s + 8 * g, wm * WM + a, lane);
                al[a] = tr_frag<FM>(Al, 32 * s + 8 * g, wm * WM + a, lane);
            }
#pragma unroll
            for (int b = 0; b < WN; ++b) {
                bh[b] = tr_frag<FN>(Bh, 32 * s + 8 * g, wn * WN + b, lane);
                if (SIX) bm[b] = tr_frag<FN>(Bm, 32 * s + 8 * g, wn * WN + b, lane);
                bl[b] = tr_frag<FN>(Bl, 32 * s + 8 * g, wn * WN + b, lane);
            }
#pragma unroll
            for (int a = 0; a < WM; ++a)
#pragma unroll
                for (int b = 0; b < WN; ++b) {
                    acc[a][b] = mfma_bf16(al[a], bh[b], acc[a][b]);
                    acc[a][b] = mfma_bf16(ah[a], bl[b], acc[a][b]);
                    if (SIX) {
                        acc[a][b] = mfma_bf16(am[a], bm[b], acc[a][b]);
                        acc[a][b] = mfma_bf16(am[a], bh[b], acc[a][b]);
                        acc[a][b] = mfma_bf16(ah[a], bm[b], acc[a][b]);
                    }
                    acc[a][b] = mfma_bf16(ah[a], bh[b], acc[a][b]);
                }
        }
        __syncthreads();
        if (more) {
            swrite();
            __syncthreads();
        }
    }
    // ---- epilogue: tile -> LDS (row-major FM x FN fp32) -> full-line fp32 atomics
    float* Cs = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int a = 0; a < WM; ++a)
#pragma unroll
        for (int b = 0; b < WN; ++b)
#pragma unroll
            for (int j = 0; j < 4; ++j) Cs[(wm * WM * 16 + a * 16 + 4 * g + j) * FN + wn * WN * 16 + b * 16 + i16] = acc[a][b][j];
    __syncthreads();
    for (int e = t; e < FM * FN; e += 256) atomicAdd(dw + (size_t)(nloc + e / FN) * K + k0 + e % FN, Cs[e]);
    if (do_db) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);          // [256 / (FM/8)][FM]
        constexpr int CPR = FM / 8;
#pragma unroll
        for (int c = 0; c < 8; ++c) red[(t / CPR) * FM + 8 * (t % CPR) + c] = dbacc[c];
        __syncthreads();
        if (t < FM) {
            float tot = 0.f;
            for (int r = 0; r < 256 / CPR; ++r) tot += red[r * FM + t];
            atomicAdd(db + nloc + t, tot);
        }
    }
}

// ---- round 4: the six-term weight gradient as a software pipeline.  The kernel above puts the split of a stage between two
// barriers (nothing multiplies while 216 vector instructions and 18 LDS writes per thread run).  Here a stage is 32 token rows, the
// LDS holds two stages, and the split + LDS writes of stage s+1 (registers loaded during stage s-1) ride BETWEEN the MFMAs of
// stage s - a bf16 MFMA holds the vector issue for 8 of its 16 cycles, a wave's own vector / LDS instructions go into the other 8
// (csrc/split6_gemm.hip) - with one barrier per stage.
constexpr int BK2 = 32;
// phase ablation for timing diagnostics: a COMPILE-TIME constant of a variant build (tools/variants.sh, -DDHZ_W6_ABL=<mask>;
// outputs are then wrong).  1 no global loads, 2 no split arithmetic, 4 no LDS writes, 8 no fragment reads, 16 no MFMAs, 32 no atomics
#ifndef DHZ_W6_ABL
#define DHZ_W6_ABL 0
#endif
// G = 2 (four-wave configurations): a workgroup is TWO groups of four waves, each with its own token slab and its own pair of stage
// buffers - the waves, loads in flight and LDS of two workgroups of the one-group form - whose accumulators meet in LDS before the
// atomics: half the fp32 atomics of the launch (2 x 256 workgroups x FM x FN floats, ~2.7 ps each whatever the shape: 6 - 12 us of the
// launches below 50 us, tools/variants.sh with DHZ_W6_ABL = 32).  The groups share the one barrier per stage; the group with the
// shorter slab (by at most one stage) pays the difference with bare barriers.
template <int WM, int WN, int WAVES_M, int WAVES_N, int G = 1>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N * G, WAVES_M * WAVES_N == 4 ? 2 : 1) void wgrad_split6_kernel(const float* __restrict__ dy, int ldy, const float* __restrict__ x,
                                                              int ldx, int T, int N, int K, WgradOut out, int nsplit,
                                                              const float* __restrict__ row_scale, int rows_per_scale) {
    constexpr int NT = 64 * WAVES_M * WAVES_N;
    constexpr int FM = 16 * WM * WAVES_M, FN = 16 * WN * WAVES_N;
    constexpr int A_BYTES = BK2 * FM * 2, B_BYTES = BK2 * FN * 2;          // one piece of one stage
    constexpr int STAGE = 3 * (A_BYTES + B_BYTES);
    constexpr int NA = BK2 * (FM / 8) / NT, NB = BK2 * (FN / 8) / NT;    // 8-element chunks per thread per stage (1 or 2)
    static_assert(NA >= 1 && NB >= 1, "tile too narrow for the workgroup");
    static_assert(G == 1 || NT == 256, "two groups: four-wave configurations only");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
    const int grp = G == 1 ? 0 : (int)(threadIdx.x / NT);                   // wave-uniform
    unsigned char* const smem = smem_all + grp * 2 * STAGE;
    const int t = G == 1 ? (int)threadIdx.x : (int)(threadIdx.x % NT), lane = t & 63, w = t >> 6;
    const int i16 = lane & 15, g = lane >> 4;
    const int wm = w / WAVES_N, wn = w % WAVES_N;
    const int tiles_n = K / FN;
    int bid = blockIdx.x;
    const int split = bid % nsplit; bid /= nsplit;
    const int tn = bid % tiles_n, tm = bid / tiles_n;
    const int n0 = tm * FM, k0 = tn * FN;
    const int mat = n0 / out.nper, nloc = n0 - mat * out.nper;
    float* __restrict__ const dw = out.dw[mat];
    float* __restrict__ const db = out.db[mat];
    const int nst = T / BK2, nslab = nsplit * G, slab = split * G + grp;
    const int st0 = (int)((long long)nst * slab / nslab), st1 = (int)((long long)nst * (slab + 1) / nslab);
    int idle = 0;                                                           // stages the OTHER group of the pair runs beyond this one's
    if (G == 2) {
        const int o = split * G + (grp ^ 1);
        const int no = (int)((long long)nst * (o + 1) / nslab) - (int)((long long)nst * o / nslab);
        idle = no - (st1 - st0) > 0 ? no - (st1 - st0) : 0;
    }

    f32x4 acc[WM][WN];
#pragma unroll
    for (int a = 0; a < WM; ++a)
#pragma unroll
        for (int b = 0; b < WN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 ra[NA][2], rb[NB][2];
    // the per-image factor of dy (rows_per_scale is a multiple of the 32-row stage: one factor per stage, wave-uniform): the image index
    // walks with the stages in scalar registers - as t / rows_per_scale per chunk it was a 64-bit division, ~75 vector instructions per
    // chunk and stage between the MFMAs
    float rs = 1.f;
    int rs_img = 0, rs_left = 0;                                  // image of the next stage to load, its rows still ahead
    if (row_scale) { rs_img = (int)(((long long)st0 * BK2) / rows_per_scale); rs_left = rows_per_scale - (int)(((long long)st0 * BK2) % rows_per_scale); }
    if constexpr (DHZ_W6_ABL & 1) {
#pragma unroll
        for (int i = 0; i < NA; ++i) { ra[i][0] = ra[i][1] = f32x4{1.f, 1.f, 1.f, 1.f}; }
#pragma unroll
        for (int i = 0; i < NB; ++i) rb[i][0] = rb[i][1] = f32x4{1.f, 1.f, 1.f, 1.f};
    }
    float dbacc[NA][8];
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int c = 0; c < 8; ++c) dbacc[i][c] = 0.f;
    const bool do_db = (db != nullptr) && (tn == 0);
    int a_off[NA], b_off[NB];                                    // LDS byte offsets of this thread's chunks inside a piece image
    // global addresses as (wave-uniform stage base) + (32-bit byte offset of the lane): no 64-bit vector arithmetic per load
    unsigned pa[NA], pb[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int e = t + NT * i;
        a_off[i] = off_tr<FM>(e / (FM / 8), e % (FM / 8));
        pa[i] = (unsigned)(((e / (FM / 8)) * ldy + 8 * (e % (FM / 8))) * 4);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int e = t + NT * i;
        b_off[i] = off_tr<FN>(e / (FN / 8), e % (FN / 8));
        pb[i] = (unsigned)(((e / (FN / 8)) * ldx + 8 * (e % (FN / 8))) * 4);
    }
    const char* const dyb = reinterpret_cast<const char*>(dy + n0);
    const char* const xb = reinterpret_cast<const char*>(x + k0);
    auto gload = [&](int st) {
        if constexpr (DHZ_W6_ABL & 1) return;
        const size_t tok0 = (size_t)st * BK2;
        const char* const da = dyb + tok0 * ldy * 4;
        const char* const db_ = xb + tok0 * ldx * 4;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            ra[i][0] = *reinterpret_cast<const f32x4*>(da + pa[i]);
            ra[i][1] = *reinterpret_cast<const f32x4*>(da + pa[i] + 16);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            rb[i][0] = *reinterpret_cast<const f32x4*>(db_ + pb[i]);
            rb[i][1] = *reinterpret_cast<const f32x4*>(db_ + pb[i] + 16);
        }
        if (row_scale) {
            rs = row_scale[rs_img];
            rs_left -= BK2;
            if (rs_left <= 0) { ++rs_img; rs_left += rows_per_scale; }
        }
    };
    auto swrite = [&](int buf) {
        unsigned char* As = smem + buf * STAGE;
        unsigned char* Bs = As + 3 * A_BYTES;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            f32x4 v0 = ra[i][0], v1 = ra[i][1];
            if (row_scale) {
#pragma unroll
                for (int c = 0; c < 4; ++c) { v0[c] = mul_np(v0[c], rs); v1[c] = mul_np(v1[c], rs); }
            }
            u32x4 hi, mid, lo;
            if constexpr (DHZ_W6_ABL & 2) { hi = __builtin_bit_cast(u32x4, v0); mid = __builtin_bit_cast(u32x4, v1); lo = hi ^ mid; }
            else split8x3_np(v0, v1, hi, mid, lo);
            if constexpr (DHZ_W6_ABL & 4) { asm volatile("" :: "v"(hi), "v"(mid), "v"(lo)); }
            else {
            *reinterpret_cast<u32x4*>(As + a_off[i]) = hi;
            *reinterpret_cast<u32x4*>(As + A_BYTES + a_off[i]) = mid;
            *reinterpret_cast<u32x4*>(As + 2 * A_BYTES + a_off[i]) = lo;
            }
            if (do_db) {
#pragma unroll
                for (int c = 0; c < 4; ++c) { dbacc[i][c] += v0[c]; dbacc[i][4 + c] += v1[c]; }
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            u32x4 hi, mid, lo;
            if constexpr (DHZ_W6_ABL & 2) { hi = __builtin_bit_cast(u32x4, rb[i][0]); mid = __builtin_bit_cast(u32x4, rb[i][1]); lo = hi ^ mid; }
            else split8x3_np(rb[i][0], rb[i][1], hi, mid, lo);
            if constexpr (DHZ_W6_ABL & 4) { asm volatile("" :: "v"(hi), "v"(mid), "v"(lo)); }
            else {
            *reinterpret_cast<u32x4*>(Bs + b_off[i]) = hi;
            *reinterpret_cast<u32x4*>(Bs + B_BYTES + b_off[i]) = mid;
            *reinterpret_cast<u32x4*>(Bs + 2 * B_BYTES + b_off[i]) = lo;
            }
        }
    };
    if (st0 < st1) {
        gload(st0);
        swrite(st0 & 1);
        if (st0 + 1 < st1) gload(st0 + 1);
    }
    __syncthreads();
    for (int st = st0; st < st1; ++st) {
        const int buf = st & 1;
        const unsigned char* As = smem + buf * STAGE;
        const unsigned char* Bs = As + 3 * A_BYTES;
        s16x8 af[3][WM], bf[3][WN];
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) {
#pragma unroll
            for (int a = 0; a < WM; ++a) {
                if constexpr (DHZ_W6_ABL & 8) { af[pc][a] = s16x8{(short)lane, 1, 2, 3, 4, 5, 6, (short)st}; asm volatile("" : "+v"(af[pc][a])); }
                else af[pc][a] = tr_frag<FM>(As + pc * A_BYTES, 8 * g, wm * WM + a, lane);
            }
#pragma unroll
            for (int b = 0; b < WN; ++b) {
                if constexpr (DHZ_W6_ABL & 8) { bf[pc][b] = s16x8{(short)lane, 1, 2, 3, 4, 5, 6, (short)st}; asm volatile("" : "+v"(bf[pc][b])); }
                else bf[pc][b] = tr_frag<FN>(Bs + pc * B_BYTES, 8 * g, wn * WN + b, lane);
            }
        }
        // ---- one stream: the stage's MFMAs (term-major: consecutive ones never share an accumulator), between them the split and
        //      the LDS writes of stage st+1 (the other buffer: its readers passed the barrier at the end of stage st-1)
        constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};      // (dy piece, x piece): lh hl mm mh hm hh
#pragma unroll
        for (int term = 0; term < 6; ++term)
#pragma unroll
            for (int a = 0; a < WM; ++a)
#pragma unroll
                for (int b = 0; b < WN; ++b) {
                    if constexpr (DHZ_W6_ABL & 16) { if (term == 0) asm volatile("" : "+v"(acc[a][b]) : "v"(af[0][a]), "v"(af[1][a]), "v"(af[2][a]), "v"(bf[0][b]), "v"(bf[1][b]), "v"(bf[2][b])); }
                    else acc[a][b] = mfma_bf16(af[TA[term]][a], bf[TB[term]][b], acc[a][b]);
                }
        if (st + 1 < st1) swrite(buf ^ 1);
#pragma unroll
        for (int i = 0; i < WM * WN * 6; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x202, 3, 0);     // up to three vector instructions / LDS writes per MFMA
        }
        __builtin_amdgcn_sched_barrier(0);
        if (st + 2 < st1) gload(st + 2);
        // raw barrier: a __syncthreads() would also drain vmcnt, i.e. wait for the loads just issued
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    for (int i = 0; i < idle; ++i) __builtin_amdgcn_s_barrier();
    // ---- epilogue: tile -> LDS (row-major FM x FN fp32; G = 2: the second group's tile first, the first group adds its own) ->
    // full-line fp32 atomics
    float* Cs = reinterpret_cast<float*>(smem_all);
    if (G == 2) __syncthreads();                                            // both groups are done with their stage buffers
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
    const int tt = threadIdx.x;
    if constexpr (DHZ_W6_ABL & 32) { if (Cs[tt] == 123.456f) dw[tt] = 1.f; }
    else
    for (int e = tt; e < FM * FN; e += NT * G) atomicAdd(dw + (size_t)(nloc + e / FN) * K + k0 + e % FN, Cs[e]);
    if (do_db) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem_all);      // [rows of chunks][FM]
        constexpr int CPR = FM / 8;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int e = grp * NT * NA + t + NT * i;
#pragma unroll
            for (int c = 0; c < 8; ++c) red[(e / CPR) * FM + 8 * (e % CPR) + c] = dbacc[i][c];
        }
        __syncthreads();
        if (tt < FM) {
            float tot = 0.f;
            for (int r = 0; r < G * NT * NA / CPR; ++r) tot += red[r * FM + tt];
            atomicAdd(db + nloc + tt, tot);
        }
    }
}

#ifndef DHZ_W6_PAIR
#define DHZ_W6_PAIR 1
#endif
#ifndef DHZ_W6_PAIR_MAX_STAGES
#define DHZ_W6_PAIR_MAX_STAGES 32      // slabs longer than this keep independent workgroups: the two groups share every barrier, and at 48 - 64
                                       // stages pairing measured +3 % (T = 32768, N = 1024, K = 256: 106.9 -> 110.7 us)
#endif
template <int WM, int WN, int WAVES_M, int WAVES_N, int G>
void launch_wgrad_split6_g(const float* dy, int ldy, const float* x, int ldx, int T, int N, int K, const WgradOut& out,
                           const float* row_scale, int rows_per_scale, int tiles, int nwg, hipStream_t s) {
    constexpr int FM = 16 * WM * WAVES_M, FN = 16 * WN * WAVES_N, NT = 64 * WAVES_M * WAVES_N;
    constexpr size_t stage = (size_t)BK2 * (FM + FN) * 2 * 3;
    constexpr size_t ring = 2 * G * stage;
    constexpr size_t smem = ring > (size_t)FM * FN * 4 ? ring : (size_t)FM * FN * 4;
    if (smem > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_split6_kernel<WM, WN, WAVES_M, WAVES_N, G>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL((wgrad_split6_kernel<WM, WN, WAVES_M, WAVES_N, G>), dim3(tiles * nwg), dim3(NT * G), smem, s, dy, ldy, x, ldx, T, N, K,
                       out, nwg, row_scale, rows_per_scale);
}

template <int WM, int WN, int WAVES_M, int WAVES_N>
void launch_wgrad_split6(const float* dy, int ldy, const float* x, int ldx, int T, int N, int K, const WgradOut& out,
                         const float* row_scale, int rows_per_scale, hipStream_t s) {
    constexpr int FM = 16 * WM * WAVES_M, FN = 16 * WN * WAVES_N, NT = 64 * WAVES_M * WAVES_N;
    const int tiles = (N / FM) * (K / FN);
    int nsplit = (NT == 256 ? 2 : 1) * dhz_num_cus() / tiles;            // token slabs per tile
    const int max_split = T / (BK2 * 8) > 0 ? T / (BK2 * 8) : 1;          // at least 8 stages per slab
    if (nsplit > max_split) nsplit = max_split;
    if (nsplit < 1) nsplit = 1;
    if constexpr (NT == 256) {
        if (DHZ_W6_PAIR && nsplit >= 2 && (nsplit % 2 == 0 || nsplit >= 16) && T / BK2 / nsplit <= DHZ_W6_PAIR_MAX_STAGES) {      // (an odd count loses a slab)
            launch_wgrad_split6_g<WM, WN, WAVES_M, WAVES_N, 2>(dy, ldy, x, ldx, T, N, K, out, row_scale, rows_per_scale, tiles, nsplit / 2, s);
            return;
        }
    }
    launch_wgrad_split6_g<WM, WN, WAVES_M, WAVES_N, 1>(dy, ldy, x, ldx, T, N, K, out, row_scale, rows_per_scale, tiles, nsplit, s);
}

template <int WM, int WN, bool SIX>
void launch_wgrad_split(const float* dy, int ldy, const float* x, int ldx, int T, int N, int K, const WgradOut& out,
                        const float* row_scale, int rows_per_scale, hipStream_t s) {
    constexpr int FM = 32 * WM, FN = 32 * WN;
    constexpr size_t stage = (size_t)BK * (FM + FN) * 2 * (SIX ? 3 : 2);     // hi, (mid,) lo images of both operands
    constexpr size_t smem = stage > (size_t)FM * FN * 4 ? stage : (size_t)FM * FN * 4;
    const int tiles = (N / FM) * (K / FN);
    int nsplit = 2 * dhz_num_cus() / tiles;
    const int max_split = T / (BK * 4) > 0 ? T / (BK * 4) : 1;           // at least 4 stages per workgroup
    if (nsplit > max_split) nsplit = max_split;
    if (nsplit < 1) nsplit = 1;
    if (smem > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_split_kernel<WM, WN, SIX>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)smem);
    hipLaunchKernelGGL((wgrad_split_kernel<WM, WN, SIX>), dim3(tiles * nsplit), dim3(256), smem, s, dy, ldy, x, ldx, T, N, K, out, nsplit,
                       row_scale, rows_per_scale);
}

}  // namespace

extern "C" int dhz_linear_fwd_split(const float* x, int ldx, const float* w, const float* bias, float* y, int ldy, int T, int N, int K,
                                    int terms, void* stream) {
    return dispatch_split<false>("dhz_linear_fwd_split", x, ldx, w, K, bias, y, ldy, T, N, K, terms, (hipStream_t)stream);
}

extern "C" int dhz_linear_dgrad_split(const float* dy, int ldy, const float* w, float* dx, int ldx, int T, int N, int K, int terms,
                                      void* stream) {
    return dispatch_split<true>("dhz_linear_dgrad_split", dy, ldy, w, K, nullptr, dx, ldx, T, K, N, terms, (hipStream_t)stream);
}

extern "C" int dhz_linear_fwd_split_res(const float* x, int ldx, const float* w, const float* bias, const float* res, const float* scale,
                                        float* out, int ldo, int T, int N, int K, int tokens_per_image, int Hres, int Wres, int shift,
                                        int windowed, int terms, void* stream) {
    const TokEpi epi{res, scale, tokens_per_image, Hres, Wres, shift, windowed};
    return dispatch_split<false>("dhz_linear_fwd_split_res", x, ldx, w, K, bias, out, ldo, T, N, K, terms, (hipStream_t)stream, epi, 1);
}

extern "C" int dhz_linear_dgrad_split_scaled(const float* dy, int ldy, const float* w, const float* scale, float* dx, int ldx, int T, int N,
                                             int K, int tokens_per_image, int terms, void* stream) {
    const TokEpi epi{nullptr, scale, tokens_per_image, 0, 0, 0, 0};
    return dispatch_split<true>("dhz_linear_dgrad_split_scaled", dy, ldy, w, K, nullptr, dx, ldx, T, K, N, terms, (hipStream_t)stream, epi, 1);
}

extern "C" int dhz_linear_wgrad_split(const float* dy, int ldy, const float* x, int ldx, int T, int nmat, int nper, int K,
                                      float* const* dw, float* const* db, const float* row_scale, int rows_per_scale, int terms,
                                      void* stream) {
    const char* who = "dhz_linear_wgrad_split";
    DHZ_REQUIRE(terms == 3 || terms == 6, "%s: terms=%d (3 or 6)", who, terms);
    DHZ_REQUIRE(dy && x && dw, "%s: null pointer", who);
    DHZ_REQUIRE(nmat >= 1 && nmat <= MAXMAT, "%s: nmat=%d must be 1..%d", who, nmat, MAXMAT);
    DHZ_REQUIRE(T > 0 && T % BK == 0, "%s: T=%d must be a multiple of %d", who, T, BK);
    DHZ_REQUIRE(nper % 64 == 0 && K % 64 == 0 && nper > 0 && K > 0, "%s: N=%d K=%d must be multiples of 64", who, nper, K);
    const int N = nmat * nper;
    DHZ_REQUIRE(ldy % 4 == 0 && ldx % 4 == 0 && ldy >= N && ldx >= K, "%s: bad leading dims", who);
    DHZ_REQUIRE((((uintptr_t)dy | (uintptr_t)x) & 15) == 0, "%s: operands must be 16-byte aligned", who);
    DHZ_REQUIRE(!row_scale || (rows_per_scale > 0 && T % rows_per_scale == 0), "%s: rows_per_scale=%d must divide T", who, rows_per_scale);
    WgradOut out = {};
    for (int i = 0; i < nmat; ++i) {
        DHZ_REQUIRE(dw[i], "%s: null dw[%d]", who, i);
        out.dw[i] = dw[i];
        out.db[i] = db ? db[i] : nullptr;
    }
    out.nper = nper;
    // six-term form: three images per operand - a 128 x 64 tile keeps the stage at 72 KB (two workgroups per CU)
    const int wm = nper % 128 == 0 ? 4 : 2, wn = (K % 128 == 0 && !(terms == 6 && wm == 4)) ? 4 : 2;
    hipStream_t s = (hipStream_t)stream;
    const int rps = rows_per_scale > 0 ? rows_per_scale : 1;
    static const bool old6 = getenv("DHZ_WGRAD6_OLD") != nullptr;       // diagnostics: the round-3 kernel for the six-term form
    if (terms == 6 && !old6 && (!row_scale || rps % BK2 == 0)) {       // (a per-image factor changes between stages, never inside one)
        // pipelined kernel: 128 x 64 / 64 x 128 / 64 x 64 tiles (two piece-image stages of 32 rows: <= 72 KB, two workgroups per CU)
        const int wm6 = nper % 128 == 0 ? 4 : 2, wn6 = (K % 128 == 0 && wm6 == 2) ? 4 : 2;
        // 512-thread tiles (256 x 128 / 128 x 256 / 128 x 128, one workgroup per CU) halve the traffic through L2 and the split work
        // per product, and measured -3..-7 % on the four largest shapes of the step, +25..50 % on the small ones
        // (profiles/r04_wgrad6_ablation.txt): not dispatched; DHZ_WGRAD6_TILE=1 / 2 selects them for measurements.
        static const int big = getenv("DHZ_WGRAD6_TILE") ? atoi(getenv("DHZ_WGRAD6_TILE")) : 0;
        if (big && nper % 256 == 0 && K % 128 == 0) launch_wgrad_split6<4, 4, 4, 2>(dy, ldy, x, ldx, T, N, K, out, row_scale, rps, s);
        else if (big && nper % 128 == 0 && K % 256 == 0) launch_wgrad_split6<4, 4, 2, 4>(dy, ldy, x, ldx, T, N, K, out, row_scale, rps, s);
        else if (big == 2 && nper % 128 == 0 && K % 128 == 0) launch_wgrad_split6<2, 4, 4, 2>(dy, ldy, x, ldx, T, N, K, out, row_scale, rps, s);
        else if (wm6 == 4) launch_wgrad_split6<4, 2, 2, 2>(dy, ldy, x, ldx, T, N, K, out, row_scale, rps, s);
        else if (wn6 == 4) launch_wgrad_split6<2, 4, 2, 2>(dy, ldy, x, ldx, T, N, K, out, row_scale, rps, s);
        else launch_wgrad_split6<2, 2, 2, 2>(dy, ldy, x, ldx, T, N, K, out, row_scale, rps, s);
        DHZ_CHECK_LAUNCH(who);
        return DHZ_OK;
    }
#define CASE(a, b) \
    if (wm == a && wn == b) {                                                                                  \
        if (terms == 3) launch_wgrad_split<a, b, false>(dy, ldy, x, ldx, T, N, K, out, row_scale, rps, s);     \
        else launch_wgrad_split<a, b, true>(dy, ldy, x, ldx, T, N, K, out, row_scale, rps, s);                 \
    }
    CASE(4, 4) CASE(4, 2) CASE(2, 4) CASE(2, 2)
#undef CASE
    DHZ_CHECK_LAUNCH(who);
    return DHZ_OK;
}
