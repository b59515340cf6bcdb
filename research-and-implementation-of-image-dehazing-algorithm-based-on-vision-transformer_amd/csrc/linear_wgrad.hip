// Weight/bias gradient of y = x W^T + b for token-major activations:
//     dW[N,K] += dY^T[N,T] . X[T,K]        db[N] += sum_t dY[t,:]
// The contraction runs over T = B*H*W tokens (2k .. 524k) while N,K are 32..2048, i.e. a "skinny TN GEMM".
// Library GEMMs run these at 5-20 TFLOP/s (no split over T; 30 % of the training step in the first
// profile); for C <= 128 the op is HBM-bound (T*(N+K)*4 bytes), so the kernel is a split-T streaming
// reduction: every workgroup owns one BM x BN tile of dW for a contiguous slab of tokens, streams
// dY / X slabs through double-buffered LDS with coalesced float4 loads, feeds the fp32 matrix pipe
// (v_mfma_f32_16x16x4_f32, A = dY^T and B = X both read "row = token, lanes along the feature", which is
// conflict-free with a +16 float row pad), reduces db on the fly from the staged dY registers, and
// finally adds its partial tile into dW with full-line fp32 atomics.
#include <stdlib.h>
#include "common.h"

namespace {

constexpr int TK = 32;   // tokens per LDS stage

// WM, WN: 16x16 tiles per wave along N (rows of dW) and K (cols of dW); waves are NWM x 2 (NWM = 2: 256 threads,
// NWM = 4: 512 threads and a 256-row tile - twice the FLOP per staged byte for the compute-bound shapes)
template <int WM, int WN, int NWM>
__global__ __launch_bounds__(128 * NWM) void linear_wgrad_kernel(const float* __restrict__ dy, int ldy,
                                                           const float* __restrict__ x, int ldx, int T, int N,
                                                           int K, float* __restrict__ dw, float* __restrict__ db,
                                                           int nsplit) {
    constexpr int NTHR = 128 * NWM;
    constexpr int BM = 16 * WM * NWM, BN = 32 * WN;
    constexpr int SA = BM + 16, SB = BN + 16;          // LDS row strides (floats): stride % 32 == 16
    constexpr int A4 = BM / 4, B4 = BN / 4;            // float4 per staged row
    constexpr int NA = (TK * A4 + NTHR - 1) / NTHR, NB = (TK * B4 + NTHR - 1) / NTHR;   // float4 per thread per stage
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int STAGE = TK * (SA + SB);              // floats per stage: [A | B]
    auto As = [&](int buf) -> float* { return smem + buf * STAGE; };
    auto Bs = [&](int buf) -> float* { return smem + buf * STAGE + TK * SA; };

    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int i16 = lane & 15, g = lane >> 4;
    const int wm = w >> 1, wn = w & 1;
    const int tiles_n = K / BN;
    int bid = blockIdx.x;
    const int split = bid % nsplit; bid /= nsplit;
    const int tn = bid % tiles_n, tm = bid / tiles_n;
    const int n0 = tm * BM, k0 = tn * BN;
    // token slab of this workgroup (multiples of TK)
    const int nst = T / TK;
    const int st0 = (int)((long long)nst * split / nsplit), st1 = (int)((long long)nst * (split + 1) / nsplit);

    f32x4 acc[WM][WN];
#pragma unroll
    for (int a = 0; a < WM; ++a)
#pragma unroll
        for (int b = 0; b < WN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    // Staged rows are native vectors (f32x4), NOT HIP's float4 struct: rb[] is only ever copied (global -> register -> LDS),
    // and for a struct the front end emits that as memcpy, which left the whole array in scratch memory (private segment
    // 48-80 B for NB >= 2): every prefetched row was waited for at once and bounced through scratch, exposing the HBM
    // latency of each stage - the WN >= 2 variants ran at 2.4-3.5 TB/s / 60-80 TFLOP/s while the NB = 1 ones reached 5.3 TB/s.
    f32x4 ra[NA], rb[NB];
    float4 dbacc[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) dbacc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool do_db = (db != nullptr) && (tn == 0);

    auto gload = [&](int st) {
        const size_t tok0 = (size_t)st * TK;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int e = t + NTHR * i;
            if (TK * A4 % NTHR == 0 || e < TK * A4)
                ra[i] = *reinterpret_cast<const f32x4*>(dy + (tok0 + e / A4) * ldy + n0 + (e % A4) * 4);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int e = t + NTHR * i;
            if (TK * B4 % NTHR == 0 || e < TK * B4)
                rb[i] = *reinterpret_cast<const f32x4*>(x + (tok0 + e / B4) * ldx + k0 + (e % B4) * 4);
        }
    };
    auto swrite = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int e = t + NTHR * i;
            if (TK * A4 % NTHR == 0 || e < TK * A4) {
                *reinterpret_cast<f32x4*>(&As(buf)[(e / A4) * SA + (e % A4) * 4]) = ra[i];
                dbacc[i].x += ra[i][0]; dbacc[i].y += ra[i][1]; dbacc[i].z += ra[i][2]; dbacc[i].w += ra[i][3];
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int e = t + NTHR * i;
            if (TK * B4 % NTHR == 0 || e < TK * B4)
                *reinterpret_cast<f32x4*>(&Bs(buf)[(e / B4) * SB + (e % B4) * 4]) = rb[i];
        }
    };

    if (st0 < st1) {
        gload(st0);
        swrite(0);
    }
    __syncthreads();
    for (int st = st0; st < st1; ++st) {
        const int buf = (st - st0) & 1;
        if (st + 1 < st1) gload(st + 1);
        const float* A = As(buf) + (wm * WM * 16 + i16);
        const float* B = Bs(buf) + (wn * WN * 16 + i16);
#pragma unroll
        for (int s = 0; s < TK / 4; ++s) {
            float af[WM], bf[WN];
#pragma unroll
            for (int a = 0; a < WM; ++a) af[a] = A[(4 * s + g) * SA + 16 * a];
#pragma unroll
            for (int b = 0; b < WN; ++b) bf[b] = B[(4 * s + g) * SB + 16 * b];
#pragma unroll
            for (int a = 0; a < WM; ++a)
#pragma unroll
                for (int b = 0; b < WN; ++b) acc[a][b] = mfma16(af[a], bf[b], acc[a][b]);
        }
        if (st + 1 < st1) swrite(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: partial tile -> LDS (row-major BM x BN, stride BN) -> full-line atomics into dW
    float* Cs = smem;                                    // BM*BN floats <= 2 stages of LDS
#pragma unroll
    for (int a = 0; a < WM; ++a)
#pragma unroll
        for (int b = 0; b < WN; ++b)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                Cs[(wm * WM * 16 + a * 16 + 4 * g + j) * BN + wn * WN * 16 + b * 16 + i16] = acc[a][b][j];
    __syncthreads();
    for (int e = t; e < BM * BN; e += NTHR) {
        const int r = e / BN, c = e % BN;
        atomicAdd(dw + (size_t)(n0 + r) * K + k0 + c, Cs[e]);
    }
    if (do_db) {
        // staged element e = t + 256 i sits at (row e / A4, float4-column e % A4) of every stage: dump the
        // TK*A4 per-element sums and fold the TK rows of each column.
        __syncthreads();
        float* red = smem;                                // [TK*A4][4] floats = TK*BM <= one stage
#pragma unroll
        for (int i = 0; i < NA; ++i)
            if (TK * A4 % NTHR == 0 || t + NTHR * i < TK * A4) *reinterpret_cast<float4*>(&red[(t + NTHR * i) * 4]) = dbacc[i];
        __syncthreads();
        if (t < BM) {
            const int c4 = t / 4, comp = t % 4;
            float tot = 0.f;
            for (int r = 0; r < TK; ++r) tot += red[(r * A4 + c4) * 4 + comp];
            atomicAdd(db + n0 + t, tot);
        }
    }
}

template <int WM, int WN, int NWM = 2>
int launch(const float* dy, int ldy, const float* x, int ldx, int T, int N, int K, float* dw, float* db, hipStream_t s) {
    constexpr int BM = 16 * WM * NWM, BN = 32 * WN;
    constexpr size_t stage = (size_t)TK * (BM + 16 + BN + 16) * sizeof(float);
    constexpr size_t smem = 2 * stage > (size_t)BM * BN * 4 ? 2 * stage : (size_t)BM * BN * 4;
    const int tiles = (N / BM) * (K / BN);
    // Every workgroup ends with BM*BN fp32 atomics (chip-wide ~1.3 TB/s of added bytes): large tiles want
    // fewer, longer token slabs.  DHZ_WGRAD_TARGET overrides the workgroup target (tuning aid).
    static const int env_target = getenv("DHZ_WGRAD_TARGET") ? atoi(getenv("DHZ_WGRAD_TARGET")) : 0;
    const int target = env_target > 0 ? env_target : (NWM == 4 ? 256 : 512);   // 2 (1 for the 512-thread tile) workgroups per CU
    int nsplit = (target + tiles - 1) / tiles;
    const int max_split = T / (TK * 4) > 0 ? T / (TK * 4) : 1;     // at least 4 stages per workgroup
    if (nsplit > max_split) nsplit = max_split;
    if (nsplit < 1) nsplit = 1;
    // short token slabs pay the per-workgroup epilogue (BM*BN atomics) over too few stages: below 16 stages per workgroup
    // trade splits for stages down to one workgroup per CU (measured: 70 -> 62 us at T=32768, N=512, K=128; 73 -> 64 us at
    // T=131072, N=K=128; the long-slab shapes are untouched)
    if (env_target <= 0 && (T / TK) / nsplit < 16) {
        int alt = (256 + tiles - 1) / tiles;
        if (alt < (T / TK) / 16) alt = (T / TK) / 16;
        if (alt >= 1 && alt < nsplit) nsplit = alt;
    }
    if (smem > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_wgrad_kernel<WM, WN, NWM>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL((linear_wgrad_kernel<WM, WN, NWM>), dim3(tiles * nsplit), dim3(128 * NWM), smem, s, dy, ldy, x, ldx, T, N, K,
                       dw, db, nsplit);
    return 0;
}

}  // namespace

extern "C" int dhz_linear_wgrad(const float* dy, int ldy, const float* x, int ldx, int T, int N, int K, float* dw,
                                float* db, void* stream) {
    DHZ_REQUIRE(dy && x && dw, "dhz_linear_wgrad: null pointer");
    DHZ_REQUIRE(T > 0 && T % TK == 0, "dhz_linear_wgrad: T=%d must be a multiple of %d", T, TK);
    DHZ_REQUIRE(N % 32 == 0 && K % 32 == 0 && N > 0 && K > 0, "dhz_linear_wgrad: N=%d K=%d must be multiples of 32", N, K);
    DHZ_REQUIRE(ldy % 4 == 0 && ldx % 4 == 0 && ldy >= N && ldx >= K, "dhz_linear_wgrad: bad leading dims");
    hipStream_t s = (hipStream_t)stream;
    const int wm = (N % 128 == 0) ? 4 : (N % 96 == 0) ? 3 : (N % 64 == 0) ? 2 : 1;
    const int wn = (K % 128 == 0) ? 4 : (K % 64 == 0) ? 2 : 1;
    static const int big_env = getenv("DHZ_WGRAD_BIG") ? atoi(getenv("DHZ_WGRAD_BIG")) : 0;   // measured: +-5 %, not worth it by default
    if (big_env && N % 256 == 0 && K % 128 == 0) {          // compute-bound shapes: 256 x 128 tile on 8 waves
        launch<4, 4, 4>(dy, ldy, x, ldx, T, N, K, dw, db, s);
        DHZ_CHECK_LAUNCH("dhz_linear_wgrad");
        return DHZ_OK;
    }
#define CASE(a, b) if (wm == a && wn == b) launch<a, b>(dy, ldy, x, ldx, T, N, K, dw, db, s);
    CASE(1, 1) CASE(1, 2) CASE(1, 4) CASE(2, 1) CASE(2, 2) CASE(2, 4) CASE(3, 1) CASE(3, 2) CASE(3, 4)
    CASE(4, 1) CASE(4, 2) CASE(4, 4)
#undef CASE
    DHZ_CHECK_LAUNCH("dhz_linear_wgrad");
    return DHZ_OK;
}
