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

#ifndef WG_ABL
#define WG_ABL 0          // timing diagnostics (tools/variants.sh): 1 no epilogue atomics, 2 no MFMAs, 4 no global loads, 8 no bias sums
#endif
constexpr int TK = 32;   // tokens per LDS stage
// fragment vector width of an operand whose wave tile is W 16 x 16 tiles wide, and the conflict-free LDS row stride for it
constexpr int frag_vec(int W) { return (W == 4 || W == 2) ? W : 1; }
constexpr int row_stride(int B, int W) {
    return B + (frag_vec(W) == 4 ? (B % 64 ? 16 : 0) : frag_vec(W) == 2 ? (B % 64 ? 0 : 32) : 16);
}
constexpr int MAXMAT = 4;

// Output side of one launch: the N = nmat * nper rows of dY^T X belong to up to MAXMAT separate parameters of nper rows each
// (the Q / K / V projections share their input: one launch reads X once instead of three times).
struct WgradOut {
    float* dw[MAXMAT];
    float* db[MAXMAT];
    int nper;
    const float* rs;    // optional row scale of dy: row t is multiplied by rs[t / rps] (DropPath scale of its image); rps % TK == 0
    int rps;
};

// WM, WN: 16x16 tiles per wave along N (rows of dW) and K (cols of dW); waves are NWM x 2 (NWM = 2: 256 threads,
// NWM = 4: 512 threads and a 256-row tile - twice the FLOP per staged byte for the compute-bound shapes)
template <int WM, int WN, int NWM, int TG>
__global__ __launch_bounds__(128 * NWM * TG) void linear_wgrad_kernel(const float* __restrict__ dy, int ldy,
                                                                const float* __restrict__ x, int ldx, int T, int N,
                                                                int K, WgradOut out, int nsplit) {
    constexpr int GT = 128 * NWM;                      // threads per token group
    constexpr int NTHR = GT * TG;
    constexpr int BM = 16 * WM * NWM, BN = 32 * WN;
    // Fragment reads: both operands are staged "row = token, features along the row", and lane (i16, g) needs, for MFMA k-step s,
    // the values of token 4 s + g.  With the 16 x 16 tiles a / b of a wave INTERLEAVED over the features (tile a owns rows
    // WM i + a of dW, tile b columns WN j + b) a lane's WM (WN) values of one token are consecutive: ONE ds_read_b128 / b64
    // instead of WM (WN) ds_read_b32 - 2 LDS instructions per 16 MFMAs instead of 8 at WM = WN = 4 (the b32 form spent a
    // quarter of the matrix time issuing LDS reads beside the partner wave's MFMA stream).  Conflict-free row strides: b128
    // lane groups mix two token rows whose 16-byte slots interleave when the stride is a multiple of 64 floats (no pad); b64
    // and b32 lane groups want the next row 32 / 16 floats further.  WM = 3 keeps the classic tile-after-tile order.
    constexpr int VA = frag_vec(WM), VB = frag_vec(WN);
    constexpr int SA = row_stride(BM, WM), SB = row_stride(BN, WN);
    constexpr int A4 = BM / 4, B4 = BN / 4;            // float4 per staged row
    constexpr int NA = (TK * A4 + GT - 1) / GT, NB = (TK * B4 + GT - 1) / GT;   // float4 per thread per stage
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int STAGE = TK * (SA + SB);              // floats per stage: [A | B]

    const int t = threadIdx.x, lane = t & 63, w = (t >> 6) % (2 * NWM);
    const int tg = t / GT, tl = t % GT;                // token group of this wave, thread index inside it
    const int i16 = lane & 15, g = lane >> 4;
    const int wm = w >> 1, wn = w & 1;
    float* const gsm = smem + tg * 2 * STAGE;          // this group's two stages
    auto As = [&](int buf) -> float* { return gsm + buf * STAGE; };
    auto Bs = [&](int buf) -> float* { return gsm + buf * STAGE + TK * SA; };
    const int tiles_n = K / BN;
    int bid = blockIdx.x;
    const int split = bid % nsplit; bid /= nsplit;
    const int tn = bid % tiles_n, tm = bid / tiles_n;
    const int n0 = tm * BM, k0 = tn * BN;
    const int mat = n0 / out.nper, nloc = n0 - mat * out.nper;       // BM divides nper: a tile never straddles two parameters
    float* __restrict__ const dw = out.dw[mat];
    float* __restrict__ const db = out.db[mat];
    // token slab of this workgroup (multiples of TK); its stages are dealt round-robin to the TG token groups
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

    int st_loaded = 0;                                 // stage whose rows sit in ra / rb
    auto gload = [&](int st) {
        st_loaded = st;
        if (WG_ABL & 4) return;
        const size_t tok0 = (size_t)st * TK;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int e = tl + GT * i;
            if (TK * A4 % GT == 0 || e < TK * A4)
                ra[i] = *reinterpret_cast<const f32x4*>(dy + (tok0 + e / A4) * ldy + n0 + (e % A4) * 4);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int e = tl + GT * i;
            if (TK * B4 % GT == 0 || e < TK * B4)
                rb[i] = *reinterpret_cast<const f32x4*>(x + (tok0 + e / B4) * ldx + k0 + (e % B4) * 4);
        }
    };
    auto swrite = [&](int buf) {
        if (out.rs) {                                   // a stage (TK tokens) lies inside one image: one wave-uniform factor
            const float sc = out.rs[(st_loaded * TK) / out.rps];
#pragma unroll
            for (int i = 0; i < NA; ++i) ra[i] *= sc;
        }
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int e = tl + GT * i;
            if (TK * A4 % GT == 0 || e < TK * A4) {
                *reinterpret_cast<f32x4*>(&As(buf)[(e / A4) * SA + (e % A4) * 4]) = ra[i];
                if (!(WG_ABL & 8)) { dbacc[i].x += ra[i][0]; dbacc[i].y += ra[i][1]; dbacc[i].z += ra[i][2]; dbacc[i].w += ra[i][3]; }
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int e = tl + GT * i;
            if (TK * B4 % GT == 0 || e < TK * B4)
                *reinterpret_cast<f32x4*>(&Bs(buf)[(e / B4) * SB + (e % B4) * 4]) = rb[i];
        }
    };

    // group tg owns stages st0 + tg, st0 + tg + TG, ...; all groups run the same number of barriers
    const int niter = (st1 - st0 + TG - 1) / TG;
    if (st0 + tg < st1) {
        gload(st0 + tg);
        swrite(0);
    }
    __syncthreads();
    for (int it = 0; it < niter; ++it) {
        const int buf = it & 1;
        const int st = st0 + it * TG + tg;
        const bool more = st + TG < st1;
        if (more) gload(st + TG);
        if (st < st1) {
            const float* A = As(buf) + (wm * WM * 16 + VA * i16);
            const float* B = Bs(buf) + (wn * WN * 16 + VB * i16);
#pragma unroll
            for (int s = 0; s < TK / 4; ++s) {
                float af[WM], bf[WN];
                if constexpr (VA == 4) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(&A[(4 * s + g) * SA]);
                    af[0] = v[0]; af[1] = v[1]; af[2] = v[2]; af[3] = v[3];
                } else if constexpr (VA == 2) {
                    const float2 v = *reinterpret_cast<const float2*>(&A[(4 * s + g) * SA]);
                    af[0] = v.x; af[1] = v.y;
                } else {
#pragma unroll
                    for (int a = 0; a < WM; ++a) af[a] = A[(4 * s + g) * SA + 16 * a];
                }
                if constexpr (VB == 4) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(&B[(4 * s + g) * SB]);
                    bf[0] = v[0]; bf[1] = v[1]; bf[2] = v[2]; bf[3] = v[3];
                } else if constexpr (VB == 2) {
                    const float2 v = *reinterpret_cast<const float2*>(&B[(4 * s + g) * SB]);
                    bf[0] = v.x; bf[1] = v.y;
                } else {
#pragma unroll
                    for (int b = 0; b < WN; ++b) bf[b] = B[(4 * s + g) * SB + 16 * b];
                }
#pragma unroll
                for (int a = 0; a < WM; ++a)
#pragma unroll
                    for (int b = 0; b < WN; ++b) {
                        if (WG_ABL & 2) acc[a][b][0] += af[a] + bf[b];
                        else acc[a][b] = mfma16(af[a], bf[b], acc[a][b]);
                    }
            }
        }
        if (more) swrite(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: partial tiles of the token groups summed in LDS (row-major BM x BN, stride BN) -> full-line atomics
    float* Cs = smem;                                    // BM*BN floats <= 2 stages of LDS
#pragma unroll 1
    for (int r = TG - 1; r >= 0; --r) {
        if (tg == r) {
#pragma unroll
            for (int a = 0; a < WM; ++a)
#pragma unroll
                for (int b = 0; b < WN; ++b)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int rr = VA > 1 ? VA * (4 * g + j) + a : a * 16 + 4 * g + j;
                        const int cc = VB > 1 ? VB * i16 + b : b * 16 + i16;
                        float* c = &Cs[(wm * WM * 16 + rr) * BN + wn * WN * 16 + cc];
                        *c = (r == TG - 1) ? acc[a][b][j] : *c + acc[a][b][j];
                    }
        }
        __syncthreads();
    }
    // the nsplit workgroups of a tile finish together and add into the SAME BM x BN addresses: each starts its walk at another
    // row of the tile, so that at any moment they hit different cache lines instead of queueing on one
    const int rot = (int)(((unsigned)split * 2654435761u) % (unsigned)BM) * BN;
    for (int i = t; i < BM * BN; i += NTHR) {
        int e = i + rot;
        if (e >= BM * BN) e -= BM * BN;
        const int r = e / BN, c = e % BN;
        if (!(WG_ABL & 1) || Cs[e] == 12345.678f) atomicAdd(dw + (size_t)(nloc + r) * K + k0 + c, Cs[e]);
    }
    if (do_db) {
        // staged element e = tl + GT i sits at (row e / A4, float4-column e % A4) of every stage: dump the
        // TG*TK*A4 per-element sums and fold the TG*TK rows of each column.
        __syncthreads();
        float* red = smem + tg * TK * BM;                 // [TG][TK*A4][4] floats = TG*TK*BM <= TG stages
#pragma unroll
        for (int i = 0; i < NA; ++i)
            if (TK * A4 % GT == 0 || tl + GT * i < TK * A4) *reinterpret_cast<float4*>(&red[(tl + GT * i) * 4]) = dbacc[i];
        __syncthreads();
        if (t < BM) {
            const int c4 = t / 4, comp = t % 4;
            float tot = 0.f;
            for (int r = 0; r < TG * TK; ++r) tot += smem[(r * A4 + c4) * 4 + comp];
            atomicAdd(db + nloc + t, tot);
        }
    }
}

template <int WM, int WN, int NWM = 2, int TG = 1>
int launch(const float* dy, int ldy, const float* x, int ldx, int T, int N, int K, const WgradOut& out, hipStream_t s) {
    constexpr int BM = 16 * WM * NWM, BN = 32 * WN;
    constexpr size_t stage = (size_t)TK * (row_stride(BM, WM) + row_stride(BN, WN)) * sizeof(float);
    constexpr size_t smem = 2 * TG * stage > (size_t)BM * BN * 4 ? 2 * TG * stage : (size_t)BM * BN * 4;
    const int tiles = (N / BM) * (K / BN);
    // Every workgroup ends with BM*BN fp32 atomics (chip-wide ~1.3 TB/s of added bytes): large tiles want
    // fewer, longer token slabs.  DHZ_WGRAD_TARGET overrides the workgroup target (tuning aid).
#ifdef DHZ_DIAG
    static const int env_target = getenv("DHZ_WGRAD_TARGET") ? atoi(getenv("DHZ_WGRAD_TARGET")) : 0;
#else
    constexpr int env_target = 0;
#endif
    // Workgroups are dealt in whole rounds over the 256 CUs: the count must not exceed the resident slots (a 257th
    // 512-thread workgroup, or a 513th 256-thread one, runs alone after the others: measured 198 -> 130 us at
    // T=8192, N=1536, K=512 with 12 tiles x 43 splits = 516 workgroups), so the split count is rounded DOWN.
    const int ncu = dhz_num_cus();
    int target = env_target > 0 ? env_target : (NWM * TG == 4 ? ncu : 2 * ncu);   // 2 (1 for 512 threads) workgroups per CU
    if (2 * smem > 160 * 1024 && target > ncu) target = ncu;
    int nsplit = target / tiles;
    const int max_split = T / (TK * 4 * TG) > 0 ? T / (TK * 4 * TG) : 1;     // at least 4 stages per token group
    if (nsplit > max_split) nsplit = max_split;
    if (nsplit < 1) nsplit = 1;
    // short token slabs pay the per-workgroup epilogue (BM*BN atomics) over too few stages: below 16 stages per workgroup
    // trade splits for stages down to one workgroup per CU (measured: 70 -> 62 us at T=32768, N=512, K=128; 73 -> 64 us at
    // T=131072, N=K=128; the long-slab shapes are untouched)
    if (env_target <= 0 && (T / TK) / nsplit < 16) {
        int alt = ncu / tiles;
        if (alt < (T / TK) / 16) alt = (T / TK) / 16;
        if (alt >= 1 && alt < nsplit) nsplit = alt;
    }
    if (smem > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_wgrad_kernel<WM, WN, NWM, TG>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL((linear_wgrad_kernel<WM, WN, NWM, TG>), dim3(tiles * nsplit), dim3(128 * NWM * TG), smem, s, dy, ldy, x,
                       ldx, T, N, K, out, nsplit);
    return 0;
}

// Narrow shapes (parameter rows or columns a multiple of 16 but not of 32: the embed_dim = 16 model): 16 x 16 tiles of dW, one wave per
// tile and token slab, operands straight from global memory (lanes along the features: 64-byte runs), fp32 atomics at the end.
__global__ __launch_bounds__(256) void narrow_wgrad_kernel(const float* __restrict__ dy, int ldy, const float* __restrict__ x, int ldx,
                                                           int T, int N, int K, WgradOut out, int nslab) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, i16 = lane & 15, g = lane >> 4;
    const int tiles_k = K / 16, ntile = (N / 16) * tiles_k;
    const int slab = blockIdx.x % nslab, tgrp = blockIdx.x / nslab;
    const int tile = tgrp * 4 + w;
    if (tile >= ntile) return;
    const int n0 = (tile / tiles_k) * 16, k0 = (tile % tiles_k) * 16;
    const int mat = n0 / out.nper, nloc = n0 - mat * out.nper;
    const long c0 = (long)(T / 16) * slab / nslab, c1 = (long)(T / 16) * (slab + 1) / nslab;       // 16-token chunks of this slab
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    for (long c = c0; c < c1; ++c) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const long tk = 16 * c + 4 * g + s;
            float a = dy[tk * ldy + n0 + i16];
            if (out.rs) a *= out.rs[tk / out.rps];
            bsum += a;
            acc = mfma16(a, x[tk * ldx + k0 + i16], acc);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) atomicAdd(out.dw[mat] + (long)(nloc + 4 * g + j) * K + k0 + i16, acc[j]);
    if (out.db[mat] && k0 == 0) {
        bsum += __shfl_xor(bsum, 16);
        bsum += __shfl_xor(bsum, 32);
        if (g == 0) atomicAdd(out.db[mat] + nloc + i16, bsum);
    }
}

}  // namespace

// nmat parameters of nper rows each (N = nmat * nper columns of dy, consecutive)
static int wgrad_dispatch(const char* who, const float* dy, int ldy, const float* x, int ldx, int T, int nmat, int nper, int K,
                          const WgradOut& out, hipStream_t s) {
    const int N = nmat * nper;
    DHZ_REQUIRE(T > 0 && T % TK == 0, "%s: T=%d must be a multiple of %d", who, T, TK);
    if (nper > 0 && K > 0 && (nper % 32 || K % 32) && nper % 16 == 0 && K % 16 == 0) {         // narrow shapes (embed_dim 16)
        const int ntile = (N / 16) * (K / 16), groups = (ntile + 3) / 4;
        int nslab = 2048 / groups;                                                             // ~2048 workgroups, slabs of >= 64 tokens
        if (nslab > T / 64) nslab = T / 64;
        if (nslab < 1) nslab = 1;
        hipLaunchKernelGGL(narrow_wgrad_kernel, dim3(groups * nslab), dim3(256), 0, s, dy, ldy, x, ldx, T, N, K, out, nslab);
        DHZ_CHECK_LAUNCH(who);
        return DHZ_OK;
    }
    DHZ_REQUIRE(nper % 32 == 0 && K % 32 == 0 && nper > 0 && K > 0, "%s: N=%d K=%d must be multiples of 16", who, nper, K);
    DHZ_REQUIRE(ldy % 4 == 0 && ldx % 4 == 0 && ldy >= N && ldx >= K, "%s: bad leading dims", who);
    // tile rows divide nper, so that a tile never straddles two parameters
    const int wm = (nper % 128 == 0) ? 4 : (nper % 96 == 0) ? 3 : (nper % 64 == 0) ? 2 : 1;
    const int wn = (K % 128 == 0) ? 4 : (K % 64 == 0) ? 2 : 1;
    // Two token groups per workgroup (512 threads, one 128 x 128 workgroup per CU) halve the atomic epilogue per staged byte:
    // -7.5 % over the 36 shapes of the step (tools/bench_wgrad.py); slabs shorter than 8 stages stay on one group.
#ifdef DHZ_DIAG
    static const int tg_env = getenv("DHZ_WGRAD_TG") ? atoi(getenv("DHZ_WGRAD_TG")) : 0;
#else
    constexpr int tg_env = 0;
#endif
    const int bm = 32 * wm, bn = 32 * wn;
    const int tiles = (N / bm) * (K / bn);
    const int splits2 = 256 / tiles > 0 ? 256 / tiles : 1;
    const bool two = tg_env ? tg_env == 2 : (T / TK) / splits2 >= 8;
#define CASE(a, b)                                                               \
    if (wm == a && wn == b) {                                                    \
        if (two) launch<a, b, 2, 2>(dy, ldy, x, ldx, T, N, K, out, s);           \
        else launch<a, b>(dy, ldy, x, ldx, T, N, K, out, s);                     \
    }
    CASE(1, 1) CASE(1, 2) CASE(1, 4) CASE(2, 1) CASE(2, 2) CASE(2, 4) CASE(3, 1) CASE(3, 2) CASE(3, 4)
    CASE(4, 1) CASE(4, 2) CASE(4, 4)
#undef CASE
    DHZ_CHECK_LAUNCH(who);
    return DHZ_OK;
}

extern "C" int dhz_linear_wgrad(const float* dy, int ldy, const float* x, int ldx, int T, int N, int K, float* dw,
                                float* db, void* stream) {
    DHZ_REQUIRE(dy && x && dw, "dhz_linear_wgrad: null pointer");
    WgradOut out = {};
    out.dw[0] = dw; out.db[0] = db; out.nper = N;
    return wgrad_dispatch("dhz_linear_wgrad", dy, ldy, x, ldx, T, 1, N, K, out, (hipStream_t)stream);
}

extern "C" int dhz_linear_wgrad_rs(const float* dy, int ldy, const float* x, int ldx, int T, int N, int K, float* dw, float* db,
                                   const float* row_scale, int rows_per_scale, void* stream) {
    DHZ_REQUIRE(dy && x && dw, "dhz_linear_wgrad_rs: null pointer");
    DHZ_REQUIRE(!row_scale || (rows_per_scale > 0 && rows_per_scale % TK == 0 && T % rows_per_scale == 0),
                "dhz_linear_wgrad_rs: rows_per_scale=%d must be a multiple of %d that divides T=%d", rows_per_scale, TK, T);
    WgradOut out = {};
    out.dw[0] = dw; out.db[0] = db; out.nper = N;
    out.rs = row_scale; out.rps = rows_per_scale;
    return wgrad_dispatch("dhz_linear_wgrad_rs", dy, ldy, x, ldx, T, 1, N, K, out, (hipStream_t)stream);
}

extern "C" int dhz_linear_wgrad_multi(const float* dy, int ldy, const float* x, int ldx, int T, int nmat, int nper, int K,
                                      float* const* dw, float* const* db, void* stream) {
    DHZ_REQUIRE(dy && x && dw, "dhz_linear_wgrad_multi: null pointer");
    DHZ_REQUIRE(nmat >= 1 && nmat <= MAXMAT, "dhz_linear_wgrad_multi: nmat=%d must be 1..%d", nmat, MAXMAT);
    WgradOut out = {};
    for (int i = 0; i < nmat; ++i) {
        DHZ_REQUIRE(dw[i], "dhz_linear_wgrad_multi: null dw[%d]", i);
        DHZ_REQUIRE(!db || (db[i] != nullptr) == (db[0] != nullptr), "dhz_linear_wgrad_multi: bias gradients must be all set or all null");
        out.dw[i] = dw[i];
        out.db[i] = db ? db[i] : nullptr;
    }
    out.nper = nper;
    return wgrad_dispatch("dhz_linear_wgrad_multi", dy, ldy, x, ldx, T, nmat, nper, K, out, (hipStream_t)stream);
}
