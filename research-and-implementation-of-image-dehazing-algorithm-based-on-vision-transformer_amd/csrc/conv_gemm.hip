// K8: the 4x4 / stride-2 / pad-1 convolution of Downsample (M1:606-622) on the token layout [B, H*W, Cin] -> [B, (H/2)*(W/2), Cout],
// forward, backward-data and weight gradient, as implicit GEMMs on the fp32 matrix pipe.  No im2col matrix is materialised:
// a token's Cin channels are contiguous, so a tap (ky, kx) of an output pixel is one contiguous run of Cin floats and the
// operand loader of the token GEMM (csrc/linear_gemm.hip: same tiles, swizzled LDS images, persistent workgroups) only has to
// compute a row pointer and a validity bit per row and stage.
//   forward        y[m, co]  = sum_{ky,kx,ci} x[b, 2oy-1+ky, 2ox-1+kx, ci] wp[co, (ky,kx,ci)] + bias       m = (b, oy, ox), K = 16 Cin
//   backward-data  dx[b, iy, ix, ci] = sum over the 2 x 2 taps whose parity matches (iy, ix) and co of
//                  dy[b, (iy+1-ky)/2, (ix+1-kx)/2, co] wq[(ky,kx,co), ci]      one GEMM per parity class, K = 4 Cout
//   weight grad    dwp[co, (ky,kx,ci)] += sum_m dy[m, co] x[b, 2oy-1+ky, 2ox-1+kx, ci]   split over m, fp32 atomics
// wp = weight.permute(0,2,3,1) [Cout, 16 Cin], wq = weight.permute(2,3,0,1) [16 Cout, Cin] (host side, small).
#include "common.h"

namespace {

constexpr int BK = 32;

struct ConvGeom {
    int H, W;            // input map (the side with 2x the pixels)
    int Ho, Wo;          // output map = H/2, W/2
    int Cin, Cout;
    int py, px;          // backward-data: parity class of this launch
};

// MODE 1: forward (A = x gathered, B = wp [N][K] K-contiguous)      MODE 2: backward-data (A = dy gathered, B = wq rows = contraction)
template <int WM, int WN, int MODE>
__global__ __launch_bounds__(256) void conv_gemm_kernel(const float* __restrict__ A, const float* __restrict__ Wt,
                                                        const float* __restrict__ bias, float* __restrict__ Y, ConvGeom G,
                                                        int M, int N, int K, int tiles_n, int ntiles) {
    constexpr bool WT = MODE == 1;
    constexpr int BM = 32 * WM, BN = 32 * WN;
    constexpr int SBN = BN + 4;
    constexpr int A_FLOATS = BM * BK;
    constexpr int B_FLOATS = WT ? BN * BK : BK * SBN;
    constexpr int STAGE = A_FLOATS + B_FLOATS;
    constexpr int NA = WM, NB = WN;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int i16 = lane & 15, g = lane >> 4;
    const int wm = w >> 1, wn = w & 1;
    const int nst = K / BK;
    const int CA = MODE == 1 ? G.Cin : G.Cout;             // channels of the gathered operand
    const int spt = CA / BK;                               // stages per tap
    const int ldw = MODE == 1 ? K : G.Cin;
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

    f32x4 ra[NA], rb[NB];
    const float* pa[NA];       // MODE 1: &x[b, 2oy-1, 2ox-1, 4q]   MODE 2: &dy[b, iy2, ix2, 4q]   (may lie outside the map: see valid)
    int ya[NA], xa[NA];        // MODE 1: 2oy-1, 2ox-1              MODE 2: iy2, ix2
    const float* pb[NB];
    int nb0 = 0;
    auto set_tile = [&](int tile) {
        const int tn = tile % tiles_n, tm = tile / tiles_n;
        const int m0 = tm * BM, n0 = tn * BN;
        nb0 = n0;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int e = t + 256 * i, row = e >> 3, q = e & 7;
            const int m = min(m0 + row, M - 1);
            const int b = m / (G.Ho * G.Wo), r = m - b * (G.Ho * G.Wo);
            const int oy = r / G.Wo, ox = r - oy * G.Wo;
            if (MODE == 1) {
                ya[i] = 2 * oy - 1; xa[i] = 2 * ox - 1;
                pa[i] = A + ((long long)(b * G.H + ya[i]) * G.W + xa[i]) * G.Cin + 4 * q;
            } else {
                ya[i] = oy; xa[i] = ox;
                pa[i] = A + ((long long)(b * G.Ho + oy) * G.Wo + ox) * G.Cout + 4 * q;
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int e = t + 256 * i;
            if (WT) pb[i] = Wt + (size_t)(n0 + (e >> 3)) * ldw + 4 * (e & 7);
            else pb[i] = Wt + (size_t)(e / (BN / 4)) * ldw + n0 + 4 * (e % (BN / 4));
        }
    };
    auto gload = [&](int st) {
        const int tap = st / spt, c0 = (st - tap * spt) * BK;          // uniform: the 32-deep stage lies inside one tap
        if (MODE == 1) {
            const int ky = tap >> 2, kx = tap & 3;
            const long long delta = (long long)(ky * G.W + kx) * G.Cin + c0;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const bool valid = (unsigned)(ya[i] + ky) < (unsigned)G.H && (unsigned)(xa[i] + kx) < (unsigned)G.W;
                ra[i] = valid ? *reinterpret_cast<const f32x4*>(pa[i] + delta) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int i = 0; i < NB; ++i) rb[i] = *reinterpret_cast<const f32x4*>(pb[i] + (size_t)st * BK);
        } else {
            const int a2 = tap >> 1, c2 = tap & 1;                      // tap (a2, c2) of the 2 x 2 set of this parity class
            const int dy_ = G.py - a2, dx_ = G.px - c2;                 // oy = iy2 + py - a2, ox = ix2 + px - c2
            const int ky = (1 - G.py) + 2 * a2, kx = (1 - G.px) + 2 * c2;
            const long long delta = (long long)(dy_ * G.Wo + dx_) * G.Cout + c0;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const bool valid = (unsigned)(ya[i] + dy_) < (unsigned)G.Ho && (unsigned)(xa[i] + dx_) < (unsigned)G.Wo;
                ra[i] = valid ? *reinterpret_cast<const f32x4*>(pa[i] + delta) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
            const size_t krow = (size_t)((ky * 4 + kx) * G.Cout + c0);   // rows of wq [(ky,kx,co)][ci]
#pragma unroll
            for (int i = 0; i < NB; ++i) rb[i] = *reinterpret_cast<const f32x4*>(pb[i] + krow * ldw);
        }
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
                *reinterpret_cast<f32x4*>(&Bs[(e / (BN / 4)) * SBN + 4 * (e % (BN / 4))]) = rb[i];
            }
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
        for (int st = 0; st < nst; ++st) {
            const bool last = st + 1 == nst;
            const bool more = !last || ntile >= 0;
            if (!last) gload(st + 1);
            else if (ntile >= 0) { set_tile(ntile); gload(0); }
            const float* As = smem + buf * STAGE + (wm * WM * 16 + i16) * BK;
            const float* Bs = smem + buf * STAGE + A_FLOATS;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                f32x4 af[WM], bf[WN];
#pragma unroll
                for (int a = 0; a < WM; ++a) af[a] = *reinterpret_cast<const f32x4*>(&As[a * 16 * BK + 4 * ((4 * s + g) ^ sw)]);
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
                        for (int b = 0; b < WN; ++b) acc[a][b] = mfma16(af[a][j], bf[b][j], acc[a][b]);
            }
            if (more) {
                swrite(buf ^ 1);
                __syncthreads();
                buf ^= 1;
            }
        }
        {   // epilogue: acc[a][b][j] = C[16 a + 4 g + j][16 b + i16].  Full tiles store without per-row guards (behind a
            // divergent guard hipcc waits for vmcnt(0) - all earlier stores acknowledged - before every guarded group)
            const int m0 = tm * BM + wm * WM * 16 + 4 * g, n0 = tn * BN + wn * WN * 16 + i16;
            float bv[WN];
#pragma unroll
            for (int b = 0; b < WN; ++b) bv[b] = bias ? bias[n0 + 16 * b] : 0.f;
            auto out_row = [&](int m) -> size_t {
                if (MODE == 2) {                // row (b, iy2, ix2) of this parity class -> token (b, 2 iy2 + py, 2 ix2 + px)
                    const int b = m / (G.Ho * G.Wo), r = m - b * (G.Ho * G.Wo);
                    const int iy2 = r / G.Wo, ix2 = r - iy2 * G.Wo;
                    return ((size_t)b * G.H + 2 * iy2 + G.py) * G.W + 2 * ix2 + G.px;
                }
                return (size_t)m;
            };
            if (tm * BM + BM <= M) {                                     // wave-uniform
                float* yr[WM][4];
#pragma unroll
                for (int a = 0; a < WM; ++a)
#pragma unroll
                    for (int j = 0; j < 4; ++j) yr[a][j] = Y + out_row(m0 + 16 * a + j) * N + n0;
#pragma unroll
                for (int a = 0; a < WM; ++a)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int b = 0; b < WN; ++b) yr[a][j][16 * b] = acc[a][b][j] + bv[b];
            } else {
#pragma unroll
                for (int a = 0; a < WM; ++a)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int m = m0 + 16 * a + j;
                        if (m < M) {
                            float* yr = Y + out_row(m) * N + n0;
#pragma unroll
                            for (int b = 0; b < WN; ++b) yr[16 * b] = acc[a][b][j] + bv[b];
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

template <int WM, int WN, int MODE>
void launch(const float* A, const float* Wt, const float* bias, float* Y, const ConvGeom& G, int M, int N, int K, hipStream_t s) {
    constexpr int BM = 32 * WM, BN = 32 * WN;
    constexpr size_t stage = (size_t)(BM * BK + (MODE == 1 ? BN * BK : BK * (BN + 4))) * sizeof(float);
    constexpr size_t smem = 2 * stage;
    const int tiles_n = N / BN, tiles_m = (M + BM - 1) / BM;
    const int ntiles = tiles_n * tiles_m;
    const int slots = 2 * dhz_num_cus();                          // two resident workgroups per CU
    const int grid = ntiles < slots ? ntiles : slots;
    if (smem > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_gemm_kernel<WM, WN, MODE>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL((conv_gemm_kernel<WM, WN, MODE>), dim3(grid), dim3(256), smem, s, A, Wt, bias, Y, G, M, N, K, tiles_n, ntiles);
}

template <int MODE>
void dispatch(const float* A, const float* Wt, const float* bias, float* Y, const ConvGeom& G, int M, int N, int K, hipStream_t s) {
    int wm = 2, wn = 1;
    static const int cand[8][2] = {{4, 4}, {4, 3}, {4, 2}, {2, 4}, {2, 3}, {2, 2}, {4, 1}, {2, 1}};
    long best = -1;
    const long slots = 2 * dhz_num_cus();
    for (int i = 0; i < 8; ++i) {
        const int a = cand[i][0], b = cand[i][1];
        if (N % (32 * b)) continue;
        const long blocks = (long)((M + 32 * a - 1) / (32 * a)) * (N / (32 * b));
        if (blocks >= slots) { wm = a; wn = b; break; }
        if (blocks > best) { best = blocks; wm = a; wn = b; }
    }
#define CASE(a, b) \
    if (wm == a && wn == b) launch<a, b, MODE>(A, Wt, bias, Y, G, M, N, K, s);
    CASE(4, 1) CASE(4, 2) CASE(4, 3) CASE(4, 4) CASE(2, 1) CASE(2, 2) CASE(2, 3) CASE(2, 4)
#undef CASE
}

// ------------------------------------------------------------------------------------------------ weight gradient
// dwp[n, k] += sum_m dy[m, n] xcol[m, k],  k = (tap, ci): the split-T kernel of csrc/linear_wgrad.hip (one token group) with a
// gathered x operand; a K tile (32 WN columns, WN | Cin / 32) lies inside one tap.
constexpr int TK = 32;
template <int WM, int WN>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                         float* __restrict__ dw, float* __restrict__ db, ConvGeom G, int T, int N,
                                                         int K, int nsplit, int lgWo, int lgHW) {
    constexpr int BM = 32 * WM, BN = 32 * WN;
    constexpr int SA = BM + 16, SB = BN + 16;
    constexpr int A4 = BM / 4, B4 = BN / 4;
    constexpr int NA = (TK * A4 + 255) / 256, NB = (TK * B4 + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int STAGE = TK * (SA + SB);
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int i16 = lane & 15, g = lane >> 4;
    const int wm = w >> 1, wn = w & 1;
    const int tiles_n = K / BN;
    int bid = blockIdx.x;
    const int split = bid % nsplit; bid /= nsplit;
    const int tn = bid % tiles_n, tm = bid / tiles_n;
    const int n0 = tm * BM, k0 = tn * BN;
    const int tap = k0 / G.Cin, c0 = k0 - tap * G.Cin, ky = tap >> 2, kx = tap & 3;
    const int nst = T / TK;
    const int st0 = (int)((long long)nst * split / nsplit), st1 = (int)((long long)nst * (split + 1) / nsplit);
    f32x4 acc[WM][WN];
#pragma unroll
    for (int a = 0; a < WM; ++a)
#pragma unroll
        for (int b = 0; b < WN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 ra[NA], rb[NB];
    float4 dbacc[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) dbacc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool do_db = (db != nullptr) && (tn == 0);
    auto gload = [&](int st) {
        const int tok0 = st * TK;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int e = t + 256 * i;
            if (TK * A4 % 256 == 0 || e < TK * A4) ra[i] = *reinterpret_cast<const f32x4*>(dy + (size_t)(tok0 + e / A4) * N + n0 + (e % A4) * 4);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int e = t + 256 * i;
            if (TK * B4 % 256 == 0 || e < TK * B4) {
                const int m = tok0 + e / B4;                      // output pixel (b, oy, ox); maps are powers of two
                const int b = m >> lgHW, oy = (m >> lgWo) & (G.Ho - 1), ox = m & (G.Wo - 1);
                const int iy = 2 * oy - 1 + ky, ix = 2 * ox - 1 + kx;
                const bool valid = (unsigned)iy < (unsigned)G.H && (unsigned)ix < (unsigned)G.W;
                rb[i] = valid ? *reinterpret_cast<const f32x4*>(x + ((size_t)(b * G.H + iy) * G.W + ix) * G.Cin + c0 + (e % B4) * 4)
                              : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    auto swrite = [&](int buf) {
        float* As = smem + buf * STAGE;
        float* Bs = As + TK * SA;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int e = t + 256 * i;
            if (TK * A4 % 256 == 0 || e < TK * A4) {
                *reinterpret_cast<f32x4*>(&As[(e / A4) * SA + (e % A4) * 4]) = ra[i];
                dbacc[i].x += ra[i][0]; dbacc[i].y += ra[i][1]; dbacc[i].z += ra[i][2]; dbacc[i].w += ra[i][3];
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int e = t + 256 * i;
            if (TK * B4 % 256 == 0 || e < TK * B4) *reinterpret_cast<f32x4*>(&Bs[(e / B4) * SB + (e % B4) * 4]) = rb[i];
        }
    };
    if (st0 < st1) { gload(st0); swrite(0); }
    __syncthreads();
    for (int st = st0; st < st1; ++st) {
        const int buf = (st - st0) & 1;
        const bool more = st + 1 < st1;
        if (more) gload(st + 1);
        const float* Af = smem + buf * STAGE + (wm * WM * 16 + i16);
        const float* Bf = smem + buf * STAGE + TK * SA + (wn * WN * 16 + i16);
#pragma unroll
        for (int s = 0; s < TK / 4; ++s) {
            float af[WM], bf[WN];
#pragma unroll
            for (int a = 0; a < WM; ++a) af[a] = Af[(4 * s + g) * SA + 16 * a];
#pragma unroll
            for (int b = 0; b < WN; ++b) bf[b] = Bf[(4 * s + g) * SB + 16 * b];
#pragma unroll
            for (int a = 0; a < WM; ++a)
#pragma unroll
                for (int b = 0; b < WN; ++b) acc[a][b] = mfma16(af[a], bf[b], acc[a][b]);
        }
        if (more) swrite(buf ^ 1);
        __syncthreads();
    }
    float* Cs = smem;
#pragma unroll
    for (int a = 0; a < WM; ++a)
#pragma unroll
        for (int b = 0; b < WN; ++b)
#pragma unroll
            for (int j = 0; j < 4; ++j) Cs[(wm * WM * 16 + a * 16 + 4 * g + j) * BN + wn * WN * 16 + b * 16 + i16] = acc[a][b][j];
    __syncthreads();
    for (int e = t; e < BM * BN; e += 256) atomicAdd(dw + (size_t)(n0 + e / BN) * K + k0 + e % BN, Cs[e]);
    if (do_db) {
        __syncthreads();
        float* red = smem;                                 // [TK*A4][4]
#pragma unroll
        for (int i = 0; i < NA; ++i)
            if (TK * A4 % 256 == 0 || t + 256 * i < TK * A4) *reinterpret_cast<float4*>(&red[(t + 256 * i) * 4]) = dbacc[i];
        __syncthreads();
        if (t < BM) {
            const int c4 = t / 4, comp = t % 4;
            float tot = 0.f;
            for (int r = 0; r < TK; ++r) tot += smem[(r * A4 + c4) * 4 + comp];
            atomicAdd(db + n0 + t, tot);
        }
    }
}

template <int WM, int WN>
void launch_wgrad(const float* dy, const float* x, float* dw, float* db, const ConvGeom& G, int T, int N, int K, int lgWo, int lgHW,
                  hipStream_t s) {
    constexpr int BM = 32 * WM, BN = 32 * WN;
    constexpr size_t stage = (size_t)TK * (BM + 16 + BN + 16) * sizeof(float);
    constexpr size_t smem = 2 * stage > (size_t)BM * BN * 4 ? 2 * stage : (size_t)BM * BN * 4;
    const int tiles = (N / BM) * (K / BN);
    int nsplit = 2 * dhz_num_cus() / tiles;
    const int max_split = T / (TK * 4) > 0 ? T / (TK * 4) : 1;
    if (nsplit > max_split) nsplit = max_split;
    if (nsplit < 1) nsplit = 1;
    if (smem > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_kernel<WM, WN>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)smem);
    hipLaunchKernelGGL((conv_wgrad_kernel<WM, WN>), dim3(tiles * nsplit), dim3(256), smem, s, dy, x, dw, db, G, T, N, K, nsplit, lgWo,
                       lgHW);
}

int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return (1 << l) == v ? l : -1; }

}  // namespace

extern "C" int dhz_conv4s2_fwd(const float* x, const float* wp, const float* bias, float* y, int B, int H, int W, int Cin, int Cout,
                               void* stream) {
    DHZ_REQUIRE(x && wp && y, "dhz_conv4s2_fwd: null pointer");
    DHZ_REQUIRE(B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && Cin % 32 == 0 && Cout % 32 == 0 && Cin > 0 && Cout > 0,
                "dhz_conv4s2_fwd: map %dx%d (even sizes) Cin=%d Cout=%d (multiples of 32)", H, W, Cin, Cout);
    ConvGeom G = {H, W, H / 2, W / 2, Cin, Cout, 0, 0};
    dispatch<1>(x, wp, bias, y, G, B * G.Ho * G.Wo, Cout, 16 * Cin, (hipStream_t)stream);
    DHZ_CHECK_LAUNCH("dhz_conv4s2_fwd");
    return DHZ_OK;
}

extern "C" int dhz_conv4s2_dgrad(const float* dy, const float* wq, float* dx, int B, int H, int W, int Cin, int Cout, void* stream) {
    DHZ_REQUIRE(dy && wq && dx, "dhz_conv4s2_dgrad: null pointer");
    DHZ_REQUIRE(B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && Cin % 32 == 0 && Cout % 32 == 0 && Cin > 0 && Cout > 0,
                "dhz_conv4s2_dgrad: map %dx%d (even sizes) Cin=%d Cout=%d (multiples of 32)", H, W, Cin, Cout);
    for (int p = 0; p < 4; ++p) {
        ConvGeom G = {H, W, H / 2, W / 2, Cin, Cout, p >> 1, p & 1};
        dispatch<2>(dy, wq, nullptr, dx, G, B * G.Ho * G.Wo, Cin, 4 * Cout, (hipStream_t)stream);
    }
    DHZ_CHECK_LAUNCH("dhz_conv4s2_dgrad");
    return DHZ_OK;
}

extern "C" int dhz_conv4s2_wgrad(const float* dy, const float* x, float* dwp, float* db, int B, int H, int W, int Cin, int Cout,
                                 void* stream) {
    DHZ_REQUIRE(dy && x && dwp, "dhz_conv4s2_wgrad: null pointer");
    DHZ_REQUIRE(B > 0 && H % 2 == 0 && W % 2 == 0 && Cin % 32 == 0 && Cout % 32 == 0 && Cin > 0 && Cout > 0,
                "dhz_conv4s2_wgrad: map %dx%d (even sizes) Cin=%d Cout=%d (multiples of 32)", H, W, Cin, Cout);
    const int Ho = H / 2, Wo = W / 2;
    const int lgWo = ilog2(Wo), lgHW = ilog2(Ho * Wo);
    DHZ_REQUIRE(lgWo >= 0 && lgHW >= 0, "dhz_conv4s2_wgrad: output map %dx%d must be powers of two", Ho, Wo);
    const int T = B * Ho * Wo;
    DHZ_REQUIRE(T % TK == 0, "dhz_conv4s2_wgrad: B*Ho*Wo = %d must be a multiple of %d", T, TK);
    ConvGeom G = {H, W, Ho, Wo, Cin, Cout, 0, 0};
    const int N = Cout, K = 16 * Cin;
    const int wm = (N % 128 == 0) ? 4 : (N % 96 == 0) ? 3 : (N % 64 == 0) ? 2 : 1;
    const int wn = (Cin % 128 == 0) ? 4 : (Cin % 64 == 0) ? 2 : 1;               // a K tile stays inside one tap
#define CASE(a, b) \
    if (wm == a && wn == b) launch_wgrad<a, b>(dy, x, dwp, db, G, T, N, K, lgWo, lgHW, (hipStream_t)stream);
    CASE(1, 1) CASE(1, 2) CASE(1, 4) CASE(2, 1) CASE(2, 2) CASE(2, 4) CASE(3, 1) CASE(3, 2) CASE(3, 4) CASE(4, 1) CASE(4, 2) CASE(4, 4)
#undef CASE
    DHZ_CHECK_LAUNCH("dhz_conv4s2_wgrad");
    return DHZ_OK;
}
