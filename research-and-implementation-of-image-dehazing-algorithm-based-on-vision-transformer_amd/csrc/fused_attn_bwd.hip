// Fused BACKWARD of the window-attention branch of a LeWin block at C = 32 (one head, head_dim 32):
//
//   out = x + s[b] * OutProj( ProbSparseAttention( QKV( roll+partition( LayerNorm(x) ) ) ) )       (M1:839-872, ATT:287-342, 385-461)
//
// given d(out), in ONE kernel: the forward's intermediates are RECOMPUTED per window from x (LayerNorm, Q/K/V projection, scores
// of the selected rows, double softmax, O = P V) - only the 64 selection ranks per window are read back from the forward - and
// every gradient is produced on chip:
//   dy = s dout;   dctx = dy Wo;   dO[r] = dctx[top r], dO[25] = sum over the unselected queries;
//   dV = P2^T dO,  dP2 = dO V^T,  softmax backward twice (-> d bias), dQ[top] = dS K,  dK = dS^T Q[top];
//   dxn = [dQ | dK | dV] Wqkv;    dx = dout + LayerNorm-backward(dxn);
//   dWq/k/v += [dQ|dK|dV]^T xn, dWo += dy^T ctx (accumulated in REGISTERS across the windows of a persistent workgroup, one atomic
//   per element and workgroup at the end), biases, d gamma, d beta likewise.
// It replaces, per block: dhz_reverse_residual_bwd, two backward-data GEMMs, three weight-gradient launches, dhz_ps_attn_bwd and
// dhz_ln_partition_bwd, and lets the fused forward drop its xn / QKV / context / statistics saves (7 C floats per token -> 64
// bytes per window).  288 MFMAs per wave and window against 112 in the forward (fp32 MFMA and VALU time add on this pipe).
// LDS 75 KB (two workgroups per CU): the regions of the attention-core backward (csrc/ps_attn.hip) with the projection phases
// overlaid on them once they are dead, + the two backward-orientation weight images (16 KB).
#include <stdlib.h>
#include "common.h"

namespace {

constexpr int NT = 64, NU = 25, SS = 68, DS = 36, C = 32;

__device__ __forceinline__ float row8_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 1));
    v = fmaxf(v, __shfl_xor(v, 2));
    return fmaxf(v, __shfl_xor(v, 4));
}
__device__ __forceinline__ float row8_sum(float v) {
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    return v + __shfl_xor(v, 4);
}
// the forward kernel's double softmax (csrc/fused_attn.hip step 2d), 8 columns per thread; scores arrive scaled
__device__ __forceinline__ void double_softmax8(const float* x, const float* brow, const float* mrow, float* p1, float* p2) {
    float mx = x[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) mx = fmaxf(mx, x[i]);
    mx = row8_max(mx);
    float e[8], sum = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { e[i] = __expf(x[i] - mx); sum += e[i]; }
    sum = __builtin_amdgcn_rcpf(row8_sum(sum));
    float a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { p1[i] = e[i] * sum; a[i] = p1[i]; }
    if (brow) {
        const float4 b0 = *reinterpret_cast<const float4*>(brow), b1 = *reinterpret_cast<const float4*>(brow + 4);
        a[0] += b0.x; a[1] += b0.y; a[2] += b0.z; a[3] += b0.w; a[4] += b1.x; a[5] += b1.y; a[6] += b1.z; a[7] += b1.w;
    }
    if (mrow) {
        const float4 b0 = *reinterpret_cast<const float4*>(mrow), b1 = *reinterpret_cast<const float4*>(mrow + 4);
        a[0] += b0.x; a[1] += b0.y; a[2] += b0.z; a[3] += b0.w; a[4] += b1.x; a[5] += b1.y; a[6] += b1.z; a[7] += b1.w;
    }
    float mx2 = a[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) mx2 = fmaxf(mx2, a[i]);
    mx2 = row8_max(mx2);
    float sum2 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { e[i] = __expf(a[i] - mx2); sum2 += e[i]; }
    sum2 = __builtin_amdgcn_rcpf(row8_sum(sum2));
#pragma unroll
    for (int i = 0; i < 8; ++i) p2[i] = e[i] * sum2;
}

__device__ __forceinline__ void ld4(const float* p, float* dst) {
    const float4 v = *reinterpret_cast<const float4*>(p);
    dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
}

struct BwdSmem {
    float k[NT * DS];        // K; later dK; at the very end the dxn staging rows (wave-private)
    float v[NT * DS];        // V; later dV
    float qr[32 * DS];       // Q[top] rows (25 live)                 } later, contiguous: xn rows [64][DS]
    float dor[32 * DS];      // dO[top] rows, row 25 = unselected sum  }
    float p1[32 * SS];       // selected scores -> P1; later dQ[top] (32 x DS)
    float p2[32 * SS];       // P2; later dA                           } later, contiguous from p2: dy rows [64][DS]
    float dsb[32 * SS];      // dP2 -> dS                              }
    float o[32 * DS];        // O = P2 V (row 25 = mean(V)): the context rows of the out-projection's weight gradient
    float wt[4096];          // backward-orientation weight fragments: Wqkv (3 x 2 x 2 float4 per lane), Wo (2 x 2)
    float gam[C], bet[C];
    float bq[3 * C];         // Q, K, V projection biases
    float red[2 * C];        // end-of-kernel reduction of d gamma / d beta
    int top[32];
    uint8_t rank[NT];
};

struct BwdOut {
    float* dx;
    float* dw[3];            // dWq, dWk, dWv [32, 32]
    float* db[3];            // dbq, dbk, dbv [32]
    float* dwo;
    float* dbo;
    float* dgamma;
    float* dbeta;
    float* dbias_part;       // [grid][64][64] or null
};

// wqkv_t[m(3)][tn(2)][s4(2)][lane(64)] float4 over r = W_m[8 g + 4 s4 + r][16 tn + i16]      (B[k = feature][n = c])
// wo_t  [tn(2)][s4(2)][lane(64)] float4 over r        = Wo [8 g + 4 s4 + r][16 tn + i16]      (B[k = o][n = c])
__global__ void prepack_bwd_kernel(const float* __restrict__ wq, const float* __restrict__ wk, const float* __restrict__ wv,
                                   const float* __restrict__ wo, float* __restrict__ wt) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= 4096) return;
    const int r = e & 3, lane = (e >> 2) & 63, i16 = lane & 15, g = lane >> 4;
    int rest = e >> 8;                                   // 0..15
    const int s4 = rest & 1, tn = (rest >> 1) & 1, m = rest >> 2;        // m = 3: Wo
    const float* W = m == 0 ? wq : (m == 1 ? wk : (m == 2 ? wv : wo));
    wt[e] = W[(8 * g + 4 * s4 + r) * C + 16 * tn + i16];
}

template <bool HAS_BIAS>
__global__ __launch_bounds__(256, 2) void fused_window_attn_bwd_c32_kernel(
    const float* __restrict__ x, const float* __restrict__ dout, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float4* __restrict__ wqkv_p, const float* __restrict__ bqkv, const float* __restrict__ wt_g,
    const float* __restrict__ bias, const float* __restrict__ mask, const float* __restrict__ dscale,
    const uint8_t* __restrict__ rank_in, BwdOut out, int Hres, int Wres, int shift, int nwin) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    BwdSmem& sm = *reinterpret_cast<BwdSmem*>(smem_raw);
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int i16 = lane & 15, g = lane >> 4;
    const int nWw = Wres >> 3, nW = (Hres >> 3) * nWw;
    const int tl = 16 * w + i16;                          // token row of this lane in the row phases = row of its A fragments
    const float scale = 0.17677669529663687f;             // 1/sqrt(32)
    float* const xn_rows = sm.qr;                         // [64][DS] over qr | dor
    float* const dy_rows = sm.p2;                         // [64][DS] over p2 | dsb
    auto tok4sum = [](float v) -> float { v += __shfl_xor(v, 16); return v + __shfl_xor(v, 32); };
    auto src_token = [&](int win) -> size_t {
        const int bimg = win / nW, wdx = win % nW;
        int hh = (wdx / nWw) * 8 + (tl >> 3) + shift; if (hh >= Hres) hh -= Hres;
        int ww = (wdx % nWw) * 8 + (tl & 7) + shift; if (ww >= Wres) ww -= Wres;
        return (size_t)bimg * Hres * Wres + (size_t)hh * Wres + ww;
    };

    // forward-orientation projection weights (the forward's prepack): 48 VGPRs for the kernel's lifetime
    float4 wr_qkv[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) wr_qkv[i] = wqkv_p[i * 64 + lane];
    for (int i = t; i < 4096; i += 256) sm.wt[i] = wt_g[i];
    if (t < C) { sm.gam[t] = gamma[t]; sm.bet[t] = beta[t]; }
    if (t < 3 * C) sm.bq[t] = bqkv[t];
    if (t < 32) sm.top[t] = 0;
    const float4* wt4 = reinterpret_cast<const float4*>(sm.wt) + lane;

    // persistent gradient accumulators
    f32x4 accW[3] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    f32x4 accO = {0.f, 0.f, 0.f, 0.f};
    float dgam[8], dbet[8], accr[16], dbf = 0.f;          // dbf: thread t < 96: column t of d[Q|K|V]; 96 <= t < 128: column of dy
#pragma unroll
    for (int i = 0; i < 8; ++i) { dgam[i] = 0.f; dbet[i] = 0.f; }
#pragma unroll
    for (int i = 0; i < 16; ++i) accr[i] = 0.f;

    float4 xv[2], dv4[2], xnext[2], dnext[2];
    int win = blockIdx.x;
    if (win < nwin) {
        const size_t st = src_token(win);
        const float4* xp = reinterpret_cast<const float4*>(x + st * C + 8 * g);
        const float4* dp = reinterpret_cast<const float4*>(dout + st * C + 8 * g);
        xv[0] = xp[0]; xv[1] = xp[1]; dv4[0] = dp[0]; dv4[1] = dp[1];
    }
    __syncthreads();
#pragma unroll 1
    for (; win < nwin; win += gridDim.x) {
        const int bimg = win / nW, wdx = win % nW;
        const size_t src_tok = src_token(win);
        // ---- A. LayerNorm recompute (4 lanes per token: lane bits 4, 5), dy = s dout
        float xr[8] = {xv[0].x, xv[0].y, xv[0].z, xv[0].w, xv[1].x, xv[1].y, xv[1].z, xv[1].w};
        float dr[8] = {dv4[0].x, dv4[0].y, dv4[0].z, dv4[0].w, dv4[1].x, dv4[1].y, dv4[1].z, dv4[1].w};
        float xh[8], dya[8];
        float rstd;
        {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) s += xr[i];
            const float mean = tok4sum(s) * (1.0f / C);
            float var = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) { const float a = xr[i] - mean; var += a * a; }
            rstd = rsqrtf(tok4sum(var) * (1.0f / C) + 1e-5f);
            const float sc = dscale ? dscale[bimg] : 1.0f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                xh[i] = (xr[i] - mean) * rstd;
                dya[i] = sc * dr[i];
            }
        }
        __syncthreads();                                  // S0: the previous window's reads of every region are done
        if (t < NT) {
            const uint8_t r = rank_in[(size_t)win * NT + t];
            sm.rank[t] = r;
            if (r < NU) sm.top[r] = t;
        }
        for (int e = t; e < (32 - NU) * DS; e += 256) { sm.qr[NU * DS + e] = 0.f; sm.dor[NU * DS + e] = 0.f; }
        // prefetch the next window's rows (nothing of this window waits for a global load behind this point except bias / mask)
        if (win + (int)gridDim.x < nwin) {
            const size_t st = src_token(win + gridDim.x);
            const float4* xp = reinterpret_cast<const float4*>(x + st * C + 8 * g);
            const float4* dp = reinterpret_cast<const float4*>(dout + st * C + 8 * g);
            xnext[0] = xp[0]; xnext[1] = xp[1]; dnext[0] = dp[0]; dnext[1] = dp[1];
        }
        __syncthreads();                                  // S1: ranks visible
        // ---- B. Q, K, V recompute (the forward's instruction sequence: bit-identical values) -> K, V rows, Q[top] rows;
        //         dctx = dy Wo -> dO[top] rows + the sum over the unselected queries
        {
            float xa[8];                                  // LayerNorm output = this lane's A fragment (recomputed where needed: registers)
#pragma unroll
            for (int i = 0; i < 8; ++i) xa[i] = xh[i] * sm.gam[8 * g + i] + sm.bet[8 * g + i];
            f32x4 acc[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const float bj = sm.bq[(j >> 1) * C + 16 * (j & 1) + i16];
                acc[j] = f32x4{bj, bj, bj, bj};
            }
#pragma unroll
            for (int s4 = 0; s4 < 2; ++s4) {
#pragma unroll
                for (int j = 0; j < 6; ++j) acc[j] = mfma16(xa[4 * s4 + 0], wr_qkv[j * 2 + s4].x, acc[j]);
#pragma unroll
                for (int j = 0; j < 6; ++j) acc[j] = mfma16(xa[4 * s4 + 1], wr_qkv[j * 2 + s4].y, acc[j]);
#pragma unroll
                for (int j = 0; j < 6; ++j) acc[j] = mfma16(xa[4 * s4 + 2], wr_qkv[j * 2 + s4].z, acc[j]);
#pragma unroll
                for (int j = 0; j < 6; ++j) acc[j] = mfma16(xa[4 * s4 + 3], wr_qkv[j * 2 + s4].w, acc[j]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * w + 4 * g + r;
                const int rr = sm.rank[row];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    sm.k[row * DS + 16 * j + i16] = acc[2 + j][r];
                    sm.v[row * DS + 16 * j + i16] = acc[4 + j][r];
                }
                if (rr < NU) { sm.qr[rr * DS + i16] = acc[0][r]; sm.qr[rr * DS + 16 + i16] = acc[1][r]; }
            }
        }
        {
            f32x4 dacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int s4 = 0; s4 < 2; ++s4) {
                const float4 b0 = wt4[(12 + 0 * 2 + s4) * 64], b1 = wt4[(12 + 1 * 2 + s4) * 64];
                dacc[0] = mfma16(dya[4 * s4 + 0], b0.x, dacc[0]); dacc[1] = mfma16(dya[4 * s4 + 0], b1.x, dacc[1]);
                dacc[0] = mfma16(dya[4 * s4 + 1], b0.y, dacc[0]); dacc[1] = mfma16(dya[4 * s4 + 1], b1.y, dacc[1]);
                dacc[0] = mfma16(dya[4 * s4 + 2], b0.z, dacc[0]); dacc[1] = mfma16(dya[4 * s4 + 2], b1.z, dacc[1]);
                dacc[0] = mfma16(dya[4 * s4 + 3], b0.w, dacc[0]); dacc[1] = mfma16(dya[4 * s4 + 3], b1.w, dacc[1]);
            }
            float us0 = 0.f, us1 = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rr = sm.rank[16 * w + 4 * g + r];
                if (rr < NU) { sm.dor[rr * DS + i16] = dacc[0][r]; sm.dor[rr * DS + 16 + i16] = dacc[1][r]; }
                else { us0 += dacc[0][r]; us1 += dacc[1][r]; }
            }
            us0 = tok4sum(us0); us1 = tok4sum(us1);
            if (g == 0) { atomicAdd(&sm.dor[NU * DS + i16], us0); atomicAdd(&sm.dor[NU * DS + 16 + i16], us1); }
        }
        __syncthreads();                                  // S2
        // ---- C. scores of the selected rows: Sr = Q[top] K^T (32 x 64), two tiles per wave
        {
            const int tr = w & 1;
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
                const int tc = (w >> 1) + 2 * ii;
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                acc = tile_mma<8>(sm.qr + 16 * tr * DS, DS, 1, sm.k + 16 * tc * DS, DS, 1, acc);
#pragma unroll
                for (int j = 0; j < 4; ++j) sm.p1[(16 * tr + 4 * g + j) * SS + 16 * tc + i16] = acc[j];
            }
        }
        __syncthreads();                                  // S3
        {
            const int r = t >> 3, c0 = (t & 7) * 8;
            float p1[8], p2[8];
            if (r < NU) {
                const int qrow = sm.top[r];
                float xs[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) xs[i] = sm.p1[r * SS + c0 + i] * scale;
                const float* brow = bias ? bias + (size_t)qrow * NT + c0 : nullptr;
                const float* mrow = mask ? mask + ((size_t)wdx * NT + qrow) * NT + c0 : nullptr;
                double_softmax8(xs, brow, mrow, p1, p2);
            } else {
                const float f = (r == NU) ? (1.0f / NT) : 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) { p1[i] = 0.f; p2[i] = f; }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) { sm.p1[r * SS + c0 + i] = p1[i]; sm.p2[r * SS + c0 + i] = p2[i]; }
        }
        __syncthreads();                                  // S4
        // ---- D. dV = P2^T dO[top] (row 25 of P2 = 1/64: the mean(V) path), dP2 = dO[top] V^T, O = P2 V
        {
            f32x4 accv[2], accp[2], acco = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int tc = 0; tc < 2; ++tc) {
                f32x4 z = {0.f, 0.f, 0.f, 0.f};
                accv[tc] = tile_mma<8>(sm.p2 + 16 * w, 1, SS, sm.dor + 16 * tc, 1, DS, z);
            }
            const int tr = w & 1;
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
                const int tc = (w >> 1) + 2 * ii;
                f32x4 z = {0.f, 0.f, 0.f, 0.f};
                accp[ii] = tile_mma<8>(sm.dor + 16 * tr * DS, DS, 1, sm.v + 16 * tc * DS, DS, 1, z);
            }
            // O tile (rows 16 tr.., columns 16 (w >> 1)..): A(i = r, k = n) = P2[r][n], B(k = n, j = e) = V[n][e]
            acco = tile_mma<16>(sm.p2 + 16 * tr * SS, SS, 1, sm.v + 16 * (w >> 1), 1, DS, acco);
            __syncthreads();                              // S5: all reads of V, dO done
#pragma unroll
            for (int tc = 0; tc < 2; ++tc)
#pragma unroll
                for (int j = 0; j < 4; ++j) sm.v[(16 * w + 4 * g + j) * DS + 16 * tc + i16] = accv[tc][j];
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
                const int tc = (w >> 1) + 2 * ii;
#pragma unroll
                for (int j = 0; j < 4; ++j) sm.dsb[(16 * tr + 4 * g + j) * SS + 16 * tc + i16] = accp[ii][j];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) sm.o[(16 * tr + 4 * g + j) * DS + 16 * (w >> 1) + i16] = acco[j];
        }
        __syncthreads();                                  // S6
        // ---- E. softmax backward (twice); dA rows parked in P2 for the bias-gradient owners
        {
            const int r = t >> 3, c0 = (t & 7) * 8;
            float dp[8], p1[8], p2[8];
            float dot2 = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                dp[i] = sm.dsb[r * SS + c0 + i]; p1[i] = sm.p1[r * SS + c0 + i]; p2[i] = sm.p2[r * SS + c0 + i];
                dot2 += dp[i] * p2[i];
            }
            dot2 = row8_sum(dot2);
            float da[8], dot1 = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) { da[i] = p2[i] * (dp[i] - dot2); dot1 += da[i] * p1[i]; }
            dot1 = row8_sum(dot1);
            if (HAS_BIAS) {
#pragma unroll
                for (int i = 0; i < 8; ++i) sm.p2[r * SS + c0 + i] = da[i];
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) sm.dsb[r * SS + c0 + i] = p1[i] * (da[i] - dot1) * scale;
        }
        __syncthreads();                                  // S7
        if (HAS_BIAS) {                                   // thread t owns (row w + 4 i, column lane) of the 64 x 64 table
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int rk = sm.rank[w + 4 * i];        // wave-uniform
                if (rk < NU) accr[i] += sm.p2[rk * SS + lane];
            }
        }
        // ---- dQ[top] = dS K (32 x 32, one tile per wave), dK = dS^T Q[top] (64 x 32, two tiles per wave)
        {
            const int tr = w & 1, tcq = w >> 1;
            f32x4 accq = {0.f, 0.f, 0.f, 0.f}, acck[2];
            accq = tile_mma<16>(sm.dsb + 16 * tr * SS, SS, 1, sm.k + 16 * tcq, 1, DS, accq);
#pragma unroll
            for (int tc = 0; tc < 2; ++tc) {
                f32x4 z = {0.f, 0.f, 0.f, 0.f};
                acck[tc] = tile_mma<8>(sm.dsb + 16 * w, 1, SS, sm.qr + 16 * tc, 1, DS, z);
            }
            __syncthreads();                              // S8: all reads of K, Q[top], P1, P2, dS done
            float* dqs = sm.p1;                           // 32 x DS
#pragma unroll
            for (int j = 0; j < 4; ++j) dqs[(16 * tr + 4 * g + j) * DS + 16 * tcq + i16] = accq[j];
#pragma unroll
            for (int tc = 0; tc < 2; ++tc)
#pragma unroll
                for (int j = 0; j < 4; ++j) sm.k[(16 * w + 4 * g + j) * DS + 16 * tc + i16] = acck[tc][j];
        }
        // ---- F. xn rows and dy rows into the dead regions (row layout: this lane's token, its 8 channels)
        {
            float xa[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) xa[i] = xh[i] * sm.gam[8 * g + i] + sm.bet[8 * g + i];
            *reinterpret_cast<float4*>(&xn_rows[tl * DS + 8 * g]) = make_float4(xa[0], xa[1], xa[2], xa[3]);
            *reinterpret_cast<float4*>(&xn_rows[tl * DS + 8 * g + 4]) = make_float4(xa[4], xa[5], xa[6], xa[7]);
        }
        *reinterpret_cast<float4*>(&dy_rows[tl * DS + 8 * g]) = make_float4(dya[0], dya[1], dya[2], dya[3]);
        *reinterpret_cast<float4*>(&dy_rows[tl * DS + 8 * g + 4]) = make_float4(dya[4], dya[5], dya[6], dya[7]);
        __syncthreads();                                  // S9
        // ---- G. weight gradients (contraction over the 64 tokens) and dxn
        f32x4 dxacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        {
            const float* dqs = sm.p1;
            // dW[m][f][c] += sum_tok d_m[tok][f] xn[tok][c]: 12 tiles (6 feature tiles x 2 column tiles), three per wave
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int idx = 3 * w + i, ft = idx >> 1, tn = idx & 1, m = ft >> 1, f0 = 16 * (ft & 1);
                f32x4 a = accW[i];
                if (m == 0) {
#pragma unroll 4
                    for (int s = 0; s < 16; ++s) {
                        const int tok = 4 * s + g;
                        const int rr = sm.rank[tok];
                        const float af = rr < NU ? dqs[rr * DS + f0 + i16] : 0.f;
                        a = mfma16(af, xn_rows[tok * DS + 16 * tn + i16], a);
                    }
                } else {
                    const float* src = (m == 1 ? sm.k : sm.v) + f0 + i16;
#pragma unroll 4
                    for (int s = 0; s < 16; ++s) {
                        const int tok = 4 * s + g;
                        a = mfma16(src[tok * DS], xn_rows[tok * DS + 16 * tn + i16], a);
                    }
                }
                accW[i] = a;
            }
            // dWo[o][c] += sum_tok dy[tok][o] ctx[tok][c], ctx[tok] = O[rank or 25]: one tile per wave
            {
                const int ot = w >> 1, tn = w & 1;
#pragma unroll 4
                for (int s = 0; s < 16; ++s) {
                    const int tok = 4 * s + g;
                    const int rr = sm.rank[tok];
                    const int rm = rr < NU ? rr : NU;
                    accO = mfma16(dy_rows[tok * DS + 16 * ot + i16], sm.o[rm * DS + 16 * tn + i16], accO);
                }
            }
            // bias gradients of the projections: column sums of dQ (selected rows only), dK, dV
            if (t < 96) {
                const int m = t >> 5, f = t & 31;
                float s = 0.f;
                if (m == 0) {
                    for (int r = 0; r < NU; ++r) s += dqs[r * DS + f];
                } else {
                    const float* src = (m == 1 ? sm.k : sm.v) + f;
                    for (int r = 0; r < NT; ++r) s += src[r * DS];
                }
                dbf += s;
            } else if (t < 128) {                         // d(out-projection bias) = column sums of dy
                const float* src = dy_rows + (t - 96);
                float s = 0.f;
                for (int r = 0; r < NT; ++r) s += src[r * DS];
                dbf += s;
            }
            // dxn[tok][c] = sum_f dqkv[tok][f] Wqkv[f][c]: rows of this wave, two column tiles, 24 k-steps
            const int rr = sm.rank[tl];
#pragma unroll
            for (int m = 0; m < 3; ++m)
#pragma unroll
                for (int s4 = 0; s4 < 2; ++s4) {
                    float a[4];
                    if (m == 0) {
                        if (rr < NU) ld4(&dqs[rr * DS + 8 * g + 4 * s4], a);
                        else { a[0] = a[1] = a[2] = a[3] = 0.f; }
                    } else {
                        ld4(&(m == 1 ? sm.k : sm.v)[tl * DS + 8 * g + 4 * s4], a);
                    }
                    const float4 b0 = wt4[((m * 2 + 0) * 2 + s4) * 64], b1 = wt4[((m * 2 + 1) * 2 + s4) * 64];
                    dxacc[0] = mfma16(a[0], b0.x, dxacc[0]); dxacc[1] = mfma16(a[0], b1.x, dxacc[1]);
                    dxacc[0] = mfma16(a[1], b0.y, dxacc[0]); dxacc[1] = mfma16(a[1], b1.y, dxacc[1]);
                    dxacc[0] = mfma16(a[2], b0.z, dxacc[0]); dxacc[1] = mfma16(a[2], b1.z, dxacc[1]);
                    dxacc[0] = mfma16(a[3], b0.w, dxacc[0]); dxacc[1] = mfma16(a[3], b1.w, dxacc[1]);
                }
        }
        __syncthreads();                                  // S10: K, V, P1, xn rows, dy rows, O are dead
        // ---- H. dxn through the (wave-private) rows of the K region into the row layout; LayerNorm backward + shortcut
        {
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int r = 0; r < 4; ++r) sm.k[(16 * w + 4 * g + r) * DS + 16 * tn + i16] = dxacc[tn][r];
            float dn[8];
            ld4(&sm.k[tl * DS + 8 * g], dn);
            ld4(&sm.k[tl * DS + 8 * g + 4], dn + 4);
            float dh[8], s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                dh[i] = dn[i] * sm.gam[8 * g + i];
                s1 += dh[i]; s2 += dh[i] * xh[i];
                dgam[i] += dn[i] * xh[i];
                dbet[i] += dn[i];
            }
            s1 = tok4sum(s1) * (1.0f / C);
            s2 = tok4sum(s2) * (1.0f / C);
            float o8[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) o8[i] = dr[i] + rstd * (dh[i] - s1 - xh[i] * s2);
            float4* op = reinterpret_cast<float4*>(out.dx + src_tok * C + 8 * g);
            op[0] = make_float4(o8[0], o8[1], o8[2], o8[3]);
            op[1] = make_float4(o8[4], o8[5], o8[6], o8[7]);
        }
        xv[0] = xnext[0]; xv[1] = xnext[1]; dv4[0] = dnext[0]; dv4[1] = dnext[1];
    }

    // ---- flush the per-workgroup accumulators: one atomic per element
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int idx = 3 * w + i, ft = idx >> 1, tn = idx & 1, m = ft >> 1, f0 = 16 * (ft & 1);
#pragma unroll
        for (int j = 0; j < 4; ++j) atomicAdd(out.dw[m] + (f0 + 4 * g + j) * C + 16 * tn + i16, accW[i][j]);
    }
    {
        const int ot = w >> 1, tn = w & 1;
#pragma unroll
        for (int j = 0; j < 4; ++j) atomicAdd(out.dwo + (16 * ot + 4 * g + j) * C + 16 * tn + i16, accO[j]);
    }
    if (t < 96 && out.db[t >> 5]) atomicAdd(out.db[t >> 5] + (t & 31), dbf);
    if (t >= 96 && t < 128 && out.dbo) atomicAdd(out.dbo + (t - 96), dbf);
    __syncthreads();
    if (t < 2 * C) sm.red[t] = 0.f;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        // fold the 16 token lanes (i16) that share a channel octet, then one LDS atomic per (wave, g, channel)
        float a = dgam[i], b = dbet[i];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
        if (i16 == 0) {
            atomicAdd(&sm.red[8 * g + i], a);
            atomicAdd(&sm.red[C + 8 * g + i], b);
        }
    }
    __syncthreads();
    if (t < C) {
        atomicAdd(out.dgamma + t, sm.red[t]);
        atomicAdd(out.dbeta + t, sm.red[C + t]);
    }
    if (HAS_BIAS) {
        float* dst = out.dbias_part + (size_t)blockIdx.x * NT * NT;
#pragma unroll
        for (int i = 0; i < 16; ++i) dst[(w + 4 * i) * NT + lane] = accr[i];
    }
}

}  // namespace

extern "C" int dhz_fused_attn_bwd_parts(int nwin) {
    const int cap = 2 * dhz_num_cus();
    return nwin < cap ? nwin : cap;
}

extern "C" int dhz_fused_attn_bwd_prepack(const float* wq, const float* wk, const float* wv, const float* wo, float* wt, int C_,
                                          void* stream) {
    DHZ_REQUIRE(wq && wk && wv && wo && wt, "dhz_fused_attn_bwd_prepack: null pointer");
    DHZ_REQUIRE(C_ == 32, "dhz_fused_attn_bwd_prepack: C=%d unsupported (32)", C_);
    hipLaunchKernelGGL(prepack_bwd_kernel, dim3(16), dim3(256), 0, (hipStream_t)stream, wq, wk, wv, wo, wt);
    DHZ_CHECK_LAUNCH("dhz_fused_attn_bwd_prepack");
    return DHZ_OK;
}

extern "C" int dhz_fused_window_attn_bwd(const float* x, const float* dout, const float* gamma, const float* beta,
                                         const float* wqkv_p, const float* bqkv, const float* wt, const float* bias,
                                         const float* mask, const float* drop_scale, const uint8_t* rank, float* dx, float* dwq,
                                         float* dwk, float* dwv, float* dbq, float* dbk, float* dbv, float* dwo, float* dbo,
                                         float* dgamma, float* dbeta, float* dbias_part, int B, int Hres, int Wres, int C_,
                                         int shift, void* stream) {
    DHZ_REQUIRE(x && dout && gamma && beta && wqkv_p && bqkv && wt && rank && dx && dwq && dwk && dwv && dwo && dgamma && dbeta,
                "dhz_fused_window_attn_bwd: null pointer");
    DHZ_REQUIRE(C_ == 32, "dhz_fused_window_attn_bwd: C=%d unsupported (32)", C_);
    DHZ_REQUIRE(B > 0 && Hres % 8 == 0 && Wres % 8 == 0 && Hres >= 8 && Wres >= 8 && shift >= 0 && shift < 8,
                "dhz_fused_window_attn_bwd: bad geometry %dx%d shift %d", Hres, Wres, shift);
    DHZ_REQUIRE((bias != nullptr) == (dbias_part != nullptr), "dhz_fused_window_attn_bwd: bias and dbias_part go together");
    DHZ_REQUIRE(!mask || shift > 0, "dhz_fused_window_attn_bwd: a mask is only meaningful for shifted windows");
    const int nwin = B * (Hres / 8) * (Wres / 8);
    const int grid = dhz_fused_attn_bwd_parts(nwin);
    BwdOut o;
    o.dx = dx; o.dw[0] = dwq; o.dw[1] = dwk; o.dw[2] = dwv; o.db[0] = dbq; o.db[1] = dbk; o.db[2] = dbv;
    o.dwo = dwo; o.dbo = dbo; o.dgamma = dgamma; o.dbeta = dbeta; o.dbias_part = dbias_part;
    const size_t smem = sizeof(BwdSmem);
    hipStream_t s = (hipStream_t)stream;
    if (bias) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fused_window_attn_bwd_c32_kernel<true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(fused_window_attn_bwd_c32_kernel<true>, dim3(grid), dim3(256), smem, s, x, dout, gamma, beta,
                           reinterpret_cast<const float4*>(wqkv_p), bqkv, wt, bias, mask, drop_scale, rank, o, Hres, Wres, shift, nwin);
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fused_window_attn_bwd_c32_kernel<false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(fused_window_attn_bwd_c32_kernel<false>, dim3(grid), dim3(256), smem, s, x, dout, gamma, beta,
                           reinterpret_cast<const float4*>(wqkv_p), bqkv, wt, bias, mask, drop_scale, rank, o, Hres, Wres, shift, nwin);
    }
    DHZ_CHECK_LAUNCH("dhz_fused_window_attn_bwd");
    return DHZ_OK;
}
