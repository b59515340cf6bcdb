// HBM-bound kernels around the window attention: K1 (LN + shift + partition), K4-tail (reverse +
// un-shift + residual), K5-middle (LeFF depthwise 3x3 with both GELUs, token layout), K10 (Charbonnier)
// and K12 (AdamW).  All are single-pass, float4-vectorised, with the roll / partition / reverse
// permutations folded into address arithmetic (never materialised).
#include <math.h>
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------ K1
// first column of quad v of lane li: quads interleaved over the lanes, or (WIDE: bf16 storage) PAIRS of adjacent quads per lane -
// 16 bytes of bf16 per lane and access as in fp32, and half the lanes (hence half the per-lane index / shuffle work) per token:
// at C = 64 the bf16 kernels were instruction-bound at 2.8 TB/s where the fp32 ones stream at 5.9
template <bool WIDE>
__device__ __forceinline__ int quad_col(int li, int v, int lpt) {
    return WIDE ? 8 * (li + (v >> 1) * lpt) + 4 * (v & 1) : 4 * (li + v * lpt);
}

// One token = C floats.  LPT lanes cooperate on a token (C/4 float4, up to 4 per lane).
template <int VPL, typename T, bool WIDE = false>   // float4 per lane; T = storage type of x / xw
__global__ __launch_bounds__(256) void ln_partition_fwd_kernel(const T* __restrict__ x,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, T* __restrict__ xw,
                                                               float* __restrict__ stats, int ntok, int Hres, int Wres,
                                                               int C, int shift, int lpt, int partition) {
    const int tpw = 64 / lpt;                                  // tokens per wave
    const int lane = threadIdx.x & 63;
    const int sub = lane / lpt, li = lane % lpt;
    const int wave_global = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    const int HW = Hres * Wres;
    const float invC = 1.0f / (float)C;
    float4 gm[VPL], bt[VPL];
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
        gm[v] = *reinterpret_cast<const float4*>(gamma + quad_col<WIDE>(li, v, lpt));
        bt[v] = *reinterpret_cast<const float4*>(beta + quad_col<WIDE>(li, v, lpt));
    }
    for (int base = wave_global * tpw; base < ntok; base += nwaves * tpw) {
        const int tok = base + sub;
        const bool ok = tok < ntok;
        float4 xv[VPL];
        float s = 0.f;
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
            xv[v] = ok ? ld4(x + (size_t)tok * C + quad_col<WIDE>(li, v, lpt)) : make_float4(0, 0, 0, 0);
            s += xv[v].x + xv[v].y + xv[v].z + xv[v].w;
        }
        for (int o = 1; o < lpt; o <<= 1) s += __shfl_xor(s, o);
        const float mean = s * invC;
        float var = 0.f;
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
            const float a = xv[v].x - mean, b = xv[v].y - mean, c = xv[v].z - mean, d = xv[v].w - mean;
            var += a * a + b * b + c * c + d * d;
        }
        for (int o = 1; o < lpt; o <<= 1) var += __shfl_xor(var, o);
        const float rstd = rsqrtf(var * invC + 1e-5f);
        if (ok) {
            const int bimg = tok / HW, p = tok % HW;
            const size_t dst = partition ? (size_t)bimg * HW + window_slot(p / Wres, p % Wres, Hres, Wres, shift) : (size_t)tok;
#pragma unroll
            for (int v = 0; v < VPL; ++v) {
                float4 y;
                y.x = (xv[v].x - mean) * rstd * gm[v].x + bt[v].x;
                y.y = (xv[v].y - mean) * rstd * gm[v].y + bt[v].y;
                y.z = (xv[v].z - mean) * rstd * gm[v].z + bt[v].z;
                y.w = (xv[v].w - mean) * rstd * gm[v].w + bt[v].w;
                st4(xw + dst * C + quad_col<WIDE>(li, v, lpt), y);
            }
            if (li == 0 && stats) *reinterpret_cast<float2*>(stats + 2 * (size_t)tok) = make_float2(mean, rstd);
        }
    }
}

template <int VPL, typename T, bool WIDE = false>
__global__ __launch_bounds__(256) void ln_partition_bwd_kernel(const T* __restrict__ dxw,
                                                               const T* __restrict__ x,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ stats,
                                                               const T* dres, T* dx,
                                                               float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                               int ntok, int Hres, int Wres, int C, int shift, int lpt,
                                                               int partition, int lay, T* __restrict__ dx2,
                                                               const float* __restrict__ scale2) {
    // lay (dhz_ln_partition_bwd_lay): bit 0 - dres lies in the window order of (shift, partition) like dxw; bit 1 - dx is WRITTEN in
    // the window order of shift (lay >> 8) on the same map: the layout the consumer (the attention branch's backward) reads.
    // dx2 (optional): a SECOND copy of dx in that window order, times scale2[image] - the scaled, window-ordered d(out) operand of the
    // out-projection's backward products where their kernels take no row factor (bf16 storage)
    __shared__ float red[2 * 1024];                            // dgamma | dbeta  (C <= 1024)
    const int tpw = 64 / lpt;
    const int lane = threadIdx.x & 63;
    const int sub = lane / lpt, li = lane % lpt;
    const int wave_global = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    const int HW = Hres * Wres;
    const float invC = 1.0f / (float)C;
    for (int e = threadIdx.x; e < 2 * C; e += blockDim.x) red[e] = 0.f;
    __syncthreads();
    float4 gm[VPL], dg[VPL], db[VPL];
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
        gm[v] = *reinterpret_cast<const float4*>(gamma + quad_col<WIDE>(li, v, lpt));
        dg[v] = make_float4(0, 0, 0, 0);
        db[v] = make_float4(0, 0, 0, 0);
    }
    // Two token groups per trip, all loads of both issued before the first use: with the grid capped for the dgamma/dbeta
    // atomics (<= 2 workgroups per CU) a single group left only ~24 KB per CU in flight.
    struct Group {
        float4 xh[VPL], dy[VPL], rs[VPL];
        float mean, rstd, sc2;
        int tok, drow, drow2;
        bool ok;
    };
    auto load = [&](int base, Group& G) {
        G.tok = base + sub;
        G.ok = G.tok < ntok;
        G.mean = 0.f; G.rstd = 0.f;
        if (G.ok) {
            const int bimg = G.tok / HW, p = G.tok % HW;
            const int hh = p / Wres, ww = p - hh * Wres;
            const size_t src = partition ? (size_t)bimg * HW + window_slot(hh, ww, Hres, Wres, shift) : (size_t)G.tok;
            const size_t rsrc = (lay & 1) ? src : (size_t)G.tok;
            G.drow = (lay & 2) ? bimg * HW + window_slot(hh, ww, Hres, Wres, lay >> 8) : G.tok;
            if (dx2) {
                G.drow2 = bimg * HW + window_slot(hh, ww, Hres, Wres, lay >> 8);
                G.sc2 = scale2 ? scale2[bimg] : 1.f;
            }
#pragma unroll
            for (int v = 0; v < VPL; ++v) {
                G.xh[v] = ld4(x + (size_t)G.tok * C + quad_col<WIDE>(li, v, lpt));
                G.dy[v] = ld4(dxw + src * C + quad_col<WIDE>(li, v, lpt));
                if (dres) G.rs[v] = ld4(dres + rsrc * C + quad_col<WIDE>(li, v, lpt));
            }
            const float2 st = *reinterpret_cast<const float2*>(stats + 2 * (size_t)G.tok);
            G.mean = st.x; G.rstd = st.y;
        }
    };
    auto compute = [&](Group& G) {
        float s1 = 0.f, s2 = 0.f;
        const float mean = G.mean, rstd = G.rstd;
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
            float4 xn = make_float4(0, 0, 0, 0), d = make_float4(0, 0, 0, 0);
            if (G.ok) {
                const float4 xv = G.xh[v];
                xn = make_float4((xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd, (xv.w - mean) * rstd);
                d = G.dy[v];
            }
            dg[v].x += d.x * xn.x; dg[v].y += d.y * xn.y; dg[v].z += d.z * xn.z; dg[v].w += d.w * xn.w;
            db[v].x += d.x; db[v].y += d.y; db[v].z += d.z; db[v].w += d.w;
            // dxhat = dy * gamma
            d.x *= gm[v].x; d.y *= gm[v].y; d.z *= gm[v].z; d.w *= gm[v].w;
            s1 += d.x + d.y + d.z + d.w;
            s2 += d.x * xn.x + d.y * xn.y + d.z * xn.z + d.w * xn.w;
            G.xh[v] = xn; G.dy[v] = d;
        }
        for (int o = 1; o < lpt; o <<= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
        s1 *= invC; s2 *= invC;
        if (G.ok) {
#pragma unroll
            for (int v = 0; v < VPL; ++v) {
                float4 r;
                r.x = rstd * (G.dy[v].x - s1 - G.xh[v].x * s2);
                r.y = rstd * (G.dy[v].y - s1 - G.xh[v].y * s2);
                r.z = rstd * (G.dy[v].z - s1 - G.xh[v].z * s2);
                r.w = rstd * (G.dy[v].w - s1 - G.xh[v].w * s2);
                if (dres) { r.x += G.rs[v].x; r.y += G.rs[v].y; r.z += G.rs[v].z; r.w += G.rs[v].w; }
                st4(dx + (size_t)G.drow * C + quad_col<WIDE>(li, v, lpt), r);
                if (dx2) st4(dx2 + (size_t)G.drow2 * C + quad_col<WIDE>(li, v, lpt), make_float4(G.sc2 * r.x, G.sc2 * r.y, G.sc2 * r.z, G.sc2 * r.w));
            }
        }
    };
    const int stride = nwaves * tpw;
    // (bf16 storage halves the bytes per load: four groups per trip keep the same bytes in flight; VPL <= 2 there keeps the
    // register budget)
    constexpr int NG = (sizeof(T) == 2 && VPL <= 2) ? 4 : 2;
    for (int base = wave_global * tpw; base < ntok; base += NG * stride) {
        Group G[NG];
#pragma unroll
        for (int i = 0; i < NG; ++i) load(base + i * stride, G[i]);          // past the end: ok = false, nothing loaded or stored
#pragma unroll
        for (int i = 0; i < NG; ++i) compute(G[i]);
    }
    // reduce dgamma/dbeta: lanes with equal li inside the wave, then waves through LDS, then one atomic per channel
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
        for (int o = lpt; o < 64; o <<= 1) {
            dg[v].x += __shfl_xor(dg[v].x, o); dg[v].y += __shfl_xor(dg[v].y, o);
            dg[v].z += __shfl_xor(dg[v].z, o); dg[v].w += __shfl_xor(dg[v].w, o);
            db[v].x += __shfl_xor(db[v].x, o); db[v].y += __shfl_xor(db[v].y, o);
            db[v].z += __shfl_xor(db[v].z, o); db[v].w += __shfl_xor(db[v].w, o);
        }
        if (sub == 0) {
            const int c = quad_col<WIDE>(li, v, lpt);
            atomicAdd(&red[c + 0], dg[v].x); atomicAdd(&red[c + 1], dg[v].y);
            atomicAdd(&red[c + 2], dg[v].z); atomicAdd(&red[c + 3], dg[v].w);
            atomicAdd(&red[C + c + 0], db[v].x); atomicAdd(&red[C + c + 1], db[v].y);
            atomicAdd(&red[C + c + 2], db[v].z); atomicAdd(&red[C + c + 3], db[v].w);
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < C; e += blockDim.x) {
        atomicAdd(dgamma + e, red[e]);
        atomicAdd(dbeta + e, red[C + e]);
    }
}

// ------------------------------------------------------------------------------------------------ K4 tail
template <bool BWD, typename T>
__global__ __launch_bounds__(256) void reverse_residual_kernel(const T* __restrict__ a,        // yw (fwd) / dout (bwd)
                                                               const T* __restrict__ shortcut,
                                                               const float* __restrict__ scale, T* __restrict__ o,
                                                               int ntok, int Hres, int Wres, int C4, int shift, int partition) {
    const int HW = Hres * Wres;
    const size_t total = (size_t)ntok * C4;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int tok = (int)(e / C4), c = (int)(e % C4);
        const int bimg = tok / HW, p = tok % HW;
        const size_t slot = partition ? (size_t)bimg * HW + window_slot(p / Wres, p % Wres, Hres, Wres, shift) : (size_t)tok;
        const float sc = scale ? scale[bimg] : 1.0f;
        if (!BWD) {
            const float4 y = ld4(a + 4 * (slot * C4 + c));
            const float4 s = ld4(shortcut + 4 * e);
            st4(o + 4 * e, make_float4(s.x + sc * y.x, s.y + sc * y.y, s.z + sc * y.z, s.w + sc * y.w));
        } else {
            const float4 g = ld4(a + 4 * e);
            st4(o + 4 * (slot * C4 + c), make_float4(sc * g.x, sc * g.y, sc * g.z, sc * g.w));
        }
    }
}

// ------------------------------------------------------------------------------------------------ K5 middle
#ifndef DWB_WAVES
#define DWB_WAVES 3        // waves per SIMD the backward kernel is compiled for (3 workgroups per CU)
#endif
#ifndef DW_ABL
#define DW_ABL 0        // timing diagnostics of leff_dwconv_bwd (tools/variants.sh): 2 no dw accumulation, 4 no u load, 8 no GELU
#endif
constexpr int TW = 16, TH = 8;                 // spatial tile (positions)
constexpr int HWID = TW + 2, HHGT = TH + 2;    // with halo
constexpr int CT = 32;                         // channels per workgroup of the 8-lane form (128 B of fp32 per position)
// LPP = lanes per position (4 channels each): a workgroup is 32 position slots x LPP lanes and covers 4 LPP channels.  Measured on one
// box (tools/bench_dwconv.py, sums over the step's shapes): forward LPP 8 -> 16: fp32 846 -> 813 us, bf16 465 -> 447 us (whole 128-byte
// lines per position in bf16, 256 B in fp32); backward LPP 8 -> 16: fp32 1211 -> 1513 us, bf16 647 -> 707 us (78 KB of LDS: two
// workgroups of 512 threads per CU, 128 VGPRs with spills).  The forward runs 16 where Ch allows, the backward 8.
// The bf16 backward moves 2.7 - 3.1 TB/s: per tile and wave ~1250 VALU issue slots (a quarter of them the 64-bit address arithmetic of
// the staging loads) - issuing the next tile's loads as raw bf16 pairs before the position loop (32 more registers) measured 640 ->
// 670 us at two workgroups per CU and 970 us with the spills of three: it is the issue slots, not the load latency.
#ifndef DW_FWD_LPP
#define DW_FWD_LPP 16
#endif
#ifndef DW_BWD_LPP
#define DW_BWD_LPP 8
#endif

// LDS tile: [HHGT][HWID][CT] floats
template <typename T, int LPP>
__global__ __launch_bounds__(32 * LPP) void leff_dwconv_fwd_kernel(const T* __restrict__ u, const float* __restrict__ w,
                                                              const float* __restrict__ bconv, T* __restrict__ tpre,
                                                              T* __restrict__ z, int Hres, int Wres, int Ch,
                                                              int tiles_x, int tiles_y) {
    constexpr int CT = 4 * LPP;
    __shared__ __attribute__((aligned(16))) float g[HHGT * HWID * CT];
    const int t = threadIdx.x;
    // workgroups are dealt round-robin to the 8 XCDs: renumber so that the channel groups of one tile - the 128-byte pieces of
    // the same token rows - run on ONE XCD (its L2 then sees whole rows): 4.6 -> 5.1 TB/s.  (The persistent backward kernel
    // measured 5 % SLOWER with the same renumbering and keeps the plain order.)
    const int lid = (gridDim.x & 7) == 0 ? (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    const int cg = lid % (Ch / CT);
    int rest = lid / (Ch / CT);
    const int tx = rest % tiles_x; rest /= tiles_x;
    const int ty = rest % tiles_y;
    const int bimg = rest / tiles_y;
    const int c4 = t % LPP, ch0 = cg * CT + c4 * 4;
    const int x0 = tx * TW - 1, y0 = ty * TH - 1;
    const T* ub = u + (size_t)bimg * Hres * Wres * Ch;
    // stage gelu(u) with a 1-pixel halo (zero outside the image: Conv2d padding=1)
    {
        constexpr int NPOS = HHGT * HWID, NIT = (NPOS + 31) / 32;
        float4 ru[NIT];
        bool ok[NIT];
#pragma unroll
        for (int i = 0; i < NIT; ++i) {                       // all loads in flight before the first GELU
            const int pos = (t / LPP) + 32 * i;
            const int pc = pos < NPOS ? pos : NPOS - 1;
            const int yy = y0 + pc / HWID, xx = x0 + pc % HWID;
            ok[i] = pos < NPOS && yy >= 0 && yy < Hres && xx >= 0 && xx < Wres;
            const int yc = min(max(yy, 0), Hres - 1), xc = min(max(xx, 0), Wres - 1);
            ru[i] = ld4(ub + ((size_t)yc * Wres + xc) * Ch + ch0);
        }
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int pos = (t / LPP) + 32 * i;
            if (pos < NPOS) {
                float4 val = make_float4(0, 0, 0, 0);
                if (ok[i]) {
                    const f32x4 gv = gelu_f4(f32x4{ru[i].x, ru[i].y, ru[i].z, ru[i].w});
                    val = make_float4(gv[0], gv[1], gv[2], gv[3]);
                }
                *reinterpret_cast<float4*>(&g[pos * CT + c4 * 4]) = val;
            }
        }
    }
    float wk[4][9];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int kk = 0; kk < 9; ++kk) wk[c][kk] = w[(ch0 + c) * 9 + kk];
    const float4 bb = *reinterpret_cast<const float4*>(bconv + ch0);
    __syncthreads();
    for (int pos = t / LPP; pos < TH * TW; pos += 32) {
        const int py = pos / TW, px = pos % TW;
        const int yy = ty * TH + py, xx = tx * TW + px;
        if (yy >= Hres || xx >= Wres) continue;
        float4 acc = bb;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const float4 gv = *reinterpret_cast<const float4*>(&g[((py + ky) * HWID + px + kx) * CT + c4 * 4]);
                acc.x += wk[0][ky * 3 + kx] * gv.x; acc.y += wk[1][ky * 3 + kx] * gv.y;
                acc.z += wk[2][ky * 3 + kx] * gv.z; acc.w += wk[3][ky * 3 + kx] * gv.w;
            }
        const size_t o = ((size_t)bimg * Hres * Wres + (size_t)yy * Wres + xx) * Ch + ch0;
        float4 zz, zp;
        {
            f32x4 zv, pv;
            gelu_both4(f32x4{acc.x, acc.y, acc.z, acc.w}, zv, pv);
            zz = make_float4(zv[0], zv[1], zv[2], zv[3]);
            zp = make_float4(pv[0], pv[1], pv[2], pv[3]);
        }
        if (tpre) st4(tpre + o, zp);      // saved for backward: gelu'(t), not t itself
        st4(z + o, zz);
    }
}

// Backward: persistent over tiles of one channel group so that dw/db are accumulated in registers and
// hit global memory with one atomic per (channel, tap) per workgroup.
//   du[p]  = gelu'(u[p]) * sum_k w[k] dt[p - off(k)]          (transpose of the forward correlation), dt = dz * gelu'(t)
//   dw[k]  = sum_q dt[q] g[q + off(k)] = sum_p g[p] dt[p - off(k)],  g = gelu(u)   (g and dt are zero outside the image)
// Written over p, BOTH sums use the same nine dt neighbours of a position and only its own u: the tile stages dt (with a
// one-pixel halo) in LDS, u is read once per position straight into registers and one GELU evaluation yields g and g'.
// (The earlier form staged gelu(u) with a halo as well: 19 LDS reads and 2.4 GELU evaluations per element, 3.6 TB/s.)
template <typename T, int LPP>
__global__ __launch_bounds__(32 * LPP, LPP == 8 ? DWB_WAVES : 4) void leff_dwconv_bwd_kernel(const T* __restrict__ dz, const T* __restrict__ u,
                                                              const T* __restrict__ tpre, const float* __restrict__ w,
                                                              T* __restrict__ du, float* __restrict__ dw,
                                                              float* __restrict__ db, const float* __restrict__ dzscale, int B,
                                                              int Hres, int Wres, int Ch, int tiles_x, int tiles_y, int wg_per_cg) {
    constexpr int CT = 4 * LPP;
    __shared__ __attribute__((aligned(16))) float ds[HHGT * HWID * CT];    // dt = dz * gelu'(t) with halo
    __shared__ __attribute__((aligned(16))) float us[TH * TW * CT];        // u of the tile: each thread parks ITS OWN loads here
    const int t = threadIdx.x;
    const int cg = blockIdx.x % (Ch / CT);
    const int wslot = blockIdx.x / (Ch / CT);
    const int c4 = t % LPP, ch0 = cg * CT + c4 * 4;
    float wk[4][9], dwk[4][9], dbk[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int kk = 0; kk < 9; ++kk) { wk[c][kk] = w[(ch0 + c) * 9 + kk]; dwk[c][kk] = 0.f; }
    const int ntiles = B * tiles_x * tiles_y;
    constexpr int NIT = TH * TW / 32;                                       // interior positions per thread
    for (int tile = wslot; tile < ntiles; tile += wg_per_cg) {
        const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, bimg = tile / (tiles_x * tiles_y);
        const int x0 = tx * TW - 1, y0 = ty * TH - 1;
        const size_t ib = (size_t)bimg * Hres * Wres;
        const float zsc = dzscale ? dzscale[bimg] : 1.f;     // per-image factor of dz (the DropPath scale, folded in here)
        __syncthreads();
        // this thread's u values (interior positions; clamped addresses, masked at the store): in flight across the staging
        float4 uv[NIT];
        if (!(DW_ABL & 4)) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int pos = (t / LPP) + 32 * it;
                const int yy = min(ty * TH + pos / TW, Hres - 1), xx = min(tx * TW + pos % TW, Wres - 1);
                uv[it] = ld4(u + (ib + (size_t)yy * Wres + xx) * Ch + ch0);
            }
        }
        // staging of dt in two batches of 3 positions per thread: all loads of a batch are issued before any is consumed
        // (branch-free, addresses clamped into the image) so that their latencies overlap
        constexpr int NPOS = HHGT * HWID;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            float4 rt[3], rz[3];
            bool ok[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int pos = (t / LPP) + 32 * (3 * half + i);
                const int pc = pos < NPOS ? pos : NPOS - 1;
                const int yy = y0 + pc / HWID, xx = x0 + pc % HWID;
                ok[i] = pos < NPOS && yy >= 0 && yy < Hres && xx >= 0 && xx < Wres;
                const int yc = min(max(yy, 0), Hres - 1), xc = min(max(xx, 0), Wres - 1);
                const size_t o = (ib + (size_t)yc * Wres + xc) * Ch + ch0;
                rt[i] = ld4(tpre + o);                                   // tpre holds gelu'(t)
                rz[i] = ld4(dz + o);
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int pos = (t / LPP) + 32 * (3 * half + i);
                if (pos < NPOS) {
                    float4 dv = make_float4(0, 0, 0, 0);
                    if (ok[i]) dv = make_float4(zsc * rz[i].x * rt[i].x, zsc * rz[i].y * rt[i].y, zsc * rz[i].z * rt[i].z, zsc * rz[i].w * rt[i].w);
                    *reinterpret_cast<float4*>(&ds[pos * CT + c4 * 4]) = dv;
                }
            }
        }
        if (!(DW_ABL & 4)) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) *reinterpret_cast<float4*>(&us[((t / LPP) + 32 * it) * CT + c4 * 4]) = uv[it];
        }
        __syncthreads();
        // the position loop is NOT unrolled (unrolled, the 36 neighbour reads of four positions are hoisted: 222 VGPRs, two
        // workgroups per CU instead of three)
#pragma unroll 1
        for (int it = 0; it < NIT; ++it) {
            const int pos = (t / LPP) + 32 * it;
            const int py = pos / TW, px = pos % TW;
            const int yy = ty * TH + py, xx = tx * TW + px;
            if (yy >= Hres || xx >= Wres) continue;
            const size_t o = (ib + (size_t)yy * Wres + xx) * Ch + ch0;
            const float4 uc = (DW_ABL & 4) ? *reinterpret_cast<const float4*>(&ds[((py + 1) * HWID + px + 1) * CT + c4 * 4])
                                           : *reinterpret_cast<const float4*>(&us[pos * CT + c4 * 4]);
            float4 gc, gp;                                               // gelu(u), gelu'(u) of this position
            if (DW_ABL & 8) { gc = uc; gp = uc; }
            else {
                f32x4 gv, pv;
                gelu_both4(f32x4{uc.x, uc.y, uc.z, uc.w}, gv, pv);
                gc = make_float4(gv[0], gv[1], gv[2], gv[3]);
                gp = make_float4(pv[0], pv[1], pv[2], pv[3]);
            }
            float4 dg = make_float4(0, 0, 0, 0);
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float4 dn = *reinterpret_cast<const float4*>(&ds[((py + 2 - ky) * HWID + px + 2 - kx) * CT + c4 * 4]);
                    dg.x += wk[0][ky * 3 + kx] * dn.x; dg.y += wk[1][ky * 3 + kx] * dn.y;
                    dg.z += wk[2][ky * 3 + kx] * dn.z; dg.w += wk[3][ky * 3 + kx] * dn.w;
                    if (ky == 1 && kx == 1) { dbk[0] += dn.x; dbk[1] += dn.y; dbk[2] += dn.z; dbk[3] += dn.w; }
                    if (DW_ABL & 2) continue;
                    dwk[0][ky * 3 + kx] += gc.x * dn.x; dwk[1][ky * 3 + kx] += gc.y * dn.y;
                    dwk[2][ky * 3 + kx] += gc.z * dn.z; dwk[3][ky * 3 + kx] += gc.w * dn.w;
                }
            st4(du + o, make_float4(dg.x * gp.x, dg.y * gp.y, dg.z * gp.z, dg.w * gp.w));
        }
    }
    // reduce the 32 position-slots (t / LPP) that share a channel quad: five quantities per round through the dead dt tile
    // ([5][32 slots][CT]: two rounds, four barriers - ten rounds of one quantity with a serial 32-term sum by 32 threads each
    // cost 7 - 11 us per launch)
    const int ps = t / LPP;
    float* red5 = ds;
    static_assert(HHGT * HWID * CT >= 5 * 32 * CT, "reduction scratch must fit the dt tile");
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            const int kk = 5 * half + q;
#pragma unroll
            for (int c = 0; c < 4; ++c) red5[(q * 32 + ps) * CT + c4 * 4 + c] = (kk < 9) ? dwk[c][kk] : dbk[c];
        }
        __syncthreads();
        if (t < 5 * CT) {
            const int q = t / CT, c = t % CT, kk = 5 * half + q;
            float sum = 0.f;
#pragma unroll
            for (int p = 0; p < 32; ++p) sum += red5[(q * 32 + p) * CT + c];
            if (kk < 9) atomicAdd(dw + (cg * CT + c) * 9 + kk, sum);
            else atomicAdd(db + cg * CT + c, sum);
        }
    }
}

// ------------------------------------------------------------------------------------------------ K10
__global__ __launch_bounds__(256) void charbonnier_fwd_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                              float* __restrict__ clampd, float* __restrict__ loss_sum,
                                                              int64_t n4, float eps2, int clamp01) {
    __shared__ float part[4];
    float s = 0.f;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (int64_t)gridDim.x * blockDim.x) {
        float4 xv = reinterpret_cast<const float4*>(x)[e];
        const float4 yv = reinterpret_cast<const float4*>(y)[e];
        if (clamp01) {
            xv.x = fminf(fmaxf(xv.x, 0.f), 1.f); xv.y = fminf(fmaxf(xv.y, 0.f), 1.f);
            xv.z = fminf(fmaxf(xv.z, 0.f), 1.f); xv.w = fminf(fmaxf(xv.w, 0.f), 1.f);
        }
        if (clampd) reinterpret_cast<float4*>(clampd)[e] = xv;
        const float a = xv.x - yv.x, b = xv.y - yv.y, c = xv.z - yv.z, d = xv.w - yv.w;
        s += sqrtf(a * a + eps2) + sqrtf(b * b + eps2) + sqrtf(c * c + eps2) + sqrtf(d * d + eps2);
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(loss_sum, part[0] + part[1] + part[2] + part[3]);
}

__global__ __launch_bounds__(256) void charbonnier_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                              const float* __restrict__ gscale, const float* __restrict__ gclamp,
                                                              float* __restrict__ dx, int64_t n4, float eps2, float inv_n, int clamp01) {
    const float gs = (gscale ? gscale[0] : 1.0f) * inv_n;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (int64_t)gridDim.x * blockDim.x) {
        const float4 xv = reinterpret_cast<const float4*>(x)[e];
        const float4 yv = reinterpret_cast<const float4*>(y)[e];
        float4 gc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gclamp) gc = reinterpret_cast<const float4*>(gclamp)[e];
        auto f = [&](float xx, float yy, float extra) {
            if (clamp01 && !(xx >= 0.f && xx <= 1.f)) return 0.f;   // torch.clamp backward: pass-through on [0,1]
            const float d = xx - yy;
            return gs * d / sqrtf(d * d + eps2) + extra;
        };
        reinterpret_cast<float4*>(dx)[e] = make_float4(f(xv.x, yv.x, gc.x), f(xv.y, yv.y, gc.y), f(xv.z, yv.z, gc.z), f(xv.w, yv.w, gc.w));
    }
}

// ------------------------------------------------------------------------------------------------ K11b
// The two L1 distances of one ContrastLoss feature tap (My_CR.py:108-112) in one pass over (a, p, n), and their joint
// backward  da = c_p sign(a - p) + c_n sign(a - n)  (c_* = upstream gradient / N, read from device memory) in another -
// instead of sub / abs / mean / sign / mul chains over feature maps of up to 134 MB each.
__global__ __launch_bounds__(256) void l1_pair_fwd_kernel(const float* __restrict__ a, const float* __restrict__ p,
                                                          const float* __restrict__ n, float* __restrict__ sums, int64_t n4) {
    __shared__ float part[2][4];
    float sp = 0.f, sn = 0.f;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (int64_t)gridDim.x * blockDim.x) {
        const float4 av = reinterpret_cast<const float4*>(a)[e];
        const float4 pv = reinterpret_cast<const float4*>(p)[e];
        sp += fabsf(av.x - pv.x) + fabsf(av.y - pv.y) + fabsf(av.z - pv.z) + fabsf(av.w - pv.w);
        if (n) {
            const float4 nv = reinterpret_cast<const float4*>(n)[e];
            sn += fabsf(av.x - nv.x) + fabsf(av.y - nv.y) + fabsf(av.z - nv.z) + fabsf(av.w - nv.w);
        }
    }
    sp = wave_sum(sp); sn = wave_sum(sn);
    if ((threadIdx.x & 63) == 0) { part[0][threadIdx.x >> 6] = sp; part[1][threadIdx.x >> 6] = sn; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(sums, part[0][0] + part[0][1] + part[0][2] + part[0][3]);
        if (n) atomicAdd(sums + 1, part[1][0] + part[1][1] + part[1][2] + part[1][3]);
    }
}

__global__ __launch_bounds__(256) void l1_pair_bwd_kernel(const float* __restrict__ a, const float* __restrict__ p,
                                                          const float* __restrict__ n, const float* __restrict__ g,
                                                          float inv_n, float* __restrict__ da, int64_t n4) {
    const float cp = g[0] * inv_n, cn = n ? g[1] * inv_n : 0.f;
    auto sgn = [](float d) { return d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f); };
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (int64_t)gridDim.x * blockDim.x) {
        const float4 av = reinterpret_cast<const float4*>(a)[e];
        const float4 pv = reinterpret_cast<const float4*>(p)[e];
        float4 r = make_float4(cp * sgn(av.x - pv.x), cp * sgn(av.y - pv.y), cp * sgn(av.z - pv.z), cp * sgn(av.w - pv.w));
        if (n) {
            const float4 nv = reinterpret_cast<const float4*>(n)[e];
            r.x += cn * sgn(av.x - nv.x); r.y += cn * sgn(av.y - nv.y); r.z += cn * sgn(av.z - nv.z); r.w += cn * sgn(av.w - nv.w);
        }
        reinterpret_cast<float4*>(da)[e] = r;
    }
}


// The scalar side of ContrastLoss.forward (My_CR.py:104-123) over the k taps, one launch each way instead of a dozen [k]-vector
// torch ops: means d_i = sums_i / count_i;  loss = sum_i w_i ap_i / (an_i + 1e-7)  (ablation: sum_i w_i ap_i), all_ap, all_an;
// backward: g[i] = d(loss g_loss + all_ap g_ap + all_an g_an) / d(ap_i, an_i).  k <= 64, one wave.
__global__ __launch_bounds__(64) void contrast_combine_fwd_kernel(const float* __restrict__ sums, const float* __restrict__ inv_cnt,
                                                                  const float* __restrict__ w, int k, int ablation,
                                                                  float* __restrict__ d, float* __restrict__ out) {
    const int i = threadIdx.x;
    float ap = 0.f, an = 0.f, term = 0.f;
    if (i < k) {
        ap = sums[2 * i] * inv_cnt[i]; an = sums[2 * i + 1] * inv_cnt[i];
        d[2 * i] = ap; d[2 * i + 1] = an;
        term = w[i] * (ablation ? ap : ap / (an + 1e-7f));
    }
    term = wave_sum(term); ap = wave_sum(ap); an = wave_sum(an);
    if (i == 0) { out[0] = term; out[1] = ap; out[2] = an; }
}

__global__ __launch_bounds__(64) void contrast_combine_bwd_kernel(const float* __restrict__ d, const float* __restrict__ w, int k,
                                                                  int ablation, const float* __restrict__ g_loss,
                                                                  const float* __restrict__ g_ap, const float* __restrict__ g_an,
                                                                  float* __restrict__ g) {
    const int i = threadIdx.x;
    if (i >= k) return;
    float g0 = 0.f, g1 = 0.f;
    if (g_loss) {
        if (ablation) g0 = g_loss[0] * w[i];
        else {
            const float den = d[2 * i + 1] + 1e-7f;
            g0 = g_loss[0] * w[i] / den;
            g1 = -g0 * d[2 * i] / den;
        }
    }
    if (g_ap) g0 += g_ap[0];
    if (g_an) g1 += g_an[0];
    g[2 * i] = g0; g[2 * i + 1] = g1;
}

// ------------------------------------------------------------------------------------------------ K12
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v,
                                                    uint16_t* __restrict__ p16, int64_t n, float lr,
                                                    float b1, float b2, float eps, float wd, float step_size,
                                                    float bc2_sqrt, float gscale) {
    const int64_t n4 = n >> 2;
    auto upd = [&](float& pp, float gg, float& mm, float& vv) {
        gg *= gscale;
        pp *= 1.0f - lr * wd;
        mm = b1 * mm + (1.0f - b1) * gg;
        vv = b2 * vv + (1.0f - b2) * gg * gg;
        const float denom = sqrtf(vv) / bc2_sqrt + eps;
        pp -= step_size * (mm / denom);
    };
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (int64_t)gridDim.x * blockDim.x) {
        float4 pv = reinterpret_cast<float4*>(p)[e], mv = reinterpret_cast<float4*>(m)[e], vv = reinterpret_cast<float4*>(v)[e];
        const float4 gv = reinterpret_cast<const float4*>(g)[e];
        upd(pv.x, gv.x, mv.x, vv.x); upd(pv.y, gv.y, mv.y, vv.y); upd(pv.z, gv.z, mv.z, vv.z); upd(pv.w, gv.w, mv.w, vv.w);
        reinterpret_cast<float4*>(p)[e] = pv; reinterpret_cast<float4*>(m)[e] = mv; reinterpret_cast<float4*>(v)[e] = vv;
        if (p16) st4(reinterpret_cast<bf16s*>(p16) + 4 * e, pv);          // the bf16 GEMMs' copy of the weights, in the same pass
    }
    // tail
    const int64_t e = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n) {
        upd(p[e], g[e], m[e], v[e]);
        if (p16) p16[e] = f32_to_bf16(p[e]);
    }
}

// Mlp's activation (M1:442-468, token_mlp = 'ffn': fc1 -> GELU -> fc2): y = gelu(u) and, for the backward pass, du = dy * gelu'(u) * scale
// with the DropPath factor of the image the token belongs to (scale[t / rows_per_scale], or 1).  Exact-erf GELU as everywhere (common.h).
template <typename T>
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const T* __restrict__ u, T* __restrict__ y, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 v = ld4(u + 4 * i);
        f32x4 g, gp;
        gelu_both4(f32x4{v.x, v.y, v.z, v.w}, g, gp);
        st4(y + 4 * i, make_float4(g[0], g[1], g[2], g[3]));
    }
}
template <typename T>
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ u, T* __restrict__ du, int64_t n4,
                                                       const float* __restrict__ scale, int64_t quads_per_scale) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 v = ld4(u + 4 * i), d = ld4(dy + 4 * i);
        f32x4 g, gp;
        gelu_both4(f32x4{v.x, v.y, v.z, v.w}, g, gp);
        const float sc = scale ? scale[i / quads_per_scale] : 1.f;
        st4(du + 4 * i, make_float4(d.x * gp[0] * sc, d.y * gp[1] * sc, d.z * gp[2] * sc, d.w * gp[3] * sc));
    }
}

inline int grid_for(int64_t work_items, int per_block = 256, int cap = 256 * 8) {
    int64_t g = (work_items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

}  // namespace

// ------------------------------------------------------------------------------------------------ C ABI
static int ln_geometry(int C, int* lpt, int* vpl) {
    if (C % 4 != 0 || C < 4 || C > 1024) return -1;
    int c4 = C / 4, l = 1;
    while (l < 64 && l < c4) l <<= 1;
    if (l > c4) return -1;          // C/4 must be a power of two below 64 lanes ...
    if (c4 % l != 0) return -1;
    *lpt = l;
    *vpl = c4 / l;                  // ... or a multiple of 64 float4 (C = 256, 512, 768, 1024)
    return (*vpl >= 1 && *vpl <= 4) ? 0 : -1;
}

// bf16 storage, 8 channels per lane: lanes per token and quads per lane (2 or 4), or -1 when C does not split that way
static int wide_geometry(int C, int* lpt, int* vpl) {
    if (C % 8 != 0) return -1;
    int l, v;
    if (ln_geometry(C / 2, &l, &v) != 0 || 2 * v > 4) return -1;
    *lpt = l;
    *vpl = 2 * v;
    return 0;
}

// Each token-tensor op exists as dhz_<op>_dt(..., dtype, stream) with dtype = DHZ_F32 / DHZ_BF16 (the storage type of the token
// tensors; parameters, statistics and parameter gradients are always fp32) and as the fp32 entry point dhz_<op>.
#define DT_SWITCH(dtype, who, CALL)                                                      \
    do {                                                                                 \
        if ((dtype) == DHZ_F32) { typedef float T; CALL; }                               \
        else if ((dtype) == DHZ_BF16) { typedef bf16s T; CALL; }                         \
        else { dhz_set_error("%s: unknown dtype %d", who, (int)(dtype)); return DHZ_EINVAL; } \
    } while (0)

extern "C" int dhz_ln_partition_fwd_dt(const void* x, const float* gamma, const float* beta, void* xw, float* stats,
                                       int B, int Hres, int Wres, int C, int shift, int partition, int dtype, void* stream) {
    DHZ_REQUIRE(x && gamma && beta && xw, "dhz_ln_partition_fwd: null pointer");
    DHZ_REQUIRE(B > 0 && Hres > 0 && Wres > 0 && (!partition || (Hres % 8 == 0 && Wres % 8 == 0 && shift >= 0 && shift < 8)),
                "dhz_ln_partition_fwd: bad shape");
    int lpt, vpl;
    DHZ_REQUIRE(ln_geometry(C, &lpt, &vpl) == 0, "dhz_ln_partition_fwd: unsupported C=%d", C);
    const int ntok = B * Hres * Wres;
    hipStream_t s = (hipStream_t)stream;
    int lpw, vpw;
    if (dtype == DHZ_BF16 && wide_geometry(C, &lpw, &vpw) == 0) {            // bf16: 8 channels (16 bytes) per lane
        const int grid = grid_for((int64_t)ntok * lpw);
#define LAUNCHW(V) hipLaunchKernelGGL((ln_partition_fwd_kernel<V, bf16s, true>), dim3(grid), dim3(256), 0, s, (const bf16s*)x, gamma, beta, (bf16s*)xw, stats, ntok, Hres, Wres, C, shift, lpw, partition)
        if (vpw == 2) LAUNCHW(2); else LAUNCHW(4);
#undef LAUNCHW
        DHZ_CHECK_LAUNCH("dhz_ln_partition_fwd");
        return DHZ_OK;
    }
    const int grid = grid_for((int64_t)ntok * lpt);
#define LAUNCH(V) hipLaunchKernelGGL((ln_partition_fwd_kernel<V, T>), dim3(grid), dim3(256), 0, s, (const T*)x, gamma, beta, (T*)xw, stats, ntok, Hres, Wres, C, shift, lpt, partition)
    DT_SWITCH(dtype, "dhz_ln_partition_fwd",
              switch (vpl) { case 1: LAUNCH(1); break; case 2: LAUNCH(2); break; case 3: LAUNCH(3); break; default: LAUNCH(4); });
#undef LAUNCH
    DHZ_CHECK_LAUNCH("dhz_ln_partition_fwd");
    return DHZ_OK;
}
extern "C" int dhz_ln_partition_fwd(const float* x, const float* gamma, const float* beta, float* xw, float* stats,
                                    int B, int Hres, int Wres, int C, int shift, int partition, void* stream) {
    return dhz_ln_partition_fwd_dt(x, gamma, beta, xw, stats, B, Hres, Wres, C, shift, partition, DHZ_F32, stream);
}

extern "C" int dhz_ln_partition_bwd_lay2(const void* dxw, const void* x, const float* gamma, const float* stats,
                                         const void* dres, void* dx, float* dgamma, float* dbeta, int B, int Hres, int Wres,
                                         int C, int shift, int partition, int dres_windowed, int dx_windowed, int dx_shift, void* dx2,
                                         const float* scale2, int dtype, void* stream) {
    DHZ_REQUIRE(!dx2 || (Hres % 8 == 0 && Wres % 8 == 0 && dx_shift >= 0 && dx_shift < 8 && dx2 != dres && dx2 != dx),
                "dhz_ln_partition_bwd_lay2: the second (window-ordered) output needs an Hres x Wres map of multiples of 8, a shift in [0, 8) and its own buffer");
    DHZ_REQUIRE(dxw && x && gamma && stats && dx && dgamma && dbeta, "dhz_ln_partition_bwd: null pointer");
    DHZ_REQUIRE(B > 0 && Hres > 0 && Wres > 0 && (!partition || (Hres % 8 == 0 && Wres % 8 == 0 && shift >= 0 && shift < 8)),
                "dhz_ln_partition_bwd: bad shape");
    DHZ_REQUIRE(!dres_windowed || partition, "dhz_ln_partition_bwd_lay: a window-ordered dres needs partition = 1");
    DHZ_REQUIRE(!dx_windowed || (Hres % 8 == 0 && Wres % 8 == 0 && dx_shift >= 0 && dx_shift < 8 && dx != dres),
                "dhz_ln_partition_bwd_lay: a window-ordered dx needs an Hres x Wres map of multiples of 8, a shift in [0, 8) and dx != dres");
    const int lay = (dres_windowed ? 1 : 0) | (dx_windowed ? 2 : 0) | ((dx_windowed || dx2) ? (dx_shift << 8) : 0);
    int lpt, vpl;
    DHZ_REQUIRE(ln_geometry(C, &lpt, &vpl) == 0, "dhz_ln_partition_bwd: unsupported C=%d", C);
    const int ntok = B * Hres * Wres;
    // every workgroup ends with 2C same-address atomics (dgamma, dbeta): give each wave >= 8 token groups so that small
    // maps do not pay 1024 workgroups' worth of them
    int lpw, vpw;
    const bool wide = dtype == DHZ_BF16 && wide_geometry(C, &lpw, &vpw) == 0;   // bf16: 8 channels (16 bytes) per lane
    if (wide) lpt = lpw;
    int grid = (int)(((int64_t)ntok * lpt + 256 * 8 - 1) / (256 * 8));
    const int cap = 2 * dhz_num_cus();
    grid = grid < 64 ? 64 : (grid > cap ? cap : grid);
    hipStream_t s = (hipStream_t)stream;
    if (wide) {
#define LAUNCHW(V) hipLaunchKernelGGL((ln_partition_bwd_kernel<V, bf16s, true>), dim3(grid), dim3(256), 0, s, (const bf16s*)dxw, (const bf16s*)x, gamma, stats, (const bf16s*)dres, (bf16s*)dx, dgamma, dbeta, ntok, Hres, Wres, C, shift, lpt, partition, lay, (bf16s*)dx2, scale2)
        if (vpw == 2) LAUNCHW(2); else LAUNCHW(4);
#undef LAUNCHW
        DHZ_CHECK_LAUNCH("dhz_ln_partition_bwd");
        return DHZ_OK;
    }
#define LAUNCH(V) hipLaunchKernelGGL((ln_partition_bwd_kernel<V, T>), dim3(grid), dim3(256), 0, s, (const T*)dxw, (const T*)x, gamma, stats, (const T*)dres, (T*)dx, dgamma, dbeta, ntok, Hres, Wres, C, shift, lpt, partition, lay, (T*)dx2, scale2)
    DT_SWITCH(dtype, "dhz_ln_partition_bwd",
              switch (vpl) { case 1: LAUNCH(1); break; case 2: LAUNCH(2); break; case 3: LAUNCH(3); break; default: LAUNCH(4); });
#undef LAUNCH
    DHZ_CHECK_LAUNCH("dhz_ln_partition_bwd");
    return DHZ_OK;
}
extern "C" int dhz_ln_partition_bwd_lay(const void* dxw, const void* x, const float* gamma, const float* stats,
                                        const void* dres, void* dx, float* dgamma, float* dbeta, int B, int Hres, int Wres,
                                        int C, int shift, int partition, int dres_windowed, int dx_windowed, int dx_shift, int dtype,
                                        void* stream) {
    return dhz_ln_partition_bwd_lay2(dxw, x, gamma, stats, dres, dx, dgamma, dbeta, B, Hres, Wres, C, shift, partition, dres_windowed, dx_windowed,
                                     dx_shift, nullptr, nullptr, dtype, stream);
}
extern "C" int dhz_ln_partition_bwd_dt(const void* dxw, const void* x, const float* gamma, const float* stats,
                                       const void* dres, void* dx, float* dgamma, float* dbeta, int B, int Hres, int Wres,
                                       int C, int shift, int partition, int dtype, void* stream) {
    return dhz_ln_partition_bwd_lay(dxw, x, gamma, stats, dres, dx, dgamma, dbeta, B, Hres, Wres, C, shift, partition, 0, 0, 0, dtype, stream);
}
extern "C" int dhz_ln_partition_bwd(const float* dxw, const float* x, const float* gamma, const float* stats,
                                    const float* dres, float* dx, float* dgamma, float* dbeta, int B, int Hres, int Wres,
                                    int C, int shift, int partition, void* stream) {
    return dhz_ln_partition_bwd_dt(dxw, x, gamma, stats, dres, dx, dgamma, dbeta, B, Hres, Wres, C, shift, partition, DHZ_F32, stream);
}

extern "C" int dhz_reverse_residual_fwd_dt(const void* yw, const void* shortcut, const float* scale, void* out, int B,
                                           int Hres, int Wres, int C, int shift, int partition, int dtype, void* stream) {
    DHZ_REQUIRE(yw && shortcut && out, "dhz_reverse_residual_fwd: null pointer");
    DHZ_REQUIRE(B > 0 && C % 4 == 0 && (!partition || (Hres % 8 == 0 && Wres % 8 == 0 && shift >= 0 && shift < 8)),
                "dhz_reverse_residual_fwd: bad shape");
    const int ntok = B * Hres * Wres;
    DT_SWITCH(dtype, "dhz_reverse_residual_fwd",
              hipLaunchKernelGGL((reverse_residual_kernel<false, T>), dim3(grid_for((int64_t)ntok * (C / 4))), dim3(256), 0,
                                 (hipStream_t)stream, (const T*)yw, (const T*)shortcut, scale, (T*)out, ntok, Hres, Wres, C / 4, shift,
                                 partition));
    DHZ_CHECK_LAUNCH("dhz_reverse_residual_fwd");
    return DHZ_OK;
}
extern "C" int dhz_reverse_residual_fwd(const float* yw, const float* shortcut, const float* scale, float* out, int B,
                                        int Hres, int Wres, int C, int shift, int partition, void* stream) {
    return dhz_reverse_residual_fwd_dt(yw, shortcut, scale, out, B, Hres, Wres, C, shift, partition, DHZ_F32, stream);
}

extern "C" int dhz_reverse_residual_bwd_dt(const void* dout, const float* scale, void* dyw, int B, int Hres, int Wres,
                                           int C, int shift, int partition, int dtype, void* stream) {
    DHZ_REQUIRE(dout && dyw, "dhz_reverse_residual_bwd: null pointer");
    DHZ_REQUIRE(B > 0 && C % 4 == 0 && (!partition || (Hres % 8 == 0 && Wres % 8 == 0 && shift >= 0 && shift < 8)),
                "dhz_reverse_residual_bwd: bad shape");
    const int ntok = B * Hres * Wres;
    DT_SWITCH(dtype, "dhz_reverse_residual_bwd",
              hipLaunchKernelGGL((reverse_residual_kernel<true, T>), dim3(grid_for((int64_t)ntok * (C / 4))), dim3(256), 0,
                                 (hipStream_t)stream, (const T*)dout, (const T*)nullptr, scale, (T*)dyw, ntok, Hres, Wres, C / 4, shift,
                                 partition));
    DHZ_CHECK_LAUNCH("dhz_reverse_residual_bwd");
    return DHZ_OK;
}
extern "C" int dhz_reverse_residual_bwd(const float* dout, const float* scale, float* dyw, int B, int Hres, int Wres,
                                        int C, int shift, int partition, void* stream) {
    return dhz_reverse_residual_bwd_dt(dout, scale, dyw, B, Hres, Wres, C, shift, partition, DHZ_F32, stream);
}

extern "C" int dhz_gelu_fwd_dt(const void* u, void* y, int64_t n, int dtype, void* stream) {
    DHZ_REQUIRE(u && y && n > 0 && n % 4 == 0, "dhz_gelu_fwd: bad arguments (n must be a multiple of 4)");
    DT_SWITCH(dtype, "dhz_gelu_fwd", hipLaunchKernelGGL((gelu_fwd_kernel<T>), dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream,
                                                        (const T*)u, (T*)y, n / 4));
    DHZ_CHECK_LAUNCH("dhz_gelu_fwd");
    return DHZ_OK;
}
extern "C" int dhz_gelu_bwd_dt(const void* dy, const void* u, void* du, int64_t n, const float* scale, int64_t elems_per_scale, int dtype,
                               void* stream) {
    DHZ_REQUIRE(dy && u && du && n > 0 && n % 4 == 0, "dhz_gelu_bwd: bad arguments (n must be a multiple of 4)");
    DHZ_REQUIRE(!scale || (elems_per_scale > 0 && elems_per_scale % 4 == 0 && n % elems_per_scale == 0),
                "dhz_gelu_bwd: elems_per_scale=%lld must be a multiple of 4 that divides n", (long long)elems_per_scale);
    DT_SWITCH(dtype, "dhz_gelu_bwd", hipLaunchKernelGGL((gelu_bwd_kernel<T>), dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream,
                                                        (const T*)dy, (const T*)u, (T*)du, n / 4, scale, scale ? elems_per_scale / 4 : 1));
    DHZ_CHECK_LAUNCH("dhz_gelu_bwd");
    return DHZ_OK;
}

// lanes per position of the two depthwise kernels: 16 (64 channels per workgroup) where asked for and Ch allows
static int dw_lanes_per_position(int want, int Ch) { return (want == 16 && Ch % 64 == 0) ? 16 : 8; }

extern "C" int dhz_leff_dwconv_fwd_dt(const void* u, const float* w, const float* b, void* t, void* z, int B, int Hres,
                                      int Wres, int Ch, int dtype, void* stream) {
    DHZ_REQUIRE(u && w && b && z, "dhz_leff_dwconv_fwd: null pointer");
    DHZ_REQUIRE(B > 0 && Hres > 0 && Wres > 0 && Ch % CT == 0, "dhz_leff_dwconv_fwd: Ch=%d must be a multiple of %d", Ch, CT);
    const int tiles_x = (Wres + TW - 1) / TW, tiles_y = (Hres + TH - 1) / TH;
    const int lpp = dw_lanes_per_position(DW_FWD_LPP, Ch);
    const int grid = B * tiles_x * tiles_y * (Ch / (4 * lpp));
#define DW_FWD(LPP_) hipLaunchKernelGGL((leff_dwconv_fwd_kernel<T, LPP_>), dim3(grid), dim3(32 * LPP_), 0, (hipStream_t)stream, (const T*)u, \
                                        w, b, (T*)t, (T*)z, Hres, Wres, Ch, tiles_x, tiles_y)
    DT_SWITCH(dtype, "dhz_leff_dwconv_fwd", if (lpp == 16) DW_FWD(16); else DW_FWD(8));
#undef DW_FWD
    DHZ_CHECK_LAUNCH("dhz_leff_dwconv_fwd");
    return DHZ_OK;
}
extern "C" int dhz_leff_dwconv_fwd(const float* u, const float* w, const float* b, float* t, float* z, int B, int Hres,
                                   int Wres, int Ch, void* stream) {
    return dhz_leff_dwconv_fwd_dt(u, w, b, t, z, B, Hres, Wres, Ch, DHZ_F32, stream);
}

extern "C" int dhz_leff_dwconv_bwd_scaled_dt(const void* dz, const void* u, const void* t, const float* w, void* du, float* dw,
                                             float* db, const float* dz_scale, int B, int Hres, int Wres, int Ch, int dtype,
                                             void* stream) {
    DHZ_REQUIRE(dz && u && t && w && du && dw && db, "dhz_leff_dwconv_bwd: null pointer");
    DHZ_REQUIRE(B > 0 && Hres > 0 && Wres > 0 && Ch % CT == 0, "dhz_leff_dwconv_bwd: Ch=%d must be a multiple of %d", Ch, CT);
    const int tiles_x = (Wres + TW - 1) / TW, tiles_y = (Hres + TH - 1) / TH;
    const int lpp = dw_lanes_per_position(DW_BWD_LPP, Ch);
    const int ntiles = B * tiles_x * tiles_y, ncg = Ch / (4 * lpp);
    int wg_per_cg = (lpp == 16 ? 512 : 256 * DWB_WAVES) / ncg;   // one resident round: 3 (2 at 512 threads) workgroups per CU, persistent over tiles
    if (wg_per_cg < 1) wg_per_cg = 1;
    if (wg_per_cg > ntiles) wg_per_cg = ntiles;
#define DW_BWD(LPP_) hipLaunchKernelGGL((leff_dwconv_bwd_kernel<T, LPP_>), dim3(wg_per_cg * ncg), dim3(32 * LPP_), 0, (hipStream_t)stream,   \
                                        (const T*)dz, (const T*)u, (const T*)t, w, (T*)du, dw, db, dz_scale, B, Hres, Wres, Ch, tiles_x,      \
                                        tiles_y, wg_per_cg)
    DT_SWITCH(dtype, "dhz_leff_dwconv_bwd", if (lpp == 16) DW_BWD(16); else DW_BWD(8));
#undef DW_BWD
    DHZ_CHECK_LAUNCH("dhz_leff_dwconv_bwd");
    return DHZ_OK;
}
extern "C" int dhz_leff_dwconv_bwd_dt(const void* dz, const void* u, const void* t, const float* w, void* du, float* dw,
                                      float* db, int B, int Hres, int Wres, int Ch, int dtype, void* stream) {
    return dhz_leff_dwconv_bwd_scaled_dt(dz, u, t, w, du, dw, db, nullptr, B, Hres, Wres, Ch, dtype, stream);
}
extern "C" int dhz_leff_dwconv_bwd(const float* dz, const float* u, const float* t, const float* w, float* du, float* dw,
                                   float* db, int B, int Hres, int Wres, int Ch, void* stream) {
    return dhz_leff_dwconv_bwd_dt(dz, u, t, w, du, dw, db, B, Hres, Wres, Ch, DHZ_F32, stream);
}

extern "C" int dhz_charbonnier_fwd(const float* x, const float* y, float* clampd, float* loss_sum, int64_t n, float eps,
                                   int clamp01, void* stream) {
    DHZ_REQUIRE(x && y && loss_sum && n > 0 && n % 4 == 0, "dhz_charbonnier_fwd: bad arguments (n must be a multiple of 4)");
    hipLaunchKernelGGL(charbonnier_fwd_kernel, dim3(grid_for(n / 4, 256, 1024)), dim3(256), 0, (hipStream_t)stream, x, y,
                       clampd, loss_sum, n / 4, eps * eps, clamp01);
    DHZ_CHECK_LAUNCH("dhz_charbonnier_fwd");
    return DHZ_OK;
}

extern "C" int dhz_charbonnier_bwd(const float* x, const float* y, const float* gscale, const float* gclamp, float* dx,
                                   int64_t n, float eps, float inv_n, int clamp01, void* stream) {
    DHZ_REQUIRE(x && y && dx && n > 0 && n % 4 == 0, "dhz_charbonnier_bwd: bad arguments (n must be a multiple of 4)");
    hipLaunchKernelGGL(charbonnier_bwd_kernel, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, x, y, gscale, gclamp,
                       dx, n / 4, eps * eps, inv_n, clamp01);
    DHZ_CHECK_LAUNCH("dhz_charbonnier_bwd");
    return DHZ_OK;
}

extern "C" int dhz_adamw_step_shadow(float* p, const float* g, float* m, float* v, void* p16, int64_t n, float lr, float beta1,
                                     float beta2, float eps, float wd, int step, float grad_scale, void* stream) {
    DHZ_REQUIRE(p && g && m && v && n > 0 && step >= 1, "dhz_adamw_step: bad arguments");
    DHZ_REQUIRE(((uintptr_t)p16 & 7) == 0, "dhz_adamw_step: the bf16 shadow must be 8-byte aligned");
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1);
    const float bc2_sqrt = (float)sqrt(bc2);
    hipLaunchKernelGGL(adamw_kernel, dim3(grid_for((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (uint16_t*)p16, n,
                       lr, beta1, beta2, eps, wd, step_size, bc2_sqrt, grad_scale);
    DHZ_CHECK_LAUNCH("dhz_adamw_step");
    return DHZ_OK;
}
extern "C" int dhz_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                              float eps, float wd, int step, float grad_scale, void* stream) {
    return dhz_adamw_step_shadow(p, g, m, v, nullptr, n, lr, beta1, beta2, eps, wd, step, grad_scale, stream);
}

extern "C" int dhz_contrast_combine_fwd(const float* sums, const float* inv_cnt, const float* w, int k, int ablation, float* d,
                                        float* out, void* stream) {
    DHZ_REQUIRE(sums && inv_cnt && w && d && out && k > 0 && k <= 64, "dhz_contrast_combine_fwd: bad arguments (1 <= k <= 64)");
    hipLaunchKernelGGL(contrast_combine_fwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sums, inv_cnt, w, k, ablation, d, out);
    DHZ_CHECK_LAUNCH("dhz_contrast_combine_fwd");
    return DHZ_OK;
}

extern "C" int dhz_contrast_combine_bwd(const float* d, const float* w, int k, int ablation, const float* g_loss, const float* g_ap,
                                        const float* g_an, float* g, void* stream) {
    DHZ_REQUIRE(d && w && g && k > 0 && k <= 64, "dhz_contrast_combine_bwd: bad arguments (1 <= k <= 64)");
    hipLaunchKernelGGL(contrast_combine_bwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, d, w, k, ablation, g_loss, g_ap, g_an, g);
    DHZ_CHECK_LAUNCH("dhz_contrast_combine_bwd");
    return DHZ_OK;
}

extern "C" int dhz_l1_pair_fwd(const float* a, const float* p, const float* n, float* sums, int64_t count, void* stream) {
    DHZ_REQUIRE(a && p && sums && count > 0 && count % 4 == 0, "dhz_l1_pair_fwd: bad arguments (count must be a multiple of 4)");
    hipLaunchKernelGGL(l1_pair_fwd_kernel, dim3(grid_for(count / 4, 256, 768)), dim3(256), 0, (hipStream_t)stream, a, p, n, sums,
                       count / 4);
    DHZ_CHECK_LAUNCH("dhz_l1_pair_fwd");
    return DHZ_OK;
}

extern "C" int dhz_l1_pair_bwd(const float* a, const float* p, const float* n, const float* g, float* da, int64_t count,
                               void* stream) {
    DHZ_REQUIRE(a && p && g && da && count > 0 && count % 4 == 0, "dhz_l1_pair_bwd: bad arguments (count must be a multiple of 4)");
    hipLaunchKernelGGL(l1_pair_bwd_kernel, dim3(grid_for(count / 4)), dim3(256), 0, (hipStream_t)stream, a, p, n, g,
                       1.0f / (float)count, da, count / 4);
    DHZ_CHECK_LAUNCH("dhz_l1_pair_bwd");
    return DHZ_OK;
}
