// Fused window-attention branch of a LeWin block (forward), C = 32 / 64 / 128 (head_dim 32):
//
//   out = x + drop_scale[b] * OutProj( ProbSparseAttention( QKV( roll+partition( LayerNorm(x) ) ) ) )   un-rolled
//
// i.e. M1:839-872 (norm1, roll, window_partition, WindowAttention -> AttentionLayer -> ProbAttention,
// window_reverse, roll back, shortcut + drop_path) in ONE kernel.  One 256-thread workgroup per 8x8 window:
//   0. the window's 64 token rows are gathered with the cyclic shift folded into the addresses, normalised
//      (4 lanes per token) and kept in LDS (xn);
//   per head h (d = 32):
//   1. [Q_h|K_h|V_h] = xn W_h^T + b_h on the fp32 matrix pipe; A fragments from LDS, B fragments straight from
//      a fragment-ordered ("prepacked") copy of the weights with coalesced float4 loads (L1/L2 resident);
//   2. S = Q_h K_h^T, sparsity measure, top-u ranks, double softmax, O = P V_h (+ mean(V) row) exactly as
//      ps_attn_fwd_kernel, all operands already in LDS;
//   3. the out-projection is accumulated head by head in registers: acc += ctx_h Wo[:, 32h:32h+32]^T with the
//      "selected row or mean row" choice folded into the A-fragment row index;
//   4. epilogue: + bias, * drop-path scale, + shortcut (re-read of x, L2 hit), scatter back to token order.
// In training mode (SAVE) the kernel additionally writes what the (unfused) backward kernels consume:
// LN statistics, xn, the packed QKV rows, the attention context and the selection ranks - 7C floats of HBM
// traffic per token instead of the 15C of the unfused forward chain; in inference mode only x is read and
// out written (2C per token) and the kernel is bound by the fp32 MFMA rate (AI 50.7 FLOP/B at C = 32).
#include <stdlib.h>
#include "common.h"

namespace {

#ifndef FUSED_PERSIST_C32
#define FUSED_PERSIST_C32 1
#endif
// Phase skipping for timing diagnostics is a COMPILE-TIME constant of a variant build (tools/abl_fused.sh links copies of the
// library with -DDHZ_FUSED_ABL=<bit mask>; outputs are then wrong): the product kernel has no such argument and no such branch.
#ifndef DHZ_FUSED_ABL
#define DHZ_FUSED_ABL 0
#endif
constexpr int NT = 64;
constexpr int NU = 25;
constexpr int SS = 68;     // row stride of 64-wide score tiles
constexpr int HS = 36;     // row stride of the per-head 32-wide tiles

// Cross-lane exchanges inside aligned groups of 4 / 8 lanes on the VALU (DPP), not through the LDS crossbar
// (ds_bpermute costs an LDS round trip per step; a softmax row needs 12 dependent steps):
//   quad_perm [1,0,3,2] = xor 1, quad_perm [2,3,0,1] = xor 2, row_half_mirror = lane i <-> 7-i within 8 lanes.
template <int CTRL>
__device__ __forceinline__ float dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float r4max(float v) { v = fmaxf(v, dpp<0xB1>(v)); return fmaxf(v, dpp<0x4E>(v)); }
__device__ __forceinline__ float r4sum(float v) { v += dpp<0xB1>(v); return v + dpp<0x4E>(v); }
__device__ __forceinline__ float r8max(float v) { v = r4max(v); return fmaxf(v, dpp<0x141>(v)); }
__device__ __forceinline__ float r8sum(float v) { v = r4sum(v); return v + dpp<0x141>(v); }

// ---- six-term projections (P6): the QKV product of a head on the bf16 matrix pipe, fp32-class (csrc/split6_gemm.hip's arithmetic:
// three bf16 truncation pieces per operand value, products hh hm mh hl lh mm, dropped terms <= 2^-24 relative, fp32 accumulation)
typedef __bf16 bf16x8_ __attribute__((ext_vector_type(8)));
typedef short s16x8_ __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4_ __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_;
typedef const __attribute__((address_space(1))) void glb_void_;
__device__ __forceinline__ f32x4 mfma_b16(u32x4_ a, u32x4_ b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_, a), __builtin_bit_cast(bf16x8_, b), c, 0, 0, 0);
}
__device__ __forceinline__ uint32_t pack_top_(float x1, float x0) {              // (x1 & 0xffff0000) | (x0 >> 16)
    return __builtin_amdgcn_perm(__float_as_uint(x1), __float_as_uint(x0), 0x07060302u);
}
// eight fp32 values -> their three bf16 pieces (hi + mid + lo == x exactly), packed as MFMA operands
__device__ __forceinline__ void split8x3_(const float* x, u32x4_& hi, u32x4_& mid, u32x4_& lo) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float x0 = x[2 * i], x1 = x[2 * i + 1];
        const float r0 = x0 - __uint_as_float(__float_as_uint(x0) & 0xffff0000u), r1 = x1 - __uint_as_float(__float_as_uint(x1) & 0xffff0000u);
        const float q0 = r0 - __uint_as_float(__float_as_uint(r0) & 0xffff0000u), q1 = r1 - __uint_as_float(__float_as_uint(r1) & 0xffff0000u);
        hi[i] = pack_top_(x1, x0);
        mid[i] = pack_top_(r1, r0);
        lo[i] = pack_top_(q1, q0);
    }
}
constexpr int P6_RUNS = 36;            // 1 KiB runs (one per wave-wide 16-byte fragment read) of a head's QKV planes per 64 channels:
                                       // 6 column tiles x 2 k-blocks of 32 x 3 pieces

// v_writelane_b32: lane LANE of `old` := the wave-uniform value (this compiler has no builtin for it).  The lane select is an
// inline constant - a second SGPR would exceed the one-scalar-operand (constant bus) limit of a VALU instruction; the value is a
// SALU result, which the hardware interlocks (no manual wait states).
template <int LANE>
__device__ __forceinline__ int writelane(int old, int val_uniform) {
    asm("v_writelane_b32 %0, %1, %2" : "+v"(old) : "s"(val_uniform), "n"(LANE));
    return old;
}

__device__ __forceinline__ void ld4(const float* p, float* dst) {
    const float4 v = *reinterpret_cast<const float4*>(p);
    dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
}

// 47.4 KB for every C: three workgroups per CU.  (Round 2 kept the LayerNorm output in an LDS tile of its own and P beside K:
// 48.6 / 73.5 / 107 KB at C = 32 / 64 / 128, i.e. two / two / one workgroup per CU.)
template <int C>
struct FusedSmem {
    float q[NT * HS];          // Q_h, later O_h (32 x HS)
    float k[NT * HS];          // K_h; dead once S is complete, then P (32 x SS: rows 0..24 selected queries, row 25 = 1/64)
    float v[NT * HS];
    float s[NT * SS];          // S = Q_h K_h^T; after the last head the out-projection staging tile (rows of wave w are its own)
    float m[NT];               // sparsity measure
    float vec[6 * C];          // gamma | beta | out-projection bias | Q, K, V biases (loop invariants: in LDS, not in registers,
                               // and no global load for them inside the window loop)
    int top[4][32];            // per-wave copy of the selected-query list (every wave derives the full ranking itself)
    uint8_t idx[NT * NU];
};

// SAVE: 0 = inference, 1 = every tensor the backward kernel chain consumes, 2 = only the selection ranks (the fused backward of
// csrc/fused_attn_bwd.hip recomputes the rest)
// P6: wqkv_p holds the six-term planes of dhz_fused_attn_prepack6 instead of fp32 fragments; the QKV product of a head runs on the bf16
// matrix pipe with its weight planes brought ONCE per workgroup by LDS-DMA into the Q / K / V / S tiles (dead at that point: 36 KiB per 64
// channels), shared by the four waves - not 4 x through L1 per wave (the round-4 attempt), three workgroups per CU kept (not the
// resident-planes form of round 5).
template <int C, int SAVE, int NW = 1, bool P6 = false>
__global__ __launch_bounds__(256 * NW, NW == 2 ? 1 : (C == 64 ? 3 : (C == 128 ? 1 : 2))) void fused_window_attn_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float4* __restrict__ wqkv_p, const float* __restrict__ bqkv, const float4* __restrict__ wo_p,
    const float* __restrict__ bo, const uint8_t* __restrict__ idx, const float* __restrict__ bias,
    const float* __restrict__ mask, const float* __restrict__ dscale, float* __restrict__ out,
    float* __restrict__ xn_save, float* __restrict__ qkv_save, float* __restrict__ ctx_save,
    float* __restrict__ stats_save, uint8_t* __restrict__ rank_save, int Hres, int Wres, int shift, int nwin) {
    constexpr int abl = DHZ_FUSED_ABL;
    constexpr int H = C / 32;
    constexpr int CPT = C / 4;            // floats per thread in the token-row phases (4 threads per token)
    constexpr int KS = C / 4;             // k-steps of the QKV GEMM
    constexpr int KS4 = KS / 4;
    constexpr int P6_QKV_BYTES = (C / 32) * (C / 64 > 0 ? C / 64 : 1) * P6_RUNS * 1024;     // the Q / K / V planes; the out-projection's follow
    constexpr int P6O_RUNS = (C / 16) * 3;                                                    // 1 KiB runs of one head's out-projection planes
    constexpr bool P6O = P6 && C >= 64 && P6O_RUNS * 1024 <= (int)sizeof(float) * NT * SS && P6O_RUNS % 4 == 0;   // ... fit the S tile (C = 64)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    // NW = 2: a 512-thread workgroup carries TWO windows, one per half (threads 0..255 / 256..511), each with its own LDS tiles; the
    // halves run the same phases in lockstep (the barriers are the workgroup's).  t, w, lane are the indices INSIDE the half.
    const int half = NW == 2 ? (int)(threadIdx.x >> 8) : 0;
    FusedSmem<C>& sm = reinterpret_cast<FusedSmem<C>*>(smem_raw)[half];
    const int t = threadIdx.x & 255, lane = t & 63, w = t >> 6;
    const int i16 = lane & 15, g = lane >> 4;
    const int nWw = Wres >> 3, nW = (Hres >> 3) * nWw;
    // token row / channel quarter of this thread in the row phases = the row / k range of its MFMA A fragment: lane (i16, g) of
    // wave w owns channels g C/4 .. of token 16 w + i16, so the LayerNorm output IS the A fragment of the projections - it never
    // visits LDS (the four lanes of a token differ in lane bits 4 and 5)
    const int tl = 16 * w + i16, qd = g;
    float* const P = sm.k;
    auto tok4sum = [](float v) -> float { v += __shfl_xor(v, 16); return v + __shfl_xor(v, 32); };
    const float scale = 0.17677669529663687f;      // 1/sqrt(32)

    // token this thread's row comes from (and goes back to) for window `win`: cyclic shift folded in
    auto src_token = [&](int win) -> size_t {
        const int bimg = win / nW, wdx = win % nW;
        int hh = (wdx / nWw) * 8 + (tl >> 3) + shift; if (hh >= Hres) hh -= Hres;
        int ww = (wdx % nWw) * 8 + (tl & 7) + shift; if (ww >= Wres) ww -= Wres;
        return (size_t)bimg * Hres * Wres + (size_t)hh * Wres + ww;
    };

    // C == 32: all projection weights (64 VGPRs of B fragments) stay in registers for the kernel's lifetime
    constexpr bool WREG = (C == 32) && FUSED_PERSIST_C32;
    constexpr bool WREG6 = WREG && P6;         // C = 32 with six-term products: the bf16 planes of all four weights live in registers (24 KiB per wave-set)
    float4 wr_qkv[(WREG && !P6) ? 12 : 1], wr_o[(WREG && !P6) ? 4 : 1];
    u32x4_ w6r[WREG6 ? 6 : 1][3], wo6r[WREG6 ? 2 : 1][3];
    if constexpr (WREG6) {
        // planes of dhz_fused_attn_prepack6 at C = 32: run ((j 3) + piece) for the six Q / K / V column tiles, then ((6 + tn) 3 + piece) for the
        // out-projection's two
        const unsigned char* pl = reinterpret_cast<const unsigned char*>(wqkv_p) + lane * 16;
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) w6r[j][pc] = *reinterpret_cast<const u32x4_*>(pl + (j * 3 + pc) * 1024);
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) wo6r[tn][pc] = *reinterpret_cast<const u32x4_*>(pl + ((6 + tn) * 3 + pc) * 1024);
    } else if constexpr (WREG) {
#pragma unroll
        for (int i = 0; i < 12; ++i) wr_qkv[i] = wqkv_p[i * 64 + lane];
#pragma unroll
        for (int i = 0; i < 4; ++i) wr_o[i] = wo_p[i * 64 + lane];
    }
    if (t < NT * NU / 16) reinterpret_cast<uint4*>(sm.idx)[t] = reinterpret_cast<const uint4*>(idx)[t];

    // loop-invariant vectors in registers: nothing inside the window loop waits on a global load except the bias /
    // mask rows (vmcnt is an in-order counter - a waited load also waits for every older load and store)
    for (int i = t; i < 6 * C; i += 256)
        sm.vec[i] = i < C ? gamma[i] : (i < 2 * C ? beta[i - C] : (i < 3 * C ? bo[i - 2 * C] : bqkv[i - 3 * C]));
    __syncthreads();
    const float4* const gmr = reinterpret_cast<const float4*>(sm.vec + qd * CPT);
    const float4* const btr = reinterpret_cast<const float4*>(sm.vec + C + qd * CPT);
    const float4* const bor = reinterpret_cast<const float4*>(sm.vec + 2 * C + qd * CPT);
    float4 xv[CPT / 4], xnext[CPT / 4];
    int win = blockIdx.x * NW + half;
    if (win < nwin) {
        const float4* xp = reinterpret_cast<const float4*>(x + src_token(win) * C + qd * CPT);
#pragma unroll
        for (int i = 0; i < CPT / 4; ++i) xv[i] = xp[i];
    }
#pragma unroll 1
    for (; win < nwin; win += gridDim.x * NW) {
        const int bimg = win / nW, wdx = win % nW;
        const size_t src_tok = src_token(win);
        // ---- 0. LayerNorm of the gathered rows (4 lanes per token); xa = this lane's A fragment for every head's projection
        float xa[KS];
        {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < CPT / 4; ++i) s += xv[i].x + xv[i].y + xv[i].z + xv[i].w;
            s = tok4sum(s);
            const float mean = s * (1.0f / C);
            float var = 0.f;
#pragma unroll
            for (int i = 0; i < CPT / 4; ++i) {
                const float a0 = xv[i].x - mean, a1 = xv[i].y - mean, a2 = xv[i].z - mean, a3 = xv[i].w - mean;
                var += a0 * a0 + a1 * a1 + a2 * a2 + a3 * a3;
            }
            var = tok4sum(var);
            const float rstd = rsqrtf(var * (1.0f / C) + 1e-5f);
#pragma unroll
            for (int i = 0; i < CPT / 4; ++i) {
                const float4 gm = gmr[i], bt = btr[i];
                float4 y;
                y.x = (xv[i].x - mean) * rstd * gm.x + bt.x;
                y.y = (xv[i].y - mean) * rstd * gm.y + bt.y;
                y.z = (xv[i].z - mean) * rstd * gm.z + bt.z;
                y.w = (xv[i].w - mean) * rstd * gm.w + bt.w;
                xa[4 * i] = y.x; xa[4 * i + 1] = y.y; xa[4 * i + 2] = y.z; xa[4 * i + 3] = y.w;
                if (SAVE == 1 && !(abl & 128)) reinterpret_cast<float4*>(xn_save + ((size_t)win * NT + tl) * C + qd * CPT)[i] = y;
            }
            if (SAVE == 1 && !(abl & 128) && qd == 0) *reinterpret_cast<float2*>(stats_save + 2 * src_tok) = make_float2(mean, rstd);
        }
        // P6: the lane's A fragments as bf16 pieces; k-block kb (32 channels of the contraction, 8 per lane group) <-> channels
        // g C/4 + 8 kb + e - the same (arbitrary, consistent) order the planes are packed in
        u32x4_ a6[P6 ? KS / 8 : 1][3];
        if constexpr (P6) {
#pragma unroll
            for (int kb = 0; kb < KS / 8; ++kb) split8x3_(&xa[8 * kb], a6[kb][0], a6[kb][1], a6[kb][2]);
        }
        f32x4 oacc[C / 16];                    // out-projection accumulators: rows 16w.., all C columns
#pragma unroll
        for (int i = 0; i < C / 16; ++i) oacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
        for (int h = 0; h < H; ++h) {
            // ---- 1. [Q_h | K_h | V_h] = xn W_h^T + b_h : wave w -> rows 16w..16w+15, 6 column tiles, interleaved chains
            if (!(abl & 1)) {
                const float (&a)[KS] = xa;
                f32x4 acc[6];
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    const float bj = sm.vec[3 * C + (j >> 1) * C + 32 * h + 16 * (j & 1) + i16];
                    acc[j] = f32x4{bj, bj, bj, bj};
                }
                if constexpr (WREG6) {
                    constexpr int TA[6] = {2, 0, 1, 0, 1, 0}, TB[6] = {0, 2, 1, 1, 0, 0};      // (token piece, weight piece): lh hl mm hm mh hh
#pragma unroll
                    for (int term = 0; term < 6; ++term)
#pragma unroll
                        for (int j = 0; j < 6; ++j) acc[j] = mfma_b16(a6[0][TA[term]], w6r[j][TB[term]], acc[j]);
                } else if constexpr (P6) {
                    unsigned char* const Bimg = reinterpret_cast<unsigned char*>(sm.q);       // q | k | v | s: 45 KiB contiguous, dead here
                    const unsigned char* const planes = reinterpret_cast<const unsigned char*>(wqkv_p);
                    if (h == 0) __syncthreads();           // the previous window's epilogue staged through S (wave-local, no barrier of its own)
#pragma unroll 1
                    for (int ch = 0; ch < C / 64; ++ch) {
                        if (ch > 0) __syncthreads();       // every wave has read the previous 64 channels' planes
                        const unsigned char* src = planes + ((size_t)(h * (C / 64) + ch) * P6_RUNS) * 1024 + lane * 16;
#pragma unroll
                        for (int r = 0; r < P6_RUNS / 4; ++r)
                            __builtin_amdgcn_global_load_lds((glb_void_*)(src + (w + 4 * r) * 1024), (lds_void_*)(Bimg + (w + 4 * r) * 1024), 16, 0, 0);
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        __syncthreads();
#pragma unroll
                        for (int kbl = 0; kbl < 2; ++kbl) {
                            const int kb = 2 * ch + kbl;
#pragma unroll
                            for (int jg = 0; jg < 6; jg += 3) {                        // three column tiles at a time: 9 fragment reads, 18 MFMAs,
                                u32x4_ bq[3][3];                                       // term-major (an accumulator is reused three MFMAs later)
#pragma unroll
                                for (int jj = 0; jj < 3; ++jj)
#pragma unroll
                                    for (int pc = 0; pc < 3; ++pc)
                                        bq[jj][pc] = *reinterpret_cast<const u32x4_*>(Bimg + (((jg + jj) * 2 + kbl) * 3 + pc) * 1024 + lane * 16);
                                constexpr int TA[6] = {2, 0, 1, 0, 1, 0}, TB[6] = {0, 2, 1, 1, 0, 0};      // (token piece, weight piece): lh hl mm hm mh hh
#pragma unroll
                                for (int term = 0; term < 6; ++term)
#pragma unroll
                                    for (int jj = 0; jj < 3; ++jj) acc[jg + jj] = mfma_b16(a6[kb][TA[term]], bq[jj][TB[term]], acc[jg + jj]);
                            }
                        }
                    }
                    __syncthreads();                       // the planes are read: the results below overwrite them
                } else if constexpr (WREG) {
#pragma unroll
                    for (int s4 = 0; s4 < 2; ++s4) {
#pragma unroll
                        for (int j = 0; j < 6; ++j) acc[j] = mfma16(a[4 * s4 + 0], wr_qkv[j * 2 + s4].x, acc[j]);
#pragma unroll
                        for (int j = 0; j < 6; ++j) acc[j] = mfma16(a[4 * s4 + 1], wr_qkv[j * 2 + s4].y, acc[j]);
#pragma unroll
                        for (int j = 0; j < 6; ++j) acc[j] = mfma16(a[4 * s4 + 2], wr_qkv[j * 2 + s4].z, acc[j]);
#pragma unroll
                        for (int j = 0; j < 6; ++j) acc[j] = mfma16(a[4 * s4 + 3], wr_qkv[j * 2 + s4].w, acc[j]);
                    }
                } else {
                    const float4* wp = wqkv_p + (size_t)h * 6 * KS4 * 64 + lane;
                    float4 bc[6], bn[6];
#pragma unroll
                    for (int j = 0; j < 6; ++j) bc[j] = wp[(j * KS4) * 64];
#pragma unroll
                    for (int s4 = 0; s4 < KS4; ++s4) {
                        if (s4 + 1 < KS4) {
#pragma unroll
                            for (int j = 0; j < 6; ++j) bn[j] = wp[(j * KS4 + s4 + 1) * 64];
                        }
#pragma unroll
                        for (int j = 0; j < 6; ++j) acc[j] = mfma16(a[4 * s4 + 0], bc[j].x, acc[j]);
#pragma unroll
                        for (int j = 0; j < 6; ++j) acc[j] = mfma16(a[4 * s4 + 1], bc[j].y, acc[j]);
#pragma unroll
                        for (int j = 0; j < 6; ++j) acc[j] = mfma16(a[4 * s4 + 2], bc[j].z, acc[j]);
#pragma unroll
                        for (int j = 0; j < 6; ++j) acc[j] = mfma16(a[4 * s4 + 3], bc[j].w, acc[j]);
#pragma unroll
                        for (int j = 0; j < 6; ++j) bc[j] = bn[j];
                    }
                }
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    float* dst = (j < 2 ? sm.q : (j < 4 ? sm.k : sm.v)) + 16 * (j & 1) + i16;
#pragma unroll
                    for (int r = 0; r < 4; ++r) dst[(16 * w + 4 * g + r) * HS] = acc[j][r];
                }
            }
            __syncthreads();
            if (SAVE == 1 && !(abl & 128)) {   // packed QKV rows for the backward kernels: [win*64 + row][3C], head slice 32h..32h+31
                const int row = t >> 2, c8 = (t & 3) * 8;
                float* dst = qkv_save + ((size_t)win * NT + row) * 3 * C + 32 * h + c8;
                *reinterpret_cast<float4*>(dst) = *reinterpret_cast<const float4*>(&sm.q[row * HS + c8]);
                *reinterpret_cast<float4*>(dst + 4) = *reinterpret_cast<const float4*>(&sm.q[row * HS + c8 + 4]);
                *reinterpret_cast<float4*>(dst + C) = *reinterpret_cast<const float4*>(&sm.k[row * HS + c8]);
                *reinterpret_cast<float4*>(dst + C + 4) = *reinterpret_cast<const float4*>(&sm.k[row * HS + c8 + 4]);
                *reinterpret_cast<float4*>(dst + 2 * C) = *reinterpret_cast<const float4*>(&sm.v[row * HS + c8]);
                *reinterpret_cast<float4*>(dst + 2 * C + 4) = *reinterpret_cast<const float4*>(&sm.v[row * HS + c8 + 4]);
            }
            // ---- 2a. S = Q_h K_h^T (4 interleaved chains); wave w writes its own 16 rows of S
            if (!(abl & 2)) {
                float a[8], kb[4][8];
                f32x4 sacc[4];
                ld4(&sm.q[(16 * w + i16) * HS + 8 * g], &a[0]);
                ld4(&sm.q[(16 * w + i16) * HS + 8 * g + 4], &a[4]);
#pragma unroll
                for (int tc = 0; tc < 4; ++tc) {
                    ld4(&sm.k[(16 * tc + i16) * HS + 8 * g], &kb[tc][0]);
                    ld4(&sm.k[(16 * tc + i16) * HS + 8 * g + 4], &kb[tc][4]);
                    sacc[tc] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int s = 0; s < 8; ++s)
#pragma unroll
                    for (int tc = 0; tc < 4; ++tc) sacc[tc] = mfma16(a[s], kb[tc][s], sacc[tc]);
#pragma unroll
                for (int tc = 0; tc < 4; ++tc)
#pragma unroll
                    for (int r = 0; r < 4; ++r) sm.s[(16 * w + 4 * g + r) * SS + 16 * tc + i16] = sacc[tc][r];
            }
            // ---- 2b. sparsity measure of the wave's own 16 queries (reads only its own rows of S: no barrier)
            if (!(abl & 4)) {
                const int qi = t >> 2, j = t & 3;
                float val[7];
#pragma unroll
                for (int i = 0; i < 7; ++i) {           // samples j, j+4, .., j+24 (25 per query); all 7 gathers in flight
                    const int sidx = j + 4 * i < NU ? j + 4 * i : j;
                    val[i] = sm.s[qi * SS + sm.idx[qi * NU + sidx]];
                }
                float mx = val[0], su = val[0];
#pragma unroll
                for (int i = 1; i < 7; ++i) {
                    const bool ok = j + 4 * i < NU;
                    mx = ok ? fmaxf(mx, val[i]) : mx;
                    su += ok ? val[i] : 0.f;
                }
                mx = r4max(mx);
                su = r4sum(su);
                if (j == 0) sm.m[qi] = mx - su * (1.0f / NT);
            }
            __syncthreads();                       // S and M complete
            // ---- 2c. every wave ranks all 64 queries itself: no partial-count exchange.  Lane j holds M[j]; for every query q
            //      the number of larger measures is ONE vector compare against the broadcast M[q] + a scalar population
            //      count of the lane mask, handed to lane q with v_writelane: 2 VALU + 3 SALU per query instead of ~4.5 VALU
            //      for the per-lane count with the index tie rule.  The window has two EQUAL measures iff the counts do not add
            //      up to 64*63/2 ordered pairs - only then (practically never) the exact tie rule is evaluated per lane.
            int myrank = lane;
            if (!(abl & 8)) {
                const float mq = sm.m[lane];
                int total = 0, rk = 0;
// four compares first (four different scalar-pair destinations), then the four population counts, then the four lane
// writes: one query at a time the chain v_cmp -> s_bcnt1 -> v_writelane is pure latency (~44 cycles per query measured)
#define RANK4(Q4)                                                                                                        \
    {                                                                                                                    \
        float mv[4];                                                                                                     \
        ld4(&sm.m[(Q4)], mv);                                                                                            \
        const unsigned long long b0 = __builtin_amdgcn_ballot_w64(mq > mv[0]);   /* {j : M[j] > M[Q4]} */                \
        const unsigned long long b1 = __builtin_amdgcn_ballot_w64(mq > mv[1]);                                           \
        const unsigned long long b2 = __builtin_amdgcn_ballot_w64(mq > mv[2]);                                           \
        const unsigned long long b3 = __builtin_amdgcn_ballot_w64(mq > mv[3]);                                           \
        const int c0 = __builtin_popcountll(b0), c1 = __builtin_popcountll(b1);                                          \
        const int c2 = __builtin_popcountll(b2), c3 = __builtin_popcountll(b3);                                          \
        total += (c0 + c1) + (c2 + c3);                                                                                  \
        rk = writelane<(Q4) + 0>(rk, c0); rk = writelane<(Q4) + 1>(rk, c1);                                              \
        rk = writelane<(Q4) + 2>(rk, c2); rk = writelane<(Q4) + 3>(rk, c3);                                              \
    }
                RANK4(0) RANK4(4) RANK4(8) RANK4(12) RANK4(16) RANK4(20) RANK4(24) RANK4(28)
                RANK4(32) RANK4(36) RANK4(40) RANK4(44) RANK4(48) RANK4(52) RANK4(56) RANK4(60)
#undef RANK4
                if (total != NT * (NT - 1) / 2) {
                    int cnt = 0;
#pragma unroll 2
                    for (int jj = 0; jj < NT; jj += 4) {
                        float mv[4];
                        ld4(&sm.m[jj], mv);
#pragma unroll
                        for (int u = 0; u < 4; ++u) cnt += (mv[u] > mq) || (mv[u] == mq && (jj + u) < lane);
                    }
                    rk = cnt;
                }
                myrank = rk;
                if (rk < NU) sm.top[w][rk] = lane;
                if (lane >= NU && lane < 32) sm.top[w][lane] = 0;
            }
            // ---- 2d. P = softmax(softmax(scale S[top]) + bias + mask): wave w -> rows 8w..8w+7
            if (!(abl & 16)) {
                const int r = t >> 3, c0 = (t & 7) * 8;
                float p2[8];
                if (r < NU) {
                    const int qrow = sm.top[w][r];
                    float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0, m0 = b0, m1 = b0;
                    if (bias) {      // issue the (L2-resident) bias / mask row loads before the first softmax
                        const float* br = bias + ((size_t)h * NT + qrow) * NT + c0;
                        b0 = *reinterpret_cast<const float4*>(br); b1 = *reinterpret_cast<const float4*>(br + 4);
                    }
                    if (mask && ((wdx / nWw) == (Hres >> 3) - 1 || (wdx % nWw) == nWw - 1)) {   // elsewhere the mask is all zero
                        const float* mr = mask + ((size_t)wdx * NT + qrow) * NT + c0;
                        m0 = *reinterpret_cast<const float4*>(mr); m1 = *reinterpret_cast<const float4*>(mr + 4);
                    }
                    float xr[8], a2[8];
                    const float4 s0 = *reinterpret_cast<const float4*>(&sm.s[qrow * SS + c0]);
                    const float4 s1 = *reinterpret_cast<const float4*>(&sm.s[qrow * SS + c0 + 4]);
                    xr[0] = s0.x * scale; xr[1] = s0.y * scale; xr[2] = s0.z * scale; xr[3] = s0.w * scale;
                    xr[4] = s1.x * scale; xr[5] = s1.y * scale; xr[6] = s1.z * scale; xr[7] = s1.w * scale;
                    float mx = xr[0];
#pragma unroll
                    for (int i = 1; i < 8; ++i) mx = fmaxf(mx, xr[i]);
                    mx = r8max(mx);
                    float sum = 0.f;
#pragma unroll
                    for (int i = 0; i < 8; ++i) { xr[i] = __expf(xr[i] - mx); sum += xr[i]; }
                    sum = __builtin_amdgcn_rcpf(r8sum(sum));
#pragma unroll
                    for (int i = 0; i < 8; ++i) a2[i] = xr[i] * sum;
                    a2[0] += b0.x + m0.x; a2[1] += b0.y + m0.y; a2[2] += b0.z + m0.z; a2[3] += b0.w + m0.w;
                    a2[4] += b1.x + m1.x; a2[5] += b1.y + m1.y; a2[6] += b1.z + m1.z; a2[7] += b1.w + m1.w;
                    float mx2 = a2[0];
#pragma unroll
                    for (int i = 1; i < 8; ++i) mx2 = fmaxf(mx2, a2[i]);
                    mx2 = r8max(mx2);
                    float sum2 = 0.f;
#pragma unroll
                    for (int i = 0; i < 8; ++i) { a2[i] = __expf(a2[i] - mx2); sum2 += a2[i]; }
                    sum2 = __builtin_amdgcn_rcpf(r8sum(sum2));
#pragma unroll
                    for (int i = 0; i < 8; ++i) p2[i] = a2[i] * sum2;
                } else {
                    const float f = (r == NU) ? (1.0f / NT) : 0.f;
#pragma unroll
                    for (int i = 0; i < 8; ++i) p2[i] = f;
                }
                *reinterpret_cast<float4*>(&P[r * SS + c0]) = make_float4(p2[0], p2[1], p2[2], p2[3]);
                *reinterpret_cast<float4*>(&P[r * SS + c0 + 4]) = make_float4(p2[4], p2[5], p2[6], p2[7]);
            }
            // prefetch the next window's rows HERE: younger than every load this window still waits for, so the
            // in-order vmcnt never makes a wait of this window sit out the prefetch's HBM latency
            if (WREG && h == H - 1 && win + (int)gridDim.x < nwin) {
                const float4* xp = reinterpret_cast<const float4*>(x + src_token(win + gridDim.x) * C + qd * CPT);
#pragma unroll
                for (int i = 0; i < CPT / 4; ++i) xnext[i] = xp[i];
            }
            __syncthreads();                       // P complete
            if constexpr (P6O) {
                // the out-projection's weight planes of this head (C/16 column tiles x 3 pieces, 1 KiB each) into the S tile - dead since
                // the softmax read it - while P V runs; they are waited for in front of the next barrier
                const unsigned char* src = reinterpret_cast<const unsigned char*>(wqkv_p) + P6_QKV_BYTES + (size_t)h * P6O_RUNS * 1024 + lane * 16;
#pragma unroll
                for (int r = 0; r < P6O_RUNS / 4; ++r)
                    __builtin_amdgcn_global_load_lds((glb_void_*)(src + (w + 4 * r) * 1024),
                                                     (lds_void_*)(reinterpret_cast<unsigned char*>(sm.s) + (w + 4 * r) * 1024), 16, 0, 0);
            }
            // ---- 2e. O_h = P V_h (32 x 32): one 16x16 tile per wave, into the dead Q tile
            float* O = sm.q;                       // 32 x HS (Q_h was last read before the first barrier of this head)
            if (!(abl & 32)) {
                const int tr = w & 1, tc = w >> 1;
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                float pa[16];
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) ld4(&P[(16 * tr + i16) * SS + 16 * g + 4 * s4], &pa[4 * s4]);
#pragma unroll
                for (int s = 0; s < 16; ++s) acc = mfma16(pa[s], sm.v[(16 * g + s) * HS + 16 * tc + i16], acc);
#pragma unroll
                for (int r = 0; r < 4; ++r) O[(16 * tr + 4 * g + r) * HS + 16 * tc + i16] = acc[r];
            }
            if constexpr (P6O) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                       // O complete (P6O: and the out-projection planes)
            // ---- 3. out-projection partial: oacc += ctx_h Wo[:, 32h:32h+32]^T ; ctx row = selected row or mean row
            if (!(abl & 64)) {
                const int rk = __shfl(myrank, 16 * w + i16);
                const int srow = rk < NU ? rk : NU;
                float a[8];
                ld4(&O[srow * HS + 8 * g], &a[0]);
                ld4(&O[srow * HS + 8 * g + 4], &a[4]);
                if constexpr (WREG6) {
                    u32x4_ ao[3];
                    split8x3_(a, ao[0], ao[1], ao[2]);
                    constexpr int TA[6] = {2, 0, 1, 0, 1, 0}, TB[6] = {0, 2, 1, 1, 0, 0};
#pragma unroll
                    for (int term = 0; term < 6; ++term)
#pragma unroll
                        for (int tn = 0; tn < 2; ++tn) oacc[tn] = mfma_b16(ao[TA[term]], wo6r[tn][TB[term]], oacc[tn]);
                } else if constexpr (P6O) {
                    u32x4_ ao[3];
                    split8x3_(a, ao[0], ao[1], ao[2]);                 // the lane's 8 consecutive k of its context row ARE a bf16 A fragment
                    const unsigned char* const Wimg = reinterpret_cast<const unsigned char*>(sm.s);
                    u32x4_ bo6[C / 16][3];
#pragma unroll
                    for (int tn = 0; tn < C / 16; ++tn)
#pragma unroll
                        for (int pc = 0; pc < 3; ++pc) bo6[tn][pc] = *reinterpret_cast<const u32x4_*>(Wimg + (tn * 3 + pc) * 1024 + lane * 16);
                    constexpr int TA[6] = {2, 0, 1, 0, 1, 0}, TB[6] = {0, 2, 1, 1, 0, 0};
#pragma unroll
                    for (int term = 0; term < 6; ++term)
#pragma unroll
                        for (int tn = 0; tn < C / 16; ++tn) oacc[tn] = mfma_b16(ao[TA[term]], bo6[tn][TB[term]], oacc[tn]);
                } else if constexpr (WREG) {
#pragma unroll
                    for (int s4 = 0; s4 < 2; ++s4) {
#pragma unroll
                        for (int tn = 0; tn < 2; ++tn) oacc[tn] = mfma16(a[4 * s4 + 0], wr_o[tn * 2 + s4].x, oacc[tn]);
#pragma unroll
                        for (int tn = 0; tn < 2; ++tn) oacc[tn] = mfma16(a[4 * s4 + 1], wr_o[tn * 2 + s4].y, oacc[tn]);
#pragma unroll
                        for (int tn = 0; tn < 2; ++tn) oacc[tn] = mfma16(a[4 * s4 + 2], wr_o[tn * 2 + s4].z, oacc[tn]);
#pragma unroll
                        for (int tn = 0; tn < 2; ++tn) oacc[tn] = mfma16(a[4 * s4 + 3], wr_o[tn * 2 + s4].w, oacc[tn]);
                    }
                } else {
                    const float4* wp = wo_p + (size_t)h * (C / 16) * 2 * 64 + lane;
                    float4 b4[C / 16][2];
#pragma unroll
                    for (int tn = 0; tn < C / 16; ++tn) { b4[tn][0] = wp[(tn * 2) * 64]; b4[tn][1] = wp[(tn * 2 + 1) * 64]; }
#pragma unroll
                    for (int s4 = 0; s4 < 2; ++s4) {
#pragma unroll
                        for (int tn = 0; tn < C / 16; ++tn) oacc[tn] = mfma16(a[4 * s4 + 0], b4[tn][s4].x, oacc[tn]);
#pragma unroll
                        for (int tn = 0; tn < C / 16; ++tn) oacc[tn] = mfma16(a[4 * s4 + 1], b4[tn][s4].y, oacc[tn]);
#pragma unroll
                        for (int tn = 0; tn < C / 16; ++tn) oacc[tn] = mfma16(a[4 * s4 + 2], b4[tn][s4].z, oacc[tn]);
#pragma unroll
                        for (int tn = 0; tn < C / 16; ++tn) oacc[tn] = mfma16(a[4 * s4 + 3], b4[tn][s4].w, oacc[tn]);
                    }
                }
                if (SAVE == 1 && !(abl & 128)) {
                    const int row = t >> 2, c8 = (t & 3) * 8;
                    const int rkr = __shfl(myrank, row);
                    const int rr = rkr < NU ? rkr : NU;
                    float* dst = ctx_save + ((size_t)win * NT + row) * C + 32 * h + c8;
                    *reinterpret_cast<float4*>(dst) = *reinterpret_cast<const float4*>(&O[rr * HS + c8]);
                    *reinterpret_cast<float4*>(dst + 4) = *reinterpret_cast<const float4*>(&O[rr * HS + c8 + 4]);
                }
                if (SAVE && !(abl & 128) && w == 0)
                    rank_save[((size_t)win * H + h) * NT + lane] = myrank < NU ? (uint8_t)myrank : (uint8_t)255;
            }
            __syncthreads();                       // O (in the Q tile), K, V, S, P are rewritten by the next head / window
        }

        // ---- 4. epilogue: stage the 64 x C projection through the S tile (dead after the last head's barrier; 64 columns at a
        //         time), add bias + shortcut (x is still in registers), scatter to token order.  Wave-local: wave w stages rows
        //         16w..16w+15 and reads exactly those (tl = 16 w + i16).
        {
            const float sc = dscale ? dscale[bimg] : 1.0f;
            float4* op = reinterpret_cast<float4*>(out + src_tok * C + qd * CPT);
            constexpr int NP = C > 64 ? C / 64 : 1;               // passes of <= 64 columns
            constexpr int PC = C / NP;                            // columns per pass
#pragma unroll
            for (int ps = 0; ps < NP; ++ps) {
#pragma unroll
                for (int tn = 0; tn < PC / 16; ++tn)
#pragma unroll
                    for (int r = 0; r < 4; ++r) sm.s[(16 * w + 4 * g + r) * SS + 16 * tn + i16] = oacc[ps * (PC / 16) + tn][r];
                // channels of this lane: qd CPT .. qd CPT + CPT - 1; in pass ps those inside [ps PC, (ps + 1) PC)
                if (NP == 1 || (qd * CPT) / PC == ps) {
                    const int c0 = qd * CPT - ps * PC;
#pragma unroll
                    for (int i = 0; i < CPT / 4; ++i) {
                        const float4 y = *reinterpret_cast<const float4*>(&sm.s[tl * SS + c0 + 4 * i]);
                        const float4 b4 = bor[i];
                        op[i] = make_float4(xv[i].x + sc * (y.x + b4.x), xv[i].y + sc * (y.y + b4.y), xv[i].z + sc * (y.z + b4.z),
                                            xv[i].w + sc * (y.w + b4.w));
                    }
                }
            }
        }
        if constexpr (WREG) {
#pragma unroll
            for (int i = 0; i < CPT / 4; ++i) xv[i] = xnext[i];
        } else if (win + (int)gridDim.x * NW < nwin) {  // (only reached when the grid is smaller than the window count)
            const float4* xp = reinterpret_cast<const float4*>(x + src_token(win + gridDim.x * NW) * C + qd * CPT);
#pragma unroll
            for (int i = 0; i < CPT / 4; ++i) xv[i] = xp[i];
        }
        // no barrier: the staging rows belong to this wave, and S is next written behind the first barrier of the next window
    }
}

// weight prepack: fragment order for the 16x16x4 MFMA B operand (lane = 16 g + i16 holds B[k = 4 s + g][j = i16]);
// four consecutive k-steps are packed into one float4 so that a wave reads 1 KiB contiguous per instruction.
// The contraction index is permuted so that lane group g owns a CONTIGUOUS k range (the A fragments then come out
// of LDS with ds_read_b128; any permutation is valid as long as A and B agree):
//   wqkv_p[h][j(6)][s4(C/16)][lane(64)] (float4 over r)  = W_m[32h + 16(j&1) + i16][g*C/4 + 4 s4 + r],  m = j>>1
//   wo_p  [h][tn(C/16)][s4(2)][lane(64)] (float4 over r) = Wo[16 tn + i16][32 h + 8 g + 4 s4 + r]
__global__ void prepack_weights_kernel(const float* __restrict__ wq, const float* __restrict__ wk,
                                       const float* __restrict__ wv, const float* __restrict__ wo,
                                       float* __restrict__ wqkv_p, float* __restrict__ wo_p, int C) {
    const int H = C / 32, KS4 = C / 16;
    const int nq = H * 6 * KS4 * 64 * 4, no = H * (C / 16) * 2 * 64 * 4;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < nq) {
        const int r = e & 3, lane = (e >> 2) & 63;
        int rest = e >> 8;
        const int s4 = rest % KS4; rest /= KS4;
        const int j = rest % 6, h = rest / 6;
        const int i16 = lane & 15, g = lane >> 4;
        const float* W = (j >> 1) == 0 ? wq : ((j >> 1) == 1 ? wk : wv);
        wqkv_p[e] = W[(size_t)(32 * h + 16 * (j & 1) + i16) * C + g * (C / 4) + 4 * s4 + r];
    } else if (e < nq + no) {
        const int f = e - nq;
        const int r = f & 3, lane = (f >> 2) & 63;
        int rest = f >> 8;
        const int s4 = rest & 1; rest >>= 1;
        const int tn = rest % (C / 16), h = rest / (C / 16);
        const int i16 = lane & 15, g = lane >> 4;
        wo_p[f] = wo[(size_t)(16 * tn + i16) * C + 32 * h + 8 * g + 4 * s4 + r];
    }
}

// six-term planes of the Q / K / V weights in the fragment order of the P6 kernel: run (1 KiB = 64 lanes x 8 bf16) index
//   ((h (C/64) + ch) 36 + (j 2 + kbl) 3 + piece),   element (lane = 16 g + i16, e) = piece of W_m[32 h + 16 (j & 1) + i16][g C/4 + 8 (2 ch + kbl) + e],
// m = j >> 1 (Q, K, V), pieces by truncation (hi + mid + lo == W exactly).  One thread per (run without piece, lane, e).
// Behind them the out-projection's planes: run ((h C/16 + tn) 3 + piece), element = piece of Wo[16 tn + i16][32 h + 8 g + e].
// C = 32 (one head, one 32-deep k-block; the kernel keeps all of it in registers): run (j 3 + piece) for Q / K / V column tile j, element
// (lane, e) = piece of W_m[16 (j & 1) + i16][8 g + e]; then run ((6 + tn) 3 + piece), element = piece of Wo[16 tn + i16][8 g + e].
__device__ __forceinline__ void prepack6_element_c32(const float* __restrict__ wq, const float* __restrict__ wk, const float* __restrict__ wv,
                                                     const float* __restrict__ wo, uint16_t* __restrict__ out, int t) {
    if (t >= 8 * 512) return;
    const int e = t & 7, lane = (t >> 3) & 63, tile = t >> 9;
    const int i16 = lane & 15, g = lane >> 4;
    float x;
    if (tile < 6) {
        const float* W = (tile >> 1) == 0 ? wq : ((tile >> 1) == 1 ? wk : wv);
        x = W[(16 * (tile & 1) + i16) * 32 + 8 * g + e];
    } else {
        x = wo[(16 * (tile - 6) + i16) * 32 + 8 * g + e];
    }
    const float hi = __uint_as_float(__float_as_uint(x) & 0xffff0000u);
    const float r1 = x - hi;
    const float mid = __uint_as_float(__float_as_uint(r1) & 0xffff0000u);
    const float r2 = r1 - mid;
    uint16_t* o = out + (size_t)(tile * 3) * 512 + lane * 8 + e;
    o[0] = (uint16_t)(__float_as_uint(x) >> 16);
    o[512] = (uint16_t)(__float_as_uint(r1) >> 16);
    o[1024] = (uint16_t)(__float_as_uint(r2) >> 16);
}
__device__ __forceinline__ void prepack6_element(const float* __restrict__ wq, const float* __restrict__ wk, const float* __restrict__ wv,
                                                 const float* __restrict__ wo, uint16_t* __restrict__ out, int C, int t) {
    if (C == 32) { prepack6_element_c32(wq, wk, wv, wo, out, t); return; }
    const int n = (C / 32) * (C / 64) * 12 * 512;              // (h, ch, j, kbl) x 64 lanes x 8 elements
    const int no = (C / 32) * (C / 16) * 512;                  // (h, tn) x 64 lanes x 8 elements
    if (t >= n + no) return;
    float x;
    uint16_t* o;
    if (t >= n) {
        const int f = t - n;
        const int e = f & 7, lane = (f >> 3) & 63;
        const int rest = f >> 9;
        const int tn = rest % (C / 16), h = rest / (C / 16);
        const int i16 = lane & 15, g = lane >> 4;
        x = wo[(size_t)(16 * tn + i16) * C + 32 * h + 8 * g + e];
        o = out + (size_t)3 * n + ((size_t)(h * (C / 16) + tn) * 3) * 512 + lane * 8 + e;
    } else {
        const int e = t & 7, lane = (t >> 3) & 63;
        int rest = t >> 9;
        const int kbl = rest & 1; rest >>= 1;
        const int j = rest % 6; rest /= 6;
        const int ch = rest % (C / 64), h = rest / (C / 64);
        const int i16 = lane & 15, g = lane >> 4;
        const float* W = (j >> 1) == 0 ? wq : ((j >> 1) == 1 ? wk : wv);
        x = W[(size_t)(32 * h + 16 * (j & 1) + i16) * C + g * (C / 4) + 8 * (2 * ch + kbl) + e];
        o = out + (((size_t)(h * (C / 64) + ch) * 12 + (j * 2 + kbl)) * 3) * 512 + lane * 8 + e;
    }
    const float hi = __uint_as_float(__float_as_uint(x) & 0xffff0000u);
    const float r1 = x - hi;
    const float mid = __uint_as_float(__float_as_uint(r1) & 0xffff0000u);
    const float r2 = r1 - mid;
    o[0] = (uint16_t)(__float_as_uint(x) >> 16);
    o[512] = (uint16_t)(__float_as_uint(r1) >> 16);
    o[1024] = (uint16_t)(__float_as_uint(r2) >> 16);
}
__global__ void prepack6_kernel(const float* __restrict__ wq, const float* __restrict__ wk, const float* __restrict__ wv,
                                const float* __restrict__ wo, uint16_t* __restrict__ out, int C) {
    prepack6_element(wq, wk, wv, wo, out, C, blockIdx.x * blockDim.x + threadIdx.x);
}
struct Prepack6Multi { const float* wq[16]; const float* wk[16]; const float* wv[16]; const float* wo[16]; uint16_t* out[16]; int C[16]; };
__global__ void prepack6_multi_kernel(const Prepack6Multi d) {
    const int m = blockIdx.y;
    prepack6_element(d.wq[m], d.wk[m], d.wv[m], d.wo[m], d.out[m], d.C[m], blockIdx.x * blockDim.x + threadIdx.x);
}

// the prepack of every fused block of one model forward in ONE launch: entry = blockIdx.y
constexpr int PREPACK_MULTI_MAX = 16;
struct PrepackMulti { const float* wq[PREPACK_MULTI_MAX]; const float* wk[PREPACK_MULTI_MAX]; const float* wv[PREPACK_MULTI_MAX];
                      const float* wo[PREPACK_MULTI_MAX]; float* wqkv_p[PREPACK_MULTI_MAX]; float* wo_p[PREPACK_MULTI_MAX]; int C[PREPACK_MULTI_MAX]; };
__global__ void prepack_weights_multi_kernel(const PrepackMulti d) {
    const int m = blockIdx.y, C = d.C[m];
    const float* wq = d.wq[m]; const float* wk = d.wk[m]; const float* wv = d.wv[m]; const float* wo = d.wo[m];
    const int H = C / 32, KS4 = C / 16;
    const int nq = H * 6 * KS4 * 64 * 4, no = H * (C / 16) * 2 * 64 * 4;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < nq) {
        const int r = e & 3, lane = (e >> 2) & 63;
        int rest = e >> 8;
        const int s4 = rest % KS4; rest /= KS4;
        const int j = rest % 6, h = rest / 6;
        const int i16 = lane & 15, g = lane >> 4;
        const float* W = (j >> 1) == 0 ? wq : ((j >> 1) == 1 ? wk : wv);
        d.wqkv_p[m][e] = W[(size_t)(32 * h + 16 * (j & 1) + i16) * C + g * (C / 4) + 4 * s4 + r];
    } else if (e < nq + no) {
        const int f = e - nq;
        const int r = f & 3, lane = (f >> 2) & 63;
        int rest = f >> 8;
        const int s4 = rest & 1; rest >>= 1;
        const int tn = rest % (C / 16), h = rest / (C / 16);
        const int i16 = lane & 15, g = lane >> 4;
        d.wo_p[m][f] = wo[(size_t)(16 * tn + i16) * C + 32 * h + 8 * g + 4 * s4 + r];
    }
}

template <int C, int SAVE, int NW = 1, bool P6 = false>
void launch_fused(hipStream_t s, int nwin, const float* x, const float* gamma, const float* beta, const float* wqkv_p,
                  const float* bqkv, const float* wo_p, const float* bo, const uint8_t* idx, const float* bias,
                  const float* mask, const float* dscale, float* out, float* xn_save, float* qkv_save, float* ctx_save,
                  float* stats_save, uint8_t* rank_save, int Hres, int Wres, int shift) {
    const size_t smem = NW * sizeof(FusedSmem<C>);
    static_assert(sizeof(FusedSmem<C>) % 16 == 0, "the second half's tiles start 16-byte aligned");
    if (smem > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fused_window_attn_fwd_kernel<C, SAVE, NW, P6>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    // persistent workgroups (only with register-resident weights): exactly as many as are RESIDENT at once - the occupancy
    // the runtime reports for this code object (registers and LDS together: 2 per CU at 222 VGPRs, although 3 would fit the
    // LDS) - a third workgroup per CU would run alone after the first two have finished.  DHZ_FUSED_WG_PER_CU overrides.
#ifdef DHZ_DIAG
    static const int env_wg = getenv("DHZ_FUSED_WG_PER_CU") ? atoi(getenv("DHZ_FUSED_WG_PER_CU")) : 0;
#else
    constexpr int env_wg = 0;
#endif
    // resident workgroups per CU of this instantiation: queried once (thread-safe function-local static initialisation)
    static const int occ = [&] {
        int q = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&q, reinterpret_cast<const void*>(&fused_window_attn_fwd_kernel<C, SAVE, NW, P6>),
                                                         256 * NW, smem) != hipSuccess || q < 1)
            q = 2;
        return q;
    }();
    const int per_cu = env_wg > 0 ? env_wg : occ;
    const int ncu = dhz_num_cus();
    int grid = (C == 32 && FUSED_PERSIST_C32) ? ncu * per_cu : nwin / NW;
    if (grid > nwin / NW) grid = nwin / NW;
    hipLaunchKernelGGL((fused_window_attn_fwd_kernel<C, SAVE, NW, P6>), dim3(grid), dim3(256 * NW), smem, s, x, gamma, beta,
                       reinterpret_cast<const float4*>(wqkv_p), bqkv, reinterpret_cast<const float4*>(wo_p), bo, idx,
                       bias, mask, dscale, out, xn_save, qkv_save, ctx_save, stats_save, rank_save, Hres, Wres, shift,
                       nwin);
}

}  // namespace

extern "C" int dhz_fused_attn_prepack(const float* wq, const float* wk, const float* wv, const float* wo, float* wqkv_p,
                                      float* wo_p, int C, void* stream) {
    DHZ_REQUIRE(wq && wk && wv && wo && wqkv_p && wo_p, "dhz_fused_attn_prepack: null pointer");
    DHZ_REQUIRE(C == 32 || C == 64 || C == 128, "dhz_fused_attn_prepack: C=%d unsupported (32, 64, 128)", C);
    const int n = 4 * C * C;
    hipLaunchKernelGGL(prepack_weights_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, wq, wk, wv, wo,
                       wqkv_p, wo_p, C);
    DHZ_CHECK_LAUNCH("dhz_fused_attn_prepack");
    return DHZ_OK;
}

extern "C" int dhz_fused_attn_prepack_multi(const float* const* wq, const float* const* wk, const float* const* wv, const float* const* wo,
                                            float* const* wqkv_p, float* const* wo_p, const int* C, int n, void* stream) {
    DHZ_REQUIRE(wq && wk && wv && wo && wqkv_p && wo_p && C && n > 0 && n <= PREPACK_MULTI_MAX,
                "dhz_fused_attn_prepack_multi: null pointer or n=%d outside 1..%d", n, PREPACK_MULTI_MAX);
    PrepackMulti d = {};
    int cmax = 0;
    for (int i = 0; i < n; ++i) {
        DHZ_REQUIRE(wq[i] && wk[i] && wv[i] && wo[i] && wqkv_p[i] && wo_p[i], "dhz_fused_attn_prepack_multi: entry %d: null pointer", i);
        DHZ_REQUIRE(C[i] == 32 || C[i] == 64 || C[i] == 128, "dhz_fused_attn_prepack_multi: entry %d: C=%d unsupported (32, 64, 128)", i, C[i]);
        d.wq[i] = wq[i]; d.wk[i] = wk[i]; d.wv[i] = wv[i]; d.wo[i] = wo[i]; d.wqkv_p[i] = wqkv_p[i]; d.wo_p[i] = wo_p[i]; d.C[i] = C[i];
        cmax = C[i] > cmax ? C[i] : cmax;
    }
    hipLaunchKernelGGL(prepack_weights_multi_kernel, dim3((4 * cmax * cmax + 255) / 256, n), dim3(256), 0, (hipStream_t)stream, d);
    DHZ_CHECK_LAUNCH("dhz_fused_attn_prepack_multi");
    return DHZ_OK;
}

static int fused_fwd_impl(bool p6, const float* x, const float* gamma, const float* beta, const float* wqkv_p,
                          const float* bqkv, const float* wo_p, const float* bo, const uint8_t* idx,
                          const float* bias, const float* mask, const float* drop_scale, float* out,
                          float* xn_save, float* qkv_save, float* ctx_save, float* stats_save,
                          uint8_t* rank_save, int B, int Hres, int Wres, int C, int shift, void* stream) {
    DHZ_REQUIRE(x && gamma && beta && wqkv_p && bqkv && wo_p && bo && idx && out, "dhz_fused_window_attn_fwd: null pointer");
    DHZ_REQUIRE(C == 32 || C == 64 || C == 128, "dhz_fused_window_attn_fwd: C=%d unsupported (32, 64, 128)", C);
    DHZ_REQUIRE(B > 0 && Hres % 8 == 0 && Wres % 8 == 0 && Hres >= 8 && Wres >= 8 && shift >= 0 && shift < 8,
                "dhz_fused_window_attn_fwd: bad geometry %dx%d shift %d", Hres, Wres, shift);
    // three modes: inference (no save pointer), training for the backward kernel chain (all five), training for the fused backward
    // (rank_save only: everything else is recomputed there)
    const bool any4 = xn_save || qkv_save || ctx_save || stats_save;
    const bool all4 = xn_save && qkv_save && ctx_save && stats_save;
    DHZ_REQUIRE(!any4 || (all4 && rank_save),
                "dhz_fused_window_attn_fwd: save buffers: none, rank_save alone, or all five");
    const int save = all4 ? 1 : (rank_save ? 2 : 0);
    DHZ_REQUIRE(!mask || shift > 0, "dhz_fused_window_attn_fwd: a mask is only meaningful for shifted windows");
    hipStream_t s = (hipStream_t)stream;
    const int nwin = B * (Hres / 8) * (Wres / 8);
#define GO(CC)                                                                                                      \
    do {                                                                                                            \
        if (save == 1) launch_fused<CC, 1>(s, nwin, x, gamma, beta, wqkv_p, bqkv, wo_p, bo, idx, bias, mask, drop_scale, \
                                           out, xn_save, qkv_save, ctx_save, stats_save, rank_save, Hres, Wres, shift);   \
        else if (save == 2) launch_fused<CC, 2>(s, nwin, x, gamma, beta, wqkv_p, bqkv, wo_p, bo, idx, bias, mask, drop_scale, \
                                                out, nullptr, nullptr, nullptr, nullptr, rank_save, Hres, Wres, shift);      \
        else launch_fused<CC, 0>(s, nwin, x, gamma, beta, wqkv_p, bqkv, wo_p, bo, idx, bias, mask, drop_scale, out, \
                                 nullptr, nullptr, nullptr, nullptr, nullptr, Hres, Wres, shift);                \
    } while (0)
    // Two windows per 512-thread workgroup (NW = 2) - what the LDS leaves room for beside resident bf16 weight planes - is a DIAGNOSTIC
    // instance (tools/diag_build.sh + DHZ_FUSED_PAIR=2): measured 320 -> 380 us (inference) / 424 -> 458 (training) at C = 64, 128 x 128 - two
    // windows in flight instead of three cost what six-term projections could return (profiles/r05_fused_attn_wall.txt).  Not dispatched.
#ifdef DHZ_DIAG
    static const int pair_env = getenv("DHZ_FUSED_PAIR") ? atoi(getenv("DHZ_FUSED_PAIR")) : 0;
#else
    constexpr int pair_env = 0;
#endif
    if (C == 64 && pair_env == 2 && nwin % 2 == 0) {
        if (save == 1) launch_fused<64, 1, 2>(s, nwin, x, gamma, beta, wqkv_p, bqkv, wo_p, bo, idx, bias, mask, drop_scale, out, xn_save,
                                              qkv_save, ctx_save, stats_save, rank_save, Hres, Wres, shift);
        else if (save == 2) launch_fused<64, 2, 2>(s, nwin, x, gamma, beta, wqkv_p, bqkv, wo_p, bo, idx, bias, mask, drop_scale, out,
                                                   nullptr, nullptr, nullptr, nullptr, rank_save, Hres, Wres, shift);
        else launch_fused<64, 0, 2>(s, nwin, x, gamma, beta, wqkv_p, bqkv, wo_p, bo, idx, bias, mask, drop_scale, out, nullptr, nullptr,
                                    nullptr, nullptr, nullptr, Hres, Wres, shift);
    } else if (p6) {
        // C = 64: all four weight products six-term, planes by LDS-DMA; C = 128: Q / K / V six-term (two 64-channel halves per head), the
        // out-projection stays on the fp32 pipe with wo_p (its planes, 24 KiB per head, do not fit the S tile); C = 32: all four six-term
        // with the planes in registers of the persistent workgroups
        DHZ_REQUIRE(C == 32 || C == 64 || C == 128, "dhz_fused_window_attn_fwd6: C=%d unsupported (32, 64, 128)", C);
#define GO6(CC)                                                                                                             \
    do {                                                                                                                    \
        if (save == 1) launch_fused<CC, 1, 1, true>(s, nwin, x, gamma, beta, wqkv_p, bqkv, wo_p, bo, idx, bias, mask, drop_scale, out, xn_save, \
                                                    qkv_save, ctx_save, stats_save, rank_save, Hres, Wres, shift);         \
        else if (save == 2) launch_fused<CC, 2, 1, true>(s, nwin, x, gamma, beta, wqkv_p, bqkv, wo_p, bo, idx, bias, mask, drop_scale, out, \
                                                         nullptr, nullptr, nullptr, nullptr, rank_save, Hres, Wres, shift); \
        else launch_fused<CC, 0, 1, true>(s, nwin, x, gamma, beta, wqkv_p, bqkv, wo_p, bo, idx, bias, mask, drop_scale, out, nullptr, nullptr, \
                                          nullptr, nullptr, nullptr, Hres, Wres, shift);                                   \
    } while (0)
        if (C == 32) GO6(32); else if (C == 64) GO6(64); else GO6(128);
#undef GO6
    } else if (C == 32) GO(32); else if (C == 64) GO(64); else GO(128);
#undef GO
    DHZ_CHECK_LAUNCH("dhz_fused_window_attn_fwd");
    return DHZ_OK;
}

extern "C" int dhz_fused_attn_prepack6_multi(const float* const* wq, const float* const* wk, const float* const* wv, const float* const* wo,
                                             void* const* wqkv6_p, const int* C, int n, void* stream) {
    DHZ_REQUIRE(wq && wk && wv && wo && wqkv6_p && C && n > 0 && n <= 16, "dhz_fused_attn_prepack6_multi: null pointer or n=%d outside 1..16", n);
    Prepack6Multi d = {};
    int tmax = 0;
    for (int i = 0; i < n; ++i) {
        DHZ_REQUIRE(wq[i] && wk[i] && wv[i] && wo[i] && wqkv6_p[i], "dhz_fused_attn_prepack6_multi: entry %d: null pointer", i);
        DHZ_REQUIRE(C[i] == 32 || C[i] == 64 || C[i] == 128, "dhz_fused_attn_prepack6_multi: entry %d: C=%d unsupported (32, 64, 128)", i, C[i]);
        d.wq[i] = wq[i]; d.wk[i] = wk[i]; d.wv[i] = wv[i]; d.wo[i] = wo[i]; d.out[i] = (uint16_t*)wqkv6_p[i]; d.C[i] = C[i];
        const int t = C[i] == 32 ? 8 * 512 : (C[i] / 32) * (C[i] / 64) * 12 * 512 + (C[i] / 32) * (C[i] / 16) * 512;
        tmax = t > tmax ? t : tmax;
    }
    hipLaunchKernelGGL(prepack6_multi_kernel, dim3((tmax + 255) / 256, n), dim3(256), 0, (hipStream_t)stream, d);
    DHZ_CHECK_LAUNCH("dhz_fused_attn_prepack6_multi");
    return DHZ_OK;
}

extern "C" int dhz_fused_window_attn_fwd(const float* x, const float* gamma, const float* beta, const float* wqkv_p,
                                         const float* bqkv, const float* wo_p, const float* bo, const uint8_t* idx,
                                         const float* bias, const float* mask, const float* drop_scale, float* out,
                                         float* xn_save, float* qkv_save, float* ctx_save, float* stats_save,
                                         uint8_t* rank_save, int B, int Hres, int Wres, int C, int shift, void* stream) {
    return fused_fwd_impl(false, x, gamma, beta, wqkv_p, bqkv, wo_p, bo, idx, bias, mask, drop_scale, out, xn_save, qkv_save, ctx_save,
                          stats_save, rank_save, B, Hres, Wres, C, shift, stream);
}

extern "C" int dhz_fused_window_attn_fwd6(const float* x, const float* gamma, const float* beta, const void* wqkv6_p,
                                          const float* bqkv, const float* wo_p, const float* bo, const uint8_t* idx,
                                          const float* bias, const float* mask, const float* drop_scale, float* out,
                                          float* xn_save, float* qkv_save, float* ctx_save, float* stats_save,
                                          uint8_t* rank_save, int B, int Hres, int Wres, int C, int shift, void* stream) {
    return fused_fwd_impl(true, x, gamma, beta, (const float*)wqkv6_p, bqkv, wo_p, bo, idx, bias, mask, drop_scale, out, xn_save, qkv_save,
                          ctx_save, stats_save, rank_save, B, Hres, Wres, C, shift, stream);
}

extern "C" int dhz_fused_attn_prepack6(const float* wq, const float* wk, const float* wv, const float* wo, void* wqkv6_p, int C, void* stream) {
    DHZ_REQUIRE(wq && wk && wv && wo && wqkv6_p, "dhz_fused_attn_prepack6: null pointer");
    DHZ_REQUIRE(C == 32 || C == 64 || C == 128, "dhz_fused_attn_prepack6: C=%d unsupported (32, 64, 128)", C);
    const int n = C == 32 ? 8 * 512 : (C / 32) * (C / 64) * 12 * 512 + (C / 32) * (C / 16) * 512;
    hipLaunchKernelGGL(prepack6_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, wq, wk, wv, wo, (uint16_t*)wqkv6_p, C);
    DHZ_CHECK_LAUNCH("dhz_fused_attn_prepack6");
    return DHZ_OK;
}
