// K3 - ProbSparse window attention core (forward + backward) for gfx950.
//
// One 256-thread workgroup (4 wave64) per (window, head).  Q, K, V of a window-head (3 x 64 x d fp32)
// are staged once into LDS; everything else - the sampled-score sparsity measure, the top-u selection,
// the double softmax, bias/mask addition, A.V and the mean(V) fill - happens on chip.  The reference
// materialises K_sample[B_,H,64,25,d], bias.repeat(B_) and mask.repeat(B) in HBM for the same result
// (ATT:88-110, 229-230, 246-258).
//
// Contractions run on the fp32-input matrix pipe (v_mfma_f32_16x16x4_f32):
//   S  = Q K^T   (64x64xd)  - dense; the 25 sampled scores per query are then *read out of S in LDS*
//                             and the 25 selected rows of S are exactly Q_reduce K^T (ATT:150), so the
//                             reference's two score products collapse into one MFMA product.
//   O  = P V     (32x64xd)  - 25 selected rows padded to 32; row 25 of P is the constant 1/64 so that
//                             row 25 of O is mean(V) (ATT:168-172) for free.
#include "common.h"

namespace {

constexpr int NT = 64;     // tokens per window
constexpr int NU = 25;     // selected queries / sampled keys
constexpr int SS = 68;     // row stride of the 64-wide score tiles

#ifndef PSF_WG32
#define PSF_WG32 3        // four fit (36.3 KB); measured no different from three in the training step (38.15 / 38.14 ms)
#endif
#ifndef PSF_WG64
#define PSF_WG64 3        // config 4: 44.19 ms per step against 44.52 with two
#endif

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t pack_top16(float x1, float x0) {           // bf16(x1) : bf16(x0) of bf16-representable values
    return __builtin_amdgcn_perm(__float_as_uint(x1), __float_as_uint(x0), 0x07060302u);
}
// 16 x 16 tile of A . B^T over a contraction of 64 for two row-major fp32 LDS tiles that hold bf16-representable values (bf16
// storage: Q, K, V, dO rows): the products on v_mfma_f32_16x16x32_bf16, exact, fp32 accumulation.  Rows that hold other fp32
// values (the summed dO row of the mean(V) path) are TRUNCATED to bf16 - only where the result is not used.
__device__ __forceinline__ bf16x8_t frag_bf16(const float* row, int half, int g) {
    const f32x4 lo = *reinterpret_cast<const f32x4*>(row + 32 * half + 8 * g);
    const f32x4 hi = *reinterpret_cast<const f32x4*>(row + 32 * half + 8 * g + 4);
    u32x4_t r;
    r[0] = pack_top16(lo[1], lo[0]); r[1] = pack_top16(lo[3], lo[2]);
    r[2] = pack_top16(hi[1], hi[0]); r[3] = pack_top16(hi[3], hi[2]);
    return __builtin_bit_cast(bf16x8_t, r);
}
__device__ __forceinline__ f32x4 tile_mma_bf16_k64(const float* A, int lda, const float* B, int ldb, int i16, int g, f32x4 acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_bf16(A + i16 * lda, 0, g), frag_bf16(B + i16 * ldb, 0, g), acc, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_bf16(A + i16 * lda, 1, g), frag_bf16(B + i16 * ldb, 1, g), acc, 0, 0, 0);
}

template <int D>
struct FwdSmem {
    static constexpr int DS = D + 4;
    float s[NT * SS];      // d = 64: Q as staged (64 x DS) -> S (after every wave holds its Q fragments); S -> O (32 x DS)
    float k[NT * DS];      // K -> rank partial counts; d = 64: -> P (32 x SS)
    float v[NT * DS];
    float q[D == 32 ? NT * DS : D == 16 ? 32 * SS : 4];   // d = 32: Q, later P (its own tile: one barrier less per window-head, and three
                                      // workgroups fit a CU either way; parking Q in the S tile measured 6 % slower there); d = 16: Q parks
                                      // in the S tile like d = 64, but P (32 x SS) does not fit the 64 x 20 K tile and gets this one
    float m[NT];
    int top[32];
    uint8_t rank[NT];
};                         // 45.5 KB at d = 32, 52.7 KB at d = 64 (66.6 before Q / P shared tiles): three workgroups per CU

__device__ __forceinline__ float row8_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 1));
    v = fmaxf(v, __shfl_xor(v, 2));
    v = fmaxf(v, __shfl_xor(v, 4));
    return v;
}
__device__ __forceinline__ float row8_sum(float v) {
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    return v;
}

// Double softmax of one selected row, 8 columns per thread (8 threads per row):
//   p1 = softmax(x);  a = p1 + bias + mask;  p2 = softmax(a)          (ATT:195, 229, 251-258, 262)
__device__ __forceinline__ void double_softmax8(const float* x, const float* brow, const float* mrow, float* p1,
                                                float* p2) {
    float mx = x[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) mx = fmaxf(mx, x[i]);
    mx = row8_max(mx);
    float e[8], sum = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { e[i] = __expf(x[i] - mx); sum += e[i]; }
    sum = __builtin_amdgcn_rcpf(row8_sum(sum));          // v_exp_f32 / v_rcp_f32: ~1 ulp each, 10x fewer instructions
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

template <int D, typename T>
__global__ __launch_bounds__(256) void ps_attn_fwd_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                          const T* __restrict__ v, int ld,
                                                          const uint8_t* __restrict__ idx,
                                                          const float* __restrict__ bias,
                                                          const float* __restrict__ mask, T* __restrict__ out,
                                                          int ldo, uint8_t* __restrict__ rank_out, int H, int nW, int nwh) {
    constexpr int DS = D + 4;
    constexpr int F = D / 4;            // float4 per row
    constexpr int RPP = 256 / F;        // rows per load pass
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    FwdSmem<D>& sm = *reinterpret_cast<FwdSmem<D>*>(smem_raw);
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    // Persistent workgroups walk the window-heads wh = blockIdx.x, + gridDim.x, ...; the Q, K, V rows of the NEXT window-head are
    // fetched into registers (native vectors, so that they stay there) while the current one runs its six barrier-separated
    // phases - otherwise every window-head opens with the full HBM latency exposed (the backward kernel does the same).
    constexpr int NR = NT / RPP;                       // staged rows per thread and tensor
    f32x4 pq[NR], pk[NR], pv[NR];
    auto prefetch = [&](int wh_) {
        const size_t tk0 = (size_t)(wh_ / H) * NT;
        const int hh = wh_ % H, c4 = t % F;
#pragma unroll
        for (int p = 0; p < NR; ++p) {
            const size_t g = (tk0 + p * RPP + t / F) * ld + hh * D + c4 * 4;
            pq[p] = ld4v(q + g);
            pk[p] = ld4v(k + g);
            pv[p] = ld4v(v + g);
        }
    };
    if ((int)blockIdx.x < nwh) prefetch(blockIdx.x);
#pragma unroll 1
    for (int wh = blockIdx.x; wh < nwh; wh += gridDim.x) {
    const int b = wh / H, h = wh % H;
    const size_t tok0 = (size_t)b * NT;
    __syncthreads();                                   // the previous window-head's scatter has read O / rank

    // ---- stage Q,K,V into LDS, then start the next window-head's loads
    {
        const int c4 = t % F;
#pragma unroll
        for (int p = 0; p < NR; ++p) {
            const int row = p * RPP + t / F;
            *reinterpret_cast<f32x4*>(&(D == 32 ? sm.q : sm.s)[row * DS + c4 * 4]) = pq[p];     // d = 64: Q parks in the S tile
            *reinterpret_cast<f32x4*>(&sm.k[row * DS + c4 * 4]) = pk[p];
            *reinterpret_cast<f32x4*>(&sm.v[row * DS + c4 * 4]) = pv[p];
        }
        if (wh + (int)gridDim.x < nwh) prefetch(wh + gridDim.x);
    }
    __syncthreads();

    // ---- S = Q K^T : wave w owns rows 16w..16w+15, all 64 columns (4 tiles)
    if constexpr (sizeof(T) == 2 && D == 64) {
        // bf16 storage: the fp32 values in the tiles ARE bf16 values, so the same products run on v_mfma_f32_16x16x32_bf16 (16 x
        // the rate of the fp32 pipe; exact products, fp32 accumulation - only the order of the sums differs): the fragment of a
        // lane is 8 consecutive channels of its row, packed from the top halves of the fp32 words
        const int i = lane & 15, g = lane >> 4;
        auto frag = [&](const float* row, int half) { return frag_bf16(row, half, g); };
        const bf16x8_t qa0 = frag(&sm.s[(16 * w + i) * DS], 0), qa1 = frag(&sm.s[(16 * w + i) * DS], 1);
        __syncthreads();                               // every wave holds its Q fragments: the tile is free for S
#pragma unroll
        for (int tc = 0; tc < 4; ++tc) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa0, frag(&sm.k[(16 * tc + i) * DS], 0), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa1, frag(&sm.k[(16 * tc + i) * DS], 1), acc, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) sm.s[(16 * w + 4 * g + j) * SS + 16 * tc + i] = acc[j];
        }
    } else {
        const int i = lane & 15, g = lane >> 4;
        float a[D / 4];
#pragma unroll
        for (int s = 0; s < D / 4; ++s) a[s] = (D == 32 ? sm.q : sm.s)[(16 * w + i) * DS + 4 * s + g];
        if (D != 32) __syncthreads();                  // every wave holds its Q fragments: the tile is free for S
#pragma unroll
        for (int tc = 0; tc < 4; ++tc) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < D / 4; ++s) acc = mfma16(a[s], sm.k[(16 * tc + i) * DS + 4 * s + g], acc);
#pragma unroll
            for (int j = 0; j < 4; ++j) sm.s[(16 * w + 4 * g + j) * SS + 16 * tc + i] = acc[j];
        }
    }
    __syncthreads();

    // ---- sparsity measure M[q] = max_s S[q,idx[q,s]] - sum_s S[q,idx[q,s]] / 64      (ATT:117)
    {
        const int qi = t >> 2, j = t & 3;
        float mx = -INFINITY, su = 0.f;
        for (int s = j; s < NU; s += 4) {
            const float val = sm.s[qi * SS + idx[qi * NU + s]];       // the 1.6 KB sample table: L1 / L2 hits
            mx = fmaxf(mx, val);
            su += val;
        }
        mx = fmaxf(mx, __shfl_xor(mx, 1)); mx = fmaxf(mx, __shfl_xor(mx, 2));
        su += __shfl_xor(su, 1); su += __shfl_xor(su, 2);
        if (j == 0) sm.m[qi] = mx - su * (1.0f / NT);
    }
    __syncthreads();

    // ---- top-u by rank: rank[q] = #{j : M[j] > M[q] or (M[j] == M[q] and j < q)}    (ATT:122)
    {
        const int qi = t & 63;
        const float m = sm.m[qi];
        int cnt = 0;
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) {
            const int j = 16 * w + jj;
            const float mj = sm.m[j];
            cnt += (mj > m) || (mj == m && j < qi);
        }
        reinterpret_cast<int*>(sm.k)[w * NT + qi] = cnt;       // K is dead since S is complete
    }
    __syncthreads();
    if (t < NT) {
        const int* part = reinterpret_cast<const int*>(sm.k);
        const int r = part[t] + part[NT + t] + part[2 * NT + t] + part[3 * NT + t];
        sm.rank[t] = r < NU ? (uint8_t)r : (uint8_t)255;
        if (r < NU) sm.top[r] = t;
    } else if (t < NT + 32 - NU) {
        sm.top[NU + t - NT] = 0;
    }
    __syncthreads();

    // ---- P = softmax(softmax(scale * S[top]) + bias[top] + mask[top])  -> LDS (over the dead K tile; the partial counts in its
    //      first KB were consumed before the barrier above)
    float* P = D == 64 ? sm.k : sm.q;
    {
        const int r = t >> 3, c0 = (t & 7) * 8;
        float p2[8];
        if (r < NU) {
            const int qrow = sm.top[r];
            const float scale = rsqrtf((float)D);
            float x[8], p1[8];
            const float4 s0 = *reinterpret_cast<const float4*>(&sm.s[qrow * SS + c0]);
            const float4 s1 = *reinterpret_cast<const float4*>(&sm.s[qrow * SS + c0 + 4]);
            x[0] = s0.x * scale; x[1] = s0.y * scale; x[2] = s0.z * scale; x[3] = s0.w * scale;
            x[4] = s1.x * scale; x[5] = s1.y * scale; x[6] = s1.z * scale; x[7] = s1.w * scale;
            const float* brow = bias ? bias + ((size_t)h * NT + qrow) * NT + c0 : nullptr;
            const float* mrow = mask ? mask + ((size_t)(b % nW) * NT + qrow) * NT + c0 : nullptr;
            double_softmax8(x, brow, mrow, p1, p2);
        } else {
            const float f = (r == NU) ? (1.0f / NT) : 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) p2[i] = f;
        }
        *reinterpret_cast<float4*>(&P[r * SS + c0]) = make_float4(p2[0], p2[1], p2[2], p2[3]);
        *reinterpret_cast<float4*>(&P[r * SS + c0 + 4]) = make_float4(p2[4], p2[5], p2[6], p2[7]);
    }
    __syncthreads();

    // ---- O = P V  (32 x D): tiles (tr, tc), tr = w&1, tc = (w>>1) + 2i ; O overwrites the dead S tile
    float* O = sm.s;
    {
        const int i = lane & 15, g = lane >> 4;
        const int tr = w & 1;
#pragma unroll
        for (int ii = 0; ii < (D >= 32 ? D / 32 : 1); ++ii) {
            const int tc = (w >> 1) + 2 * ii;
            if (D == 16 && tc > 0) break;                 // d = 16: two tiles, waves 0 and 1
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            acc = tile_mma<16>(P + 16 * tr * SS, SS, 1, sm.v + 16 * tc, 1, DS, acc);
#pragma unroll
            for (int j = 0; j < 4; ++j) O[(16 * tr + 4 * g + j) * DS + 16 * tc + i] = acc[j];
        }
    }
    __syncthreads();

    // ---- scatter: selected queries get their attention row, the others mean(V) (row 25)
    {
        const int c4 = t % F;
#pragma unroll
        for (int p = 0; p < NT / RPP; ++p) {
            const int row = p * RPP + t / F;
            const int r = sm.rank[row] < NU ? sm.rank[row] : NU;
            st4(out + (tok0 + row) * ldo + h * D + c4 * 4, *reinterpret_cast<const float4*>(&O[r * DS + c4 * 4]));
        }
        if (t < NT / 4) reinterpret_cast<uint32_t*>(rank_out + (size_t)wh * NT)[t] =
            reinterpret_cast<const uint32_t*>(sm.rank)[t];
    }
    }   // window-head loop
}

// ------------------------------------------------------------------------------------------------ backward
template <int D>
struct BwdSmem {
    static constexpr int DS = D + 4;
    float qr[32 * DS];     // Q[top] rows (25 live)
    float k[NT * DS];      // K        ; later dK staging
    float v[NT * DS];      // V        ; later dV staging
    float dor[32 * DS];    // dO[top] rows, row 25 = sum of dO over unselected queries
    float p1[32 * SS];     // scores -> P1 ; later dQ[top] staging (32 x DS)
    float p2[32 * SS];     // P2 ; later dA (gradient w.r.t. the bias-added logits) for the bias-gradient owners
    float ds[32 * SS];     // dP2 -> dS ; before that (first phase only) the 4 x D partial column sums of dO
    float acc[D != 64 ? NT * NT : 4];   // d = 16 / 32: per-workgroup bias-gradient accumulator (d = 64 keeps it in registers, see the kernel)
    int top[32];
    uint8_t rank[NT];
};

template <int D, bool HAS_BIAS, typename T>
__global__ __launch_bounds__(256) void ps_attn_bwd_kernel(
    const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v, int ld,
    const float* __restrict__ bias, const float* __restrict__ mask, const uint8_t* __restrict__ rank_in,
    const T* __restrict__ dout, int ldo, T* __restrict__ dq, T* __restrict__ dk, T* __restrict__ dv,
    int ldg, float* __restrict__ dbias_part, int B_, int H, int nW) {
    constexpr int DS = D + 4;
    constexpr int F = D / 4;
    constexpr int RPP = 256 / F;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    BwdSmem<D>& sm = *reinterpret_cast<BwdSmem<D>*>(smem_raw);
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int i16 = lane & 15, g = lane >> 4;
    const int h = blockIdx.x % H;
    const int bstep = gridDim.x / H;
    const float scale = rsqrtf((float)D);

    // Bias gradient of this workgroup's window-heads.  d = 64: summed in REGISTERS - thread t owns the 16 elements (row w + 4 i,
    // column lane) of the 64 x 64 table; without the 16 KB LDS array of round 1 the workgroup needs 78 KB and two fit a CU
    // instead of one.
    // At d = 32 the LDS accumulator of round 1 stays (two workgroups per CU either way; the register form measured 14 % slower
    // there: 16 rank tests and up to 16 LDS reads per thread and window-head instead of 8 read-modify-writes in 200 threads).
    constexpr bool REG_ACC = D == 64;
    float accr[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) accr[i] = 0.f;
    if (HAS_BIAS && !REG_ACC) {
        for (int e = t; e < NT * NT / 4; e += 256) reinterpret_cast<float4*>(sm.acc)[e] = make_float4(0.f, 0.f, 0.f, 0.f);
    }

    // The window-heads of this workgroup are processed one after the other, ~9 barrier-separated phases each.  Their
    // inputs (K, V, Q, dO rows and the ranks) are prefetched into registers one window-head ahead - native vectors, so
    // that they stay in registers - otherwise every iteration opens with ~2 us of exposed global-load latency.
    constexpr int NR = NT / RPP;                       // staged rows per thread and tensor
    f32x4 pk[NR], pv[NR], pq[NR], pg[NR];
    uint8_t prank = 255;
    auto prefetch = [&](int b) {
        const size_t tok0 = (size_t)b * NT;
        const int c4 = t % F;
#pragma unroll
        for (int p = 0; p < NR; ++p) {
            const int row = p * RPP + t / F;
            const size_t gi = (tok0 + row) * ld + h * D + c4 * 4;
            pk[p] = ld4v(k + gi);
            pv[p] = ld4v(v + gi);
            pq[p] = ld4v(q + gi);
            pg[p] = ld4v(dout + (tok0 + row) * ldo + h * D + c4 * 4);
        }
        if (t < NT) prank = rank_in[((size_t)b * H + h) * NT + t];
    };
    if ((int)(blockIdx.x / H) < B_) prefetch(blockIdx.x / H);

    for (int b = blockIdx.x / H; b < B_; b += bstep) {
        const size_t tok0 = (size_t)b * NT;
        __syncthreads();   // previous iteration's staging reads are done
        if (t < NT) {
            const uint8_t r = prank;
            sm.rank[t] = r;
            if (r < NU) sm.top[r] = t;
        } else if (t < NT + 32 - NU) {
            sm.top[NU + t - NT] = 0;
        }
        // zero the padding rows of Q[top] / dO[top]
        for (int e = t; e < (32 - NU) * DS; e += 256) { sm.qr[NU * DS + e] = 0.f; sm.dor[NU * DS + e] = 0.f; }
        __syncthreads();

        // ---- stage K, V, Q[top], dO[top]; reduce dO over the unselected queries (mean(V) path)
        {
            const int c4 = t % F;
            f32x4 dm = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int p = 0; p < NR; ++p) {
                const int row = p * RPP + t / F;
                *reinterpret_cast<f32x4*>(&sm.k[row * DS + c4 * 4]) = pk[p];
                *reinterpret_cast<f32x4*>(&sm.v[row * DS + c4 * 4]) = pv[p];
                const int r = sm.rank[row];
                if (r < NU) {
                    *reinterpret_cast<f32x4*>(&sm.qr[r * DS + c4 * 4]) = pq[p];
                    *reinterpret_cast<f32x4*>(&sm.dor[r * DS + c4 * 4]) = pg[p];
                } else {
                    dm += pg[p];
                }
            }
            if (b + bstep < B_) prefetch(b + bstep);          // in flight during the rest of this window-head
#pragma unroll
            for (int o = F; o < 64; o <<= 1) {
                dm[0] += __shfl_xor(dm[0], o); dm[1] += __shfl_xor(dm[1], o);
                dm[2] += __shfl_xor(dm[2], o); dm[3] += __shfl_xor(dm[3], o);
            }
            if (lane < F) *reinterpret_cast<f32x4*>(&sm.ds[w * D + lane * 4]) = dm;
        }
        __syncthreads();
        if (t < D) sm.dor[NU * DS + t] = sm.ds[t] + sm.ds[D + t] + sm.ds[2 * D + t] + sm.ds[3 * D + t];

        // ---- recompute scores of the selected rows: Sr = Q[top] K^T (32 x 64), 8 tiles, 2 per wave
        {
            const int tr = w & 1;
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
                const int tc = (w >> 1) + 2 * ii;
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                if constexpr (sizeof(T) == 2 && D == 64) acc = tile_mma_bf16_k64(sm.qr + 16 * tr * DS, DS, sm.k + 16 * tc * DS, DS, i16, g, acc);
                else acc = tile_mma<D / 4>(sm.qr + 16 * tr * DS, DS, 1, sm.k + 16 * tc * DS, DS, 1, acc);
#pragma unroll
                for (int j = 0; j < 4; ++j) sm.p1[(16 * tr + 4 * g + j) * SS + 16 * tc + i16] = acc[j];
            }
        }
        __syncthreads();

        // ---- P1, P2
        {
            const int r = t >> 3, c0 = (t & 7) * 8;
            float p1[8], p2[8];
            if (r < NU) {
                const int qrow = sm.top[r];
                float x[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) x[i] = sm.p1[r * SS + c0 + i] * scale;
                const float* brow = bias ? bias + ((size_t)h * NT + qrow) * NT + c0 : nullptr;
                const float* mrow = mask ? mask + ((size_t)(b % nW) * NT + qrow) * NT + c0 : nullptr;
                double_softmax8(x, brow, mrow, p1, p2);
            } else {
                const float f = (r == NU) ? (1.0f / NT) : 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) { p1[i] = 0.f; p2[i] = f; }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) { sm.p1[r * SS + c0 + i] = p1[i]; sm.p2[r * SS + c0 + i] = p2[i]; }
        }
        __syncthreads();

        // ---- dV = P2^T dO[top]  (64 x D; row 25 of P2 = 1/64 carries the mean(V) path)
        //      dP2 = dO[top] V^T (32 x 64)
        f32x4 accv[D / 16];
        f32x4 accp[2];
        {
            // dV tiles: (tn, tc): tn = w (rows 16w..), tc = 0..D/16-1
#pragma unroll
            for (int tc = 0; tc < D / 16; ++tc) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                // A(i=n, k=r) = P2[r][16w+n] ; B(k=r, j=c) = dOr[r][16tc + c]
                accv[tc] = tile_mma<8>(sm.p2 + 16 * w, 1, SS, sm.dor + 16 * tc, 1, DS, acc);
            }
            const int tr = w & 1;
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
                const int tc = (w >> 1) + 2 * ii;
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                // A(i=r, k=e) = dOr[16tr + r][e] ; B(k=e, j=n) = V[16tc + n][e]
                // (bf16 storage: row 25 of dOr - the fp32 sum of the unselected queries' dO - is truncated here; its dP2 row only meets
                // P1 = 0 below)
                if constexpr (sizeof(T) == 2 && D == 64) accp[ii] = tile_mma_bf16_k64(sm.dor + 16 * tr * DS, DS, sm.v + 16 * tc * DS, DS, i16, g, acc);
                else accp[ii] = tile_mma<D / 4>(sm.dor + 16 * tr * DS, DS, 1, sm.v + 16 * tc * DS, DS, 1, acc);
            }
        }
        __syncthreads();   // all reads of V done -> V tile becomes the dV staging buffer
        {
#pragma unroll
            for (int tc = 0; tc < D / 16; ++tc)
#pragma unroll
                for (int j = 0; j < 4; ++j) sm.v[(16 * w + 4 * g + j) * DS + 16 * tc + i16] = accv[tc][j];
            const int tr = w & 1;
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
                const int tc = (w >> 1) + 2 * ii;
#pragma unroll
                for (int j = 0; j < 4; ++j) sm.ds[(16 * tr + 4 * g + j) * SS + 16 * tc + i16] = accp[ii][j];
            }
        }
        __syncthreads();

        // ---- softmax backward (twice), bias-gradient accumulation; dV -> global
        {
            const int r = t >> 3, c0 = (t & 7) * 8;
            float dp[8], p1[8], p2[8];
            float dot2 = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                dp[i] = sm.ds[r * SS + c0 + i]; p1[i] = sm.p1[r * SS + c0 + i]; p2[i] = sm.p2[r * SS + c0 + i];
                dot2 += dp[i] * p2[i];
            }
            dot2 = row8_sum(dot2);
            float da[8], dot1 = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) { da[i] = p2[i] * (dp[i] - dot2); dot1 += da[i] * p1[i]; }
            dot1 = row8_sum(dot1);
            if (HAS_BIAS && REG_ACC) {                    // dA over the P2 values this thread has just read: picked up by the owners below
#pragma unroll
                for (int i = 0; i < 8; ++i) sm.p2[r * SS + c0 + i] = da[i];
            }
            if (HAS_BIAS && !REG_ACC && r < NU) {
                float* arow = sm.acc + sm.top[r] * NT + c0;
#pragma unroll
                for (int i = 0; i < 8; ++i) arow[i] += da[i];
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) sm.ds[r * SS + c0 + i] = p1[i] * (da[i] - dot1) * scale;

            const int c4 = t % F;
#pragma unroll
            for (int p = 0; p < NT / RPP; ++p) {
                const int row = p * RPP + t / F;
                st4(dv + (tok0 + row) * ldg + h * D + c4 * 4, *reinterpret_cast<const float4*>(&sm.v[row * DS + c4 * 4]));
            }
        }
        __syncthreads();

        if (HAS_BIAS && REG_ACC) {                            // query row `row` was selected as r: its dA row goes to the table row
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int rk = sm.rank[w + 4 * i];            // wave-uniform
                if (rk < NU) accr[i] += sm.p2[rk * SS + lane];
            }
        }

        // ---- dQ[top] = dS K (32 x D, K = 64) ; dK = dS^T Q[top] (64 x D, K = 32)
        constexpr int NQ = D >= 32 ? D / 32 : 1;      // dQ tiles per wave (d = 16: two tiles in all, waves 0 and 1)
        f32x4 accq[NQ];
        f32x4 acck[D / 16];
        {
            const int tr = w & 1;
#pragma unroll
            for (int ii = 0; ii < NQ; ++ii) {
                const int tc = (w >> 1) + 2 * ii;
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                if (D == 16 && tc > 0) { accq[ii] = acc; continue; }
                // A(i=r,k=n) = dS[16tr + r][n] ; B(k=n, j=e) = K[n][16tc + e]
                accq[ii] = tile_mma<16>(sm.ds + 16 * tr * SS, SS, 1, sm.k + 16 * tc, 1, DS, acc);
            }
#pragma unroll
            for (int tc = 0; tc < D / 16; ++tc) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                // A(i=n,k=r) = dS[r][16w + n] ; B(k=r, j=e) = Qr[r][16tc + e]
                acck[tc] = tile_mma<8>(sm.ds + 16 * w, 1, SS, sm.qr + 16 * tc, 1, DS, acc);
            }
        }
        __syncthreads();   // all reads of K / P1 done
        {
            const int tr = w & 1;
            float* dqs = sm.p1;   // 32 x DS
#pragma unroll
            for (int ii = 0; ii < NQ; ++ii) {
                const int tc = (w >> 1) + 2 * ii;
                if (D == 16 && tc > 0) continue;
#pragma unroll
                for (int j = 0; j < 4; ++j) dqs[(16 * tr + 4 * g + j) * DS + 16 * tc + i16] = accq[ii][j];
            }
#pragma unroll
            for (int tc = 0; tc < D / 16; ++tc)
#pragma unroll
                for (int j = 0; j < 4; ++j) sm.k[(16 * w + 4 * g + j) * DS + 16 * tc + i16] = acck[tc][j];
        }
        __syncthreads();
        {
            const float* dqs = sm.p1;
            const int c4 = t % F;
#pragma unroll
            for (int p = 0; p < NT / RPP; ++p) {
                const int row = p * RPP + t / F;
                const size_t go = (tok0 + row) * ldg + h * D + c4 * 4;
                st4(dk + go, *reinterpret_cast<const float4*>(&sm.k[row * DS + c4 * 4]));
                const int r = sm.rank[row];
                float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
                if (r < NU) val = *reinterpret_cast<const float4*>(&dqs[r * DS + c4 * 4]);
                st4(dq + go, val);
            }
        }
    }

    if (HAS_BIAS && REG_ACC) {
        float* dst = dbias_part + (size_t)blockIdx.x * NT * NT;
#pragma unroll
        for (int i = 0; i < 16; ++i) dst[(w + 4 * i) * NT + lane] = accr[i];
    }
    if (HAS_BIAS && !REG_ACC) {
        __syncthreads();
        float4* dst = reinterpret_cast<float4*>(dbias_part + (size_t)blockIdx.x * NT * NT);
        for (int e = t; e < NT * NT / 4; e += 256) dst[e] = reinterpret_cast<const float4*>(sm.acc)[e];
    }
}

// ------------------------------------------------------------------------------------------------ small helpers
__global__ void bias_gather_kernel(const float* __restrict__ table, float* __restrict__ bias, int H) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;   // over H*64*64
    if (e >= H * NT * NT) return;
    const int h = e / (NT * NT), i = (e / NT) % NT, j = e % NT;
    const int rel = ((i >> 3) - (j >> 3) + 7) * 15 + ((i & 7) - (j & 7) + 7);
    bias[e] = table[rel * H + h];
}

// every block's bias gather of one model forward in ONE launch (18 launches of ~4 us otherwise): entry = blockIdx.y
constexpr int BIAS_MULTI_MAX = 32;
struct BiasMulti { const float* table[BIAS_MULTI_MAX]; float* bias[BIAS_MULTI_MAX]; int H[BIAS_MULTI_MAX]; };
__global__ void bias_gather_multi_kernel(const BiasMulti d) {
    const int H = d.H[blockIdx.y];
    const int e = blockIdx.x * blockDim.x + threadIdx.x;   // over H*64*64
    if (e >= H * NT * NT) return;
    const int h = e / (NT * NT), i = (e / NT) % NT, j = e % NT;
    const int rel = ((i >> 3) - (j >> 3) + 7) * 15 + ((i & 7) - (j & 7) + 7);
    d.bias[blockIdx.y][e] = d.table[blockIdx.y][rel * H + h];
}

// a workgroup's slice of the partial tiles p0, p0 + stride, ...: thread t sums float4 elements t, 256 + t, 512 + t, 768 + t of the tiles -
// the four loads of a tile (and of the next one: unrolled by two) in flight together; one load per iteration was a chain of dependent
// latencies, 96 us for the step's 18 tables - and folds its 16 (i, j) sums onto the 225 table rows in LDS.
__device__ __forceinline__ void table_grad_fold(const float* __restrict__ part, int parts, int p0, int stride, float* tab, int t) {
    float4 s[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) s[q] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 2
    for (int p = p0; p < parts; p += stride) {
        const float4* __restrict__ src = reinterpret_cast<const float4*>(part + (size_t)p * NT * NT);
        float4 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = src[q * 256 + t];
#pragma unroll
        for (int q = 0; q < 4; ++q) { s[q].x += v[q].x; s[q].y += v[q].y; s[q].z += v[q].z; s[q].w += v[q].w; }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int e = (q * 256 + t) * 4, i = e >> 6, j = e & 63;      // 4 consecutive j in the same row i
        const int rel = ((i >> 3) - (j >> 3) + 7) * 15 + ((i & 7) - (j & 7) + 7);
        atomicAdd(&tab[rel], s[q].x); atomicAdd(&tab[rel - 1], s[q].y); atomicAdd(&tab[rel - 2], s[q].z); atomicAdd(&tab[rel - 3], s[q].w);
    }
}

// one workgroup per head: sum the per-workgroup partials, then fold (i,j) pairs onto the 225 table rows
// grid = (16 element-chunks, H heads, Z part-slices): every thread sums one (i,j) element over its slice of
// the per-workgroup partials (coalesced across the 256 threads) and adds it onto the table row rel(i,j).
// grid = (Z part-slices, H heads): a workgroup sums its slice of the per-workgroup partials (coalesced float4
// reads), folds the 4096 (i,j) elements onto the 225 table rows with LDS atomics and issues 225 global atomics.
__global__ __launch_bounds__(256) void bias_table_grad_kernel(const float* __restrict__ part, int parts,
                                                              float* __restrict__ dtable, int H) {
    __shared__ float tab[225];
    const int h = blockIdx.y, t = threadIdx.x;
    if (t < 225) tab[t] = 0.f;
    __syncthreads();
    table_grad_fold(part, parts, h + H * blockIdx.x, H * gridDim.x, tab, t);
    __syncthreads();
    if (t < 225) atomicAdd(dtable + t * H + h, tab[t]);
}

// every block's table gradient of one backward pass in ONE launch (single-process runs: fused.py defers them to the end of backward)
constexpr int TGRAD_MULTI_MAX = 32;
// start[m] .. start[m + 1]: the workgroups of entry m (slices x heads): a FLAT grid - as (max slices, max heads, entries) with early exits the
// launch dispatched 36,864 workgroups for 2,300 with work and took 96 us
struct TableGradMulti { const float* part[TGRAD_MULTI_MAX]; float* dtable[TGRAD_MULTI_MAX]; int parts[TGRAD_MULTI_MAX]; int H[TGRAD_MULTI_MAX];
                        int start[TGRAD_MULTI_MAX + 1]; };
__global__ __launch_bounds__(256) void bias_table_grad_multi_kernel(const TableGradMulti d, int n) {
    __shared__ float tab[225];
    int m = 0;
    while (m + 1 < n && (int)blockIdx.x >= d.start[m + 1]) ++m;
    const int H = d.H[m], parts = d.parts[m], local = (int)blockIdx.x - d.start[m];
    const int h = local % H, t = threadIdx.x;
    const int z = min(max(parts / H / 4, 1), 128);            // the slices dhz_bias_table_grad would launch for this entry
    const int slice = local / H;
    const float* __restrict__ part = d.part[m];
    if (t < 225) tab[t] = 0.f;
    __syncthreads();
    table_grad_fold(part, parts, h + H * slice, H * z, tab, t);
    __syncthreads();
    if (t < 225) atomicAdd(d.dtable[m] + t * H + h, tab[t]);
}

__global__ void shift_mask_kernel(float* __restrict__ mask, int Hres, int Wres, int shift) {
    // mask[w][i][j]; one thread per element
    const int nWw = Wres >> 3;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = (Hres >> 3) * nWw * NT * NT;
    if (e >= total) return;
    const int wdx = e / (NT * NT), i = (e / NT) % NT, j = e % NT;
    const int wh = wdx / nWw, ww = wdx % nWw;
    auto label = [&](int tok) {
        const int hh = wh * 8 + (tok >> 3), wc = ww * 8 + (tok & 7);
        const int lh = hh < Hres - 8 ? 0 : (hh < Hres - shift ? 1 : 2);
        const int lw = wc < Wres - 8 ? 0 : (wc < Wres - shift ? 1 : 2);
        return lh * 3 + lw;
    };
    mask[e] = (label(i) != label(j)) ? -100.0f : 0.0f;
}

}  // namespace

// ------------------------------------------------------------------------------------------------ C ABI
// LDS above 64 KiB per workgroup needs an explicit opt-in; idempotent and cheap, so done per launch.
static void allow_smem(const void* fn, size_t bytes) {
    if (bytes > 48 * 1024) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

template <typename T>
static int ps_attn_fwd_t(const T* q, const T* k, const T* v, int ld, const uint8_t* idx,
                               const float* bias, const float* mask, T* out, int ldo, uint8_t* rank, int B_,
                               int H, int nW, int d, void* stream) {
    DHZ_REQUIRE(q && k && v && idx && out && rank, "dhz_ps_attn_fwd: null pointer");
    DHZ_REQUIRE(B_ > 0 && H > 0, "dhz_ps_attn_fwd: B_=%d H=%d", B_, H);
    DHZ_REQUIRE(d == 16 || d == 32 || d == 64, "dhz_ps_attn_fwd: head_dim %d unsupported (16, 32 or 64)", d);
    DHZ_REQUIRE(ld % 4 == 0 && ldo % 4 == 0 && ld >= H * d && ldo >= H * d, "dhz_ps_attn_fwd: bad ld %d/%d", ld, ldo);
    DHZ_REQUIRE(!mask || (nW > 0 && B_ % nW == 0), "dhz_ps_attn_fwd: B_=%d not a multiple of nW=%d", B_, nW);
    hipStream_t s = (hipStream_t)stream;
    // persistent workgroups, three per CU (LDS: 36.3 KiB at d = 32, 52.7 KiB at d = 64)
    const int resident = 256 * (d == 32 ? PSF_WG32 : PSF_WG64);
    const int grid = B_ * H < resident ? B_ * H : resident;
    if (d == 16) {                                 // embed_dim 16 ("Uformer16", utils/model_utils.py:96-98): four MFMA k-steps per score tile
        hipLaunchKernelGGL((ps_attn_fwd_kernel<16, T>), dim3(grid), dim3(256), sizeof(FwdSmem<16>), s, q, k, v, ld, idx,
                           bias, mask, out, ldo, rank, H, nW > 0 ? nW : 1, B_ * H);
    } else if (d == 32) {
        allow_smem(reinterpret_cast<const void*>(&ps_attn_fwd_kernel<32, T>), sizeof(FwdSmem<32>));
        hipLaunchKernelGGL((ps_attn_fwd_kernel<32, T>), dim3(grid), dim3(256), sizeof(FwdSmem<32>), s, q, k, v, ld, idx,
                           bias, mask, out, ldo, rank, H, nW > 0 ? nW : 1, B_ * H);
    } else {
        allow_smem(reinterpret_cast<const void*>(&ps_attn_fwd_kernel<64, T>), sizeof(FwdSmem<64>));
        hipLaunchKernelGGL((ps_attn_fwd_kernel<64, T>), dim3(grid), dim3(256), sizeof(FwdSmem<64>), s, q, k, v, ld, idx,
                           bias, mask, out, ldo, rank, H, nW > 0 ? nW : 1, B_ * H);
    }
    DHZ_CHECK_LAUNCH("dhz_ps_attn_fwd");
    return DHZ_OK;
}

extern "C" int dhz_ps_attn_fwd(const float* q, const float* k, const float* v, int ld, const uint8_t* idx,
                               const float* bias, const float* mask, float* out, int ldo, uint8_t* rank, int B_,
                               int H, int nW, int d, void* stream) {
    return ps_attn_fwd_t<float>(q, k, v, ld, idx, bias, mask, out, ldo, rank, B_, H, nW, d, stream);
}
extern "C" int dhz_ps_attn_fwd_dt(const void* q, const void* k, const void* v, int ld, const uint8_t* idx,
                                  const float* bias, const float* mask, void* out, int ldo, uint8_t* rank, int B_,
                                  int H, int nW, int d, int dtype, void* stream) {
    if (dtype == DHZ_F32) return ps_attn_fwd_t<float>((const float*)q, (const float*)k, (const float*)v, ld, idx, bias, mask, (float*)out, ldo, rank, B_, H, nW, d, stream);
    if (dtype == DHZ_BF16) return ps_attn_fwd_t<bf16s>((const bf16s*)q, (const bf16s*)k, (const bf16s*)v, ld, idx, bias, mask, (bf16s*)out, ldo, rank, B_, H, nW, d, stream);
    dhz_set_error("dhz_ps_attn_fwd_dt: unknown dtype %d", dtype);
    return DHZ_EINVAL;
}

// workgroups of the persistent backward kernel: two per CU.  (At head_dim 64 that is what the LDS allows - 78 KB each since the
// bias-gradient sum lives in registers, ONE before: config 4 46.0 -> 44.7 ms per step.  At head_dim 32 three would fit (53 KB), but
// 768 workgroups measured 38.48 ms per step against 38.16 with 512 and 38.95 with 256: more partial tables to reduce, no gain in the
// kernel.)
extern "C" int dhz_ps_attn_bwd_parts_d(int B_, int H, int d) {
    if (B_ <= 0 || H <= 0) return 0;
    (void)d;
    const int cap = 2 * dhz_num_cus();
    int per_head = cap / H;
    if (per_head < 1) per_head = 1;
    if (per_head > B_) per_head = B_;
    return per_head * H;
}
extern "C" int dhz_ps_attn_bwd_parts(int B_, int H) { return dhz_ps_attn_bwd_parts_d(B_, H, 64); }   // dhz_dense_attn_bwd's count (512 slots)

template <int D, bool HB, typename T>
static void launch_bwd(int parts, hipStream_t s, const T* q, const T* k, const T* v, int ld,
                       const float* bias, const float* mask, const uint8_t* rank, const T* dout, int ldo,
                       T* dq, T* dk, T* dv, int ldg, float* dbias_part, int B_, int H, int nW) {
    const size_t smem = sizeof(BwdSmem<D>);
    allow_smem(reinterpret_cast<const void*>(&ps_attn_bwd_kernel<D, HB, T>), smem);
    hipLaunchKernelGGL((ps_attn_bwd_kernel<D, HB, T>), dim3(parts), dim3(256), smem, s, q, k, v, ld, bias, mask, rank,
                       dout, ldo, dq, dk, dv, ldg, dbias_part, B_, H, nW);
}

template <typename T>
static int ps_attn_bwd_t(const T* q, const T* k, const T* v, int ld, const float* bias,
                         const float* mask, const uint8_t* rank, const T* dout, int ldo, T* dq,
                         T* dk, T* dv, int ldg, float* dbias_part, int B_, int H, int nW, int d,
                         void* stream) {
    DHZ_REQUIRE(q && k && v && rank && dout && dq && dk && dv, "dhz_ps_attn_bwd: null pointer");
    DHZ_REQUIRE(d == 16 || d == 32 || d == 64, "dhz_ps_attn_bwd: head_dim %d unsupported (16, 32 or 64)", d);
    DHZ_REQUIRE(!bias || dbias_part, "dhz_ps_attn_bwd: bias given but dbias_part is NULL");
    DHZ_REQUIRE(ld % 4 == 0 && ldo % 4 == 0 && ldg % 4 == 0, "dhz_ps_attn_bwd: leading dims must be multiples of 4");
    DHZ_REQUIRE(!mask || (nW > 0 && B_ % nW == 0), "dhz_ps_attn_bwd: B_=%d not a multiple of nW=%d", B_, nW);
    hipStream_t s = (hipStream_t)stream;
    const int parts = dhz_ps_attn_bwd_parts_d(B_, H, d);
    if (nW <= 0) nW = 1;
    if (d == 16) {
        if (bias) launch_bwd<16, true>(parts, s, q, k, v, ld, bias, mask, rank, dout, ldo, dq, dk, dv, ldg, dbias_part, B_, H, nW);
        else launch_bwd<16, false>(parts, s, q, k, v, ld, bias, mask, rank, dout, ldo, dq, dk, dv, ldg, dbias_part, B_, H, nW);
    } else if (d == 32) {
        if (bias) launch_bwd<32, true>(parts, s, q, k, v, ld, bias, mask, rank, dout, ldo, dq, dk, dv, ldg, dbias_part, B_, H, nW);
        else launch_bwd<32, false>(parts, s, q, k, v, ld, bias, mask, rank, dout, ldo, dq, dk, dv, ldg, dbias_part, B_, H, nW);
    } else {
        if (bias) launch_bwd<64, true>(parts, s, q, k, v, ld, bias, mask, rank, dout, ldo, dq, dk, dv, ldg, dbias_part, B_, H, nW);
        else launch_bwd<64, false>(parts, s, q, k, v, ld, bias, mask, rank, dout, ldo, dq, dk, dv, ldg, dbias_part, B_, H, nW);
    }
    DHZ_CHECK_LAUNCH("dhz_ps_attn_bwd");
    return DHZ_OK;
}

extern "C" int dhz_ps_attn_bwd(const float* q, const float* k, const float* v, int ld, const float* bias,
                               const float* mask, const uint8_t* rank, const float* dout, int ldo, float* dq,
                               float* dk, float* dv, int ldg, float* dbias_part, int B_, int H, int nW, int d,
                               void* stream) {
    return ps_attn_bwd_t<float>(q, k, v, ld, bias, mask, rank, dout, ldo, dq, dk, dv, ldg, dbias_part, B_, H, nW, d, stream);
}
extern "C" int dhz_ps_attn_bwd_dt(const void* q, const void* k, const void* v, int ld, const float* bias,
                                  const float* mask, const uint8_t* rank, const void* dout, int ldo, void* dq,
                                  void* dk, void* dv, int ldg, float* dbias_part, int B_, int H, int nW, int d,
                                  int dtype, void* stream) {
    if (dtype == DHZ_F32) return ps_attn_bwd_t<float>((const float*)q, (const float*)k, (const float*)v, ld, bias, mask, rank, (const float*)dout, ldo, (float*)dq, (float*)dk, (float*)dv, ldg, dbias_part, B_, H, nW, d, stream);
    if (dtype == DHZ_BF16) return ps_attn_bwd_t<bf16s>((const bf16s*)q, (const bf16s*)k, (const bf16s*)v, ld, bias, mask, rank, (const bf16s*)dout, ldo, (bf16s*)dq, (bf16s*)dk, (bf16s*)dv, ldg, dbias_part, B_, H, nW, d, stream);
    dhz_set_error("dhz_ps_attn_bwd_dt: unknown dtype %d", dtype);
    return DHZ_EINVAL;
}

extern "C" int dhz_bias_gather(const float* table, float* bias, int H, void* stream) {
    DHZ_REQUIRE(table && bias && H > 0, "dhz_bias_gather: bad arguments");
    const int n = H * NT * NT;
    hipLaunchKernelGGL(bias_gather_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, table, bias, H);
    DHZ_CHECK_LAUNCH("dhz_bias_gather");
    return DHZ_OK;
}

extern "C" int dhz_bias_gather_multi(const float* const* tables, float* const* biases, const int* heads, int n, void* stream) {
    DHZ_REQUIRE(tables && biases && heads && n > 0 && n <= BIAS_MULTI_MAX, "dhz_bias_gather_multi: null pointer or n=%d outside 1..%d", n,
                BIAS_MULTI_MAX);
    BiasMulti d = {};
    int hmax = 0;
    for (int i = 0; i < n; ++i) {
        DHZ_REQUIRE(tables[i] && biases[i] && heads[i] > 0, "dhz_bias_gather_multi: entry %d: null pointer or H=%d", i, heads[i]);
        d.table[i] = tables[i]; d.bias[i] = biases[i]; d.H[i] = heads[i];
        hmax = heads[i] > hmax ? heads[i] : hmax;
    }
    hipLaunchKernelGGL(bias_gather_multi_kernel, dim3((hmax * NT * NT + 255) / 256, n), dim3(256), 0, (hipStream_t)stream, d);
    DHZ_CHECK_LAUNCH("dhz_bias_gather_multi");
    return DHZ_OK;
}

extern "C" int dhz_bias_table_grad(const float* dbias_part, int parts, float* dtable, int H, int accumulate,
                                   void* stream) {
    DHZ_REQUIRE(dbias_part && dtable && H > 0 && parts > 0 && parts % H == 0, "dhz_bias_table_grad: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if (!accumulate) (void)hipMemsetAsync(dtable, 0, sizeof(float) * 225 * H, s);
    int z = parts / H / 4;            // >= 4 partials (64 KiB) per workgroup
    if (z < 1) z = 1;
    if (z > 128) z = 128;
    hipLaunchKernelGGL(bias_table_grad_kernel, dim3(z, H), dim3(256), 0, s, dbias_part, parts, dtable, H);
    DHZ_CHECK_LAUNCH("dhz_bias_table_grad");
    return DHZ_OK;
}

extern "C" int dhz_bias_table_grad_multi(const float* const* dbias_part, const int* parts, float* const* dtable, const int* heads, int n,
                                         void* stream) {
    DHZ_REQUIRE(dbias_part && parts && dtable && heads && n > 0 && n <= TGRAD_MULTI_MAX, "dhz_bias_table_grad_multi: null pointer or n=%d outside 1..%d",
                n, TGRAD_MULTI_MAX);
    TableGradMulti d = {};
    for (int i = 0; i < n; ++i) {
        DHZ_REQUIRE(dbias_part[i] && dtable[i] && heads[i] > 0 && parts[i] > 0 && parts[i] % heads[i] == 0,
                    "dhz_bias_table_grad_multi: entry %d: bad arguments", i);
        d.part[i] = dbias_part[i]; d.dtable[i] = dtable[i]; d.parts[i] = parts[i]; d.H[i] = heads[i];
        int z = parts[i] / heads[i] / 4;
        z = z < 1 ? 1 : (z > 128 ? 128 : z);
        d.start[i + 1] = d.start[i] + z * heads[i];
    }
    hipLaunchKernelGGL(bias_table_grad_multi_kernel, dim3(d.start[n]), dim3(256), 0, (hipStream_t)stream, d, n);
    DHZ_CHECK_LAUNCH("dhz_bias_table_grad_multi");
    return DHZ_OK;
}

extern "C" int dhz_shift_mask(float* mask, int Hres, int Wres, int shift, void* stream) {
    DHZ_REQUIRE(mask && Hres % 8 == 0 && Wres % 8 == 0 && Hres > 8 && Wres > 8 && shift > 0 && shift < 8,
                "dhz_shift_mask: bad arguments %dx%d shift %d", Hres, Wres, shift);
    const int n = (Hres / 8) * (Wres / 8) * NT * NT;
    hipLaunchKernelGGL(shift_mask_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, mask, Hres, Wres,
                       shift);
    DHZ_CHECK_LAUNCH("dhz_shift_mask");
    return DHZ_OK;
}
