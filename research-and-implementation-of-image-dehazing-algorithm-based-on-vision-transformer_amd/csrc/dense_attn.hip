// Dense window attention (the My_model.Uformer twin, M0:428-492): per (window, head)
//     S = (Q * scale) K^T + bias[h] + mask[b % nW];   P = softmax(S);   O = P V
// Same LDS-resident structure as the ProbSparse kernel (one 256-thread workgroup per window-head, all three
// contractions on v_mfma_f32_16x16x4_f32); the backward recomputes P from Q,K (flash-style, nothing but the
// inputs is saved) and accumulates the bias gradient per workgroup in LDS.
#include "common.h"

namespace {

constexpr int NT = 64;
constexpr int SS = 68;

__device__ __forceinline__ float row4_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 1));
    return fmaxf(v, __shfl_xor(v, 2));
}
__device__ __forceinline__ float row4_sum(float v) {
    v += __shfl_xor(v, 1);
    return v + __shfl_xor(v, 2);
}

// P[r][c0..c0+15] = softmax_row(scale*S + bias + mask); 4 threads per row, in place in LDS
__device__ __forceinline__ void softmax_rows(float* S, const float* __restrict__ bias_h, const float* __restrict__ mask_w,
                                             float scale, int t) {
    const int r = t >> 2, c0 = (t & 3) * 16;
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = S[r * SS + c0 + i] * scale;
    if (bias_h) {
#pragma unroll
        for (int i = 0; i < 16; i += 4) {
            const float4 b = *reinterpret_cast<const float4*>(bias_h + r * NT + c0 + i);
            x[i] += b.x; x[i + 1] += b.y; x[i + 2] += b.z; x[i + 3] += b.w;
        }
    }
    if (mask_w) {
#pragma unroll
        for (int i = 0; i < 16; i += 4) {
            const float4 b = *reinterpret_cast<const float4*>(mask_w + r * NT + c0 + i);
            x[i] += b.x; x[i + 1] += b.y; x[i + 2] += b.z; x[i + 3] += b.w;
        }
    }
    float mx = x[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) mx = fmaxf(mx, x[i]);
    mx = row4_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { x[i] = expf(x[i] - mx); sum += x[i]; }
    sum = row4_sum(sum);
#pragma unroll
    for (int i = 0; i < 16; ++i) S[r * SS + c0 + i] = x[i] / sum;
}

template <int D>
struct DenseFwdSmem {
    static constexpr int DS = D + 4;
    float q[NT * DS];      // Q, later O
    float k[NT * DS];
    float v[NT * DS];
    float s[NT * SS];      // S -> P
};

template <int D>
__global__ __launch_bounds__(256) void dense_attn_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                             const float* __restrict__ v, int ld,
                                                             const float* __restrict__ bias,
                                                             const float* __restrict__ mask, float* __restrict__ out,
                                                             int ldo, int H, int nW, float scale) {
    constexpr int DS = D + 4, F = D / 4, RPP = 256 / F;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    DenseFwdSmem<D>& sm = *reinterpret_cast<DenseFwdSmem<D>*>(smem_raw);
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int i16 = lane & 15, g = lane >> 4;
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const size_t tok0 = (size_t)b * NT;
    {
        const int c4 = t % F;
#pragma unroll
        for (int p = 0; p < NT / RPP; ++p) {
            const int row = p * RPP + t / F;
            const size_t gi = (tok0 + row) * ld + h * D + c4 * 4;
            *reinterpret_cast<float4*>(&sm.q[row * DS + c4 * 4]) = *reinterpret_cast<const float4*>(q + gi);
            *reinterpret_cast<float4*>(&sm.k[row * DS + c4 * 4]) = *reinterpret_cast<const float4*>(k + gi);
            *reinterpret_cast<float4*>(&sm.v[row * DS + c4 * 4]) = *reinterpret_cast<const float4*>(v + gi);
        }
    }
    __syncthreads();
#pragma unroll
    for (int tc = 0; tc < 4; ++tc) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = tile_mma<D / 4>(sm.q + 16 * w * DS, DS, 1, sm.k + 16 * tc * DS, DS, 1, acc);
#pragma unroll
        for (int j = 0; j < 4; ++j) sm.s[(16 * w + 4 * g + j) * SS + 16 * tc + i16] = acc[j];
    }
    __syncthreads();
    softmax_rows(sm.s, bias ? bias + (size_t)h * NT * NT : nullptr,
                 mask ? mask + (size_t)(b % nW) * NT * NT : nullptr, scale, t);
    __syncthreads();
    float* O = sm.q;                                   // Q is dead
#pragma unroll
    for (int tc = 0; tc < D / 16; ++tc) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = tile_mma<16>(sm.s + 16 * w * SS, SS, 1, sm.v + 16 * tc, 1, DS, acc);
#pragma unroll
        for (int j = 0; j < 4; ++j) O[(16 * w + 4 * g + j) * DS + 16 * tc + i16] = acc[j];
    }
    __syncthreads();
    {
        const int c4 = t % F;
#pragma unroll
        for (int p = 0; p < NT / RPP; ++p) {
            const int row = p * RPP + t / F;
            *reinterpret_cast<float4*>(out + (tok0 + row) * ldo + h * D + c4 * 4) =
                *reinterpret_cast<const float4*>(&O[row * DS + c4 * 4]);
        }
    }
}

template <int D>
struct DenseBwdSmem {
    static constexpr int DS = D + 4;
    float q[NT * DS];      // later dQ staging
    float k[NT * DS];      // later dK staging
    float v[NT * DS];      // later dV staging
    float dO[NT * DS];
    float p[NT * SS];
    float ds[NT * SS];     // dP -> dS
    float acc[NT * NT];
};

template <int D, bool HAS_BIAS>
__global__ __launch_bounds__(256) void dense_attn_bwd_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v, int ld,
    const float* __restrict__ bias, const float* __restrict__ mask, const float* __restrict__ dout, int ldo,
    float* __restrict__ dq, float* __restrict__ dk, float* __restrict__ dv, int ldg, float* __restrict__ dbias_part,
    int B_, int H, int nW, float scale) {
    constexpr int DS = D + 4, F = D / 4, RPP = 256 / F;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    DenseBwdSmem<D>& sm = *reinterpret_cast<DenseBwdSmem<D>*>(smem_raw);
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int i16 = lane & 15, g = lane >> 4;
    const int h = blockIdx.x % H;
    const int bstep = gridDim.x / H;
    if (HAS_BIAS)
        for (int e = t; e < NT * NT / 4; e += 256) reinterpret_cast<float4*>(sm.acc)[e] = make_float4(0.f, 0.f, 0.f, 0.f);

    for (int b = blockIdx.x / H; b < B_; b += bstep) {
        const size_t tok0 = (size_t)b * NT;
        __syncthreads();
        {
            const int c4 = t % F;
#pragma unroll
            for (int p = 0; p < NT / RPP; ++p) {
                const int row = p * RPP + t / F;
                const size_t gi = (tok0 + row) * ld + h * D + c4 * 4;
                *reinterpret_cast<float4*>(&sm.q[row * DS + c4 * 4]) = *reinterpret_cast<const float4*>(q + gi);
                *reinterpret_cast<float4*>(&sm.k[row * DS + c4 * 4]) = *reinterpret_cast<const float4*>(k + gi);
                *reinterpret_cast<float4*>(&sm.v[row * DS + c4 * 4]) = *reinterpret_cast<const float4*>(v + gi);
                *reinterpret_cast<float4*>(&sm.dO[row * DS + c4 * 4]) =
                    *reinterpret_cast<const float4*>(dout + (tok0 + row) * ldo + h * D + c4 * 4);
            }
        }
        __syncthreads();
        // recompute S (row strip per wave) and dP = dO V^T
#pragma unroll
        for (int tc = 0; tc < 4; ++tc) {
            f32x4 a1 = {0.f, 0.f, 0.f, 0.f}, a2 = {0.f, 0.f, 0.f, 0.f};
            a1 = tile_mma<D / 4>(sm.q + 16 * w * DS, DS, 1, sm.k + 16 * tc * DS, DS, 1, a1);
            a2 = tile_mma<D / 4>(sm.dO + 16 * w * DS, DS, 1, sm.v + 16 * tc * DS, DS, 1, a2);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                sm.p[(16 * w + 4 * g + j) * SS + 16 * tc + i16] = a1[j];
                sm.ds[(16 * w + 4 * g + j) * SS + 16 * tc + i16] = a2[j];
            }
        }
        __syncthreads();
        softmax_rows(sm.p, bias ? bias + (size_t)h * NT * NT : nullptr,
                     mask ? mask + (size_t)(b % nW) * NT * NT : nullptr, scale, t);
        // same thread owns the same (row, 16 columns) of P and dP: dS = P * (dP - rowsum(dP*P))
        {
            const int r = t >> 2, c0 = (t & 3) * 16;
            float pr[16], dp[16], dot = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) { pr[i] = sm.p[r * SS + c0 + i]; dp[i] = sm.ds[r * SS + c0 + i]; dot += pr[i] * dp[i]; }
            dot = row4_sum(dot);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float dsv = pr[i] * (dp[i] - dot);
                if (HAS_BIAS) sm.acc[r * NT + c0 + i] += dsv;
                sm.ds[r * SS + c0 + i] = dsv * scale;
            }
        }
        __syncthreads();
        // dV = P^T dO (rows n = 16w..), dQ = dS K (rows 16w..), dK = dS^T Q (rows n = 16w..)
        f32x4 av[D / 16], aq[D / 16], ak[D / 16];
#pragma unroll
        for (int tc = 0; tc < D / 16; ++tc) {
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            av[tc] = tile_mma<16>(sm.p + 16 * w, 1, SS, sm.dO + 16 * tc, 1, DS, z);
            aq[tc] = tile_mma<16>(sm.ds + 16 * w * SS, SS, 1, sm.k + 16 * tc, 1, DS, z);
            ak[tc] = tile_mma<16>(sm.ds + 16 * w, 1, SS, sm.q + 16 * tc, 1, DS, z);
        }
        __syncthreads();
#pragma unroll
        for (int tc = 0; tc < D / 16; ++tc)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int o = (16 * w + 4 * g + j) * DS + 16 * tc + i16;
                sm.v[o] = av[tc][j]; sm.q[o] = aq[tc][j]; sm.k[o] = ak[tc][j];
            }
        __syncthreads();
        {
            const int c4 = t % F;
#pragma unroll
            for (int p = 0; p < NT / RPP; ++p) {
                const int row = p * RPP + t / F;
                const size_t go = (tok0 + row) * ldg + h * D + c4 * 4;
                *reinterpret_cast<float4*>(dq + go) = *reinterpret_cast<const float4*>(&sm.q[row * DS + c4 * 4]);
                *reinterpret_cast<float4*>(dk + go) = *reinterpret_cast<const float4*>(&sm.k[row * DS + c4 * 4]);
                *reinterpret_cast<float4*>(dv + go) = *reinterpret_cast<const float4*>(&sm.v[row * DS + c4 * 4]);
            }
        }
    }
    if (HAS_BIAS) {
        __syncthreads();
        float4* dst = reinterpret_cast<float4*>(dbias_part + (size_t)blockIdx.x * NT * NT);
        for (int e = t; e < NT * NT / 4; e += 256) dst[e] = reinterpret_cast<const float4*>(sm.acc)[e];
    }
}

template <typename Kern>
void allow_smem(Kern kern, size_t bytes) {
    if (bytes > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

}  // namespace

extern "C" int dhz_dense_attn_fwd(const float* q, const float* k, const float* v, int ld, const float* bias,
                                  const float* mask, float* out, int ldo, int B_, int H, int nW, int d, float scale,
                                  void* stream) {
    DHZ_REQUIRE(q && k && v && out, "dhz_dense_attn_fwd: null pointer");
    DHZ_REQUIRE(B_ > 0 && H > 0 && (d == 16 || d == 32 || d == 64), "dhz_dense_attn_fwd: bad B_=%d H=%d d=%d", B_, H, d);
    DHZ_REQUIRE(ld % 4 == 0 && ldo % 4 == 0, "dhz_dense_attn_fwd: leading dims must be multiples of 4");
    DHZ_REQUIRE(!mask || (nW > 0 && B_ % nW == 0), "dhz_dense_attn_fwd: B_=%d not a multiple of nW=%d", B_, nW);
    hipStream_t s = (hipStream_t)stream;
    if (nW <= 0) nW = 1;
    if (d == 16) {                                 // embed_dim 16 (utils/model_utils.py:96-98)
        hipLaunchKernelGGL(dense_attn_fwd_kernel<16>, dim3(B_ * H), dim3(256), sizeof(DenseFwdSmem<16>), s, q, k, v, ld,
                           bias, mask, out, ldo, H, nW, scale);
    } else if (d == 32) {
        allow_smem(&dense_attn_fwd_kernel<32>, sizeof(DenseFwdSmem<32>));
        hipLaunchKernelGGL(dense_attn_fwd_kernel<32>, dim3(B_ * H), dim3(256), sizeof(DenseFwdSmem<32>), s, q, k, v, ld,
                           bias, mask, out, ldo, H, nW, scale);
    } else {
        allow_smem(&dense_attn_fwd_kernel<64>, sizeof(DenseFwdSmem<64>));
        hipLaunchKernelGGL(dense_attn_fwd_kernel<64>, dim3(B_ * H), dim3(256), sizeof(DenseFwdSmem<64>), s, q, k, v, ld,
                           bias, mask, out, ldo, H, nW, scale);
    }
    DHZ_CHECK_LAUNCH("dhz_dense_attn_fwd");
    return DHZ_OK;
}

extern "C" int dhz_dense_attn_bwd(const float* q, const float* k, const float* v, int ld, const float* bias,
                                  const float* mask, const float* dout, int ldo, float* dq, float* dk, float* dv,
                                  int ldg, float* dbias_part, int B_, int H, int nW, int d, float scale, void* stream) {
    DHZ_REQUIRE(q && k && v && dout && dq && dk && dv, "dhz_dense_attn_bwd: null pointer");
    DHZ_REQUIRE(B_ > 0 && H > 0 && (d == 16 || d == 32 || d == 64), "dhz_dense_attn_bwd: bad B_=%d H=%d d=%d", B_, H, d);
    DHZ_REQUIRE(!bias || dbias_part, "dhz_dense_attn_bwd: bias given but dbias_part is NULL");
    DHZ_REQUIRE(!mask || (nW > 0 && B_ % nW == 0), "dhz_dense_attn_bwd: B_=%d not a multiple of nW=%d", B_, nW);
    hipStream_t s = (hipStream_t)stream;
    if (nW <= 0) nW = 1;
    const int parts = dhz_ps_attn_bwd_parts(B_, H);
#define LAUNCH(DD, HB)                                                                                              \
    do {                                                                                                            \
        allow_smem(&dense_attn_bwd_kernel<DD, HB>, sizeof(DenseBwdSmem<DD>));                                       \
        hipLaunchKernelGGL((dense_attn_bwd_kernel<DD, HB>), dim3(parts), dim3(256), sizeof(DenseBwdSmem<DD>), s, q, \
                           k, v, ld, bias, mask, dout, ldo, dq, dk, dv, ldg, dbias_part, B_, H, nW, scale);         \
    } while (0)
    if (d == 16) { if (bias) LAUNCH(16, true); else LAUNCH(16, false); }
    else if (d == 32) { if (bias) LAUNCH(32, true); else LAUNCH(32, false); }
    else { if (bias) LAUNCH(64, true); else LAUNCH(64, false); }
#undef LAUNCH
    DHZ_CHECK_LAUNCH("dhz_dense_attn_bwd");
    return DHZ_OK;
}
