// Shared helpers for the gfx950 kernels (wave64, fp32-input MFMA).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "dehaze_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- activation storage types: fp32, or bf16 storage with fp32 arithmetic (BASELINE config 4).  Kernels are templated on the
//      storage type T of their token tensors and go through ld4 / st4 (4 consecutive elements, aligned to 4 elements).
struct bf16s { uint16_t v; };
__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {           // round to nearest even; hipcc emits v_cvt_pk_bf16_f32
    const __bf16 b = (__bf16)f;
    return __builtin_bit_cast(uint16_t, b);
}
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
#ifndef DHZ_ST4_NT
#define DHZ_ST4_NT 0          // diagnostics (tools/variants.sh): every st4 of a translation unit as a non-temporal store.  Measured on
#endif                        // csrc/elementwise.hip: the depthwise forward alone 161 -> 115 us, the training step 35.0 -> 35.2 ms (config 4:
                              // 35.4 -> 36.2): the next kernel reads what a plain store leaves in the memory-side cache.  Not used.
__device__ __forceinline__ void st4(float* p, float4 v) {
    if (DHZ_ST4_NT) __builtin_nontemporal_store(f32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4*>(p));
    else *reinterpret_cast<float4*>(p) = v;
}
__device__ __forceinline__ float4 ld4(const bf16s* p) {
    const uint2 r = *reinterpret_cast<const uint2*>(p);
    return make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16),
                       __uint_as_float(r.y & 0xffff0000u));
}
__device__ __forceinline__ void st4(bf16s* p, float4 v) {
    uint2 r;
    r.x = (uint32_t)f32_to_bf16(v.x) | ((uint32_t)f32_to_bf16(v.y) << 16);
    r.y = (uint32_t)f32_to_bf16(v.z) | ((uint32_t)f32_to_bf16(v.w) << 16);
    typedef uint32_t u32x2_ __attribute__((ext_vector_type(2)));
    if (DHZ_ST4_NT) __builtin_nontemporal_store(u32x2_{r.x, r.y}, reinterpret_cast<u32x2_*>(p));
    else *reinterpret_cast<uint2*>(p) = r;
}
__device__ __forceinline__ f32x4 ld4v(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4v(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ f32x4 ld4v(const bf16s* p) { const float4 r = ld4(p); return f32x4{r.x, r.y, r.z, r.w}; }
__device__ __forceinline__ void st4v(bf16s* p, f32x4 v) { st4(p, make_float4(v[0], v[1], v[2], v[3])); }
__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ float ld1(const bf16s* p) { return bf16_to_f32(p->v); }
__device__ __forceinline__ void st1(float* p, float v) { *p = v; }
__device__ __forceinline__ void st1(bf16s* p, float v) { p->v = f32_to_bf16(v); }

void dhz_set_error(const char* fmt, ...);
// Compute units of the CURRENT device (256 on MI355X; 256 when no device answers, e.g. argument checks on a GPU-less host).
// Persistent-grid sizes are "resident workgroups per CU x dhz_num_cus()", never a literal.
int dhz_num_cus();

#define DHZ_REQUIRE(cond, ...)            \
    do {                                  \
        if (!(cond)) {                    \
            dhz_set_error(__VA_ARGS__);   \
            return DHZ_EINVAL;            \
        }                                 \
    } while (0)

#define DHZ_CHECK_LAUNCH(name)                                                   \
    do {                                                                         \
        hipError_t e_ = hipGetLastError();                                       \
        if (e_ != hipSuccess) {                                                  \
            dhz_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
            return DHZ_ELAUNCH;                                                  \
        }                                                                        \
    } while (0)

// v_mfma_f32_16x16x4_f32: D[16x16] += A[16x4] * B[4x16].  lane l: a = A[l&15][l>>4], b = B[l>>4][l&15];
// acc[j] = D[4*(l>>4)+j][l&15].
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 acc) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
}

// One 16x16 output tile from LDS-resident operands with arbitrary strides:
//   A(i,k) = A[i*a_rs + k*a_ks]   (i = 0..15)        B(k,j) = B[k*b_ks + j*b_rs]   (j = 0..15)
// K = 4*KSTEPS.  The contraction index visited by lane group g at step s is k = 4s+g.
template <int KSTEPS>
__device__ __forceinline__ f32x4 tile_mma(const float* __restrict__ A, int a_rs, int a_ks,
                                          const float* __restrict__ B, int b_rs, int b_ks, f32x4 acc) {
    const int lane = threadIdx.x & 63;
    const int i = lane & 15, g = lane >> 4;
    const float* ap = A + i * a_rs + g * a_ks;
    const float* bp = B + i * b_rs + g * b_ks;
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) acc = mfma16(ap[4 * s * a_ks], bp[4 * s * b_ks], acc);
    return acc;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// GELU (exact-erf form, nn.GELU default) and its derivative from ONE exponential:
//   erf(|x|/sqrt2) = 1 - (a1 t + ... + a5 t^5) exp(-x^2/2),  t = 1/(1 + p|x|/sqrt2)   (Abramowitz-Stegun 7.1.26,
//   |error| <= 1.5e-7 - at the fp32 rounding level of the cdf), and the same exponential is the Gaussian pdf
//   needed by the derivative.  ~15 VALU instructions for both values instead of ~45 with erff()+expf().
__device__ __forceinline__ void gelu_both(float x, float& g, float& gp) {
    const float ax = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);       // v_rcp_f32 (1 ulp), no IEEE division sequence
    const float e = __expf(-ax * ax);                                    // = exp(-x^2/2)
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float erf_abs = 1.0f - poly * e;
    const float cdf = 0.5f * (1.0f + copysignf(erf_abs, x));
    g = x * cdf;
    gp = cdf + x * (0.39894228040143267794f * e);
}
// Four values at once, written on vectors so that the multiply / add chain compiles to packed fp32 instructions
// (v_pk_mul_f32 / v_pk_fma_f32: two values per issue slot; reciprocal, exponential and the sign transfer stay per value): ~15
// issue-slot equivalents per value instead of ~23 - the GELU evaluations are what bounds the depthwise stage in bf16 storage
// (config 4) and a fifth of the fused LeFF kernel.  Same formula as gelu_both; the two constants of the rational argument and
// of the exponent are pre-multiplied, so results agree with it to an fp32 rounding, not bit for bit.
__device__ __forceinline__ void gelu_both4(const f32x4 x, f32x4& g, f32x4& gp) {
    f32x4 ax, t, e, sg;
#pragma unroll
    for (int i = 0; i < 4; ++i) ax[i] = fabsf(x[i]);
    const f32x4 den = 1.0f + (0.3275911f * 0.70710678118654752440f) * ax;
    const f32x4 ex = (x * x) * (-0.5f * 1.44269504088896340736f);          // exp(-x^2/2) = 2^(x^2 * -0.5 log2 e)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        t[i] = __builtin_amdgcn_rcpf(den[i]);
        e[i] = __builtin_amdgcn_exp2f(ex[i]);
    }
    const f32x4 poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const f32x4 erf_abs = 1.0f - poly * e;
#pragma unroll
    for (int i = 0; i < 4; ++i) sg[i] = copysignf(erf_abs[i], x[i]);
    const f32x4 cdf = 0.5f + 0.5f * sg;
    g = x * cdf;
    gp = cdf + x * (0.39894228040143267794f * e);
}
__device__ __forceinline__ f32x4 gelu_f4(const f32x4 x) { f32x4 g, gp; gelu_both4(x, g, gp); return g; }
__device__ __forceinline__ float gelu_f(float x) { float g, gp; gelu_both(x, g, gp); return g; }
__device__ __forceinline__ float gelu_grad_f(float x) { float g, gp; gelu_both(x, g, gp); return gp; }

// ---- six-term products on the bf16 matrix pipe (csrc/split6_gemm.hip's arithmetic) for kernels that fuse a weight product: an fp32 value is
// cut by truncation into three bf16 pieces (hi + mid + lo == x exactly); a product is hh + (hm + mh) + (hl + lh + mm), small terms first
typedef __bf16 dhz_bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t dhz_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 dhz_mfma_bf16(dhz_u32x4 a, dhz_u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(dhz_bf16x8, a), __builtin_bit_cast(dhz_bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ uint32_t dhz_pack_top(float x1, float x0) {              // (x1 & 0xffff0000) | (x0 >> 16)
    return __builtin_amdgcn_perm(__float_as_uint(x1), __float_as_uint(x0), 0x07060302u);
}
// two fp32 values -> one packed pair per piece
__device__ __forceinline__ void dhz_split2x3(float x0, float x1, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
    const float r0 = x0 - __uint_as_float(__float_as_uint(x0) & 0xffff0000u), r1 = x1 - __uint_as_float(__float_as_uint(x1) & 0xffff0000u);
    const float q0 = r0 - __uint_as_float(__float_as_uint(r0) & 0xffff0000u), q1 = r1 - __uint_as_float(__float_as_uint(r1) & 0xffff0000u);
    hi = dhz_pack_top(x1, x0);
    mid = dhz_pack_top(r1, r0);
    lo = dhz_pack_top(q1, q0);
}
// eight fp32 values (a lane's 8 consecutive k) -> the three MFMA operands
__device__ __forceinline__ void dhz_split8x3(const f32x4 a, const f32x4 b, dhz_u32x4& hi, dhz_u32x4& mid, dhz_u32x4& lo) {
    uint32_t h[4], m[4], l[4];
    dhz_split2x3(a[0], a[1], h[0], m[0], l[0]);
    dhz_split2x3(a[2], a[3], h[1], m[1], l[1]);
    dhz_split2x3(b[0], b[1], h[2], m[2], l[2]);
    dhz_split2x3(b[2], b[3], h[3], m[3], l[3]);
    hi = dhz_u32x4{h[0], h[1], h[2], h[3]};
    mid = dhz_u32x4{m[0], m[1], m[2], m[3]};
    lo = dhz_u32x4{l[0], l[1], l[2], l[3]};
}
// 64-byte-row bf16 images ([row][32 k]): the four 16-byte chunks of a row XOR-ed with {0, 2, 3, 1}[(row >> 2) & 3] - a lane group's
// ds_read_b128 of rows i16, chunk g is conflict-free (same image as csrc/split6_gemm.hip)
__device__ __forceinline__ int dhz_off64(int row, int ch) { return row * 64 + 16 * (ch ^ ((0x78 >> (2 * ((row >> 2) & 3))) & 3)); }

// position of token (hh,ww) of an Hres x Wres map inside the (shifted) window layout:
// shifted map coords h' = (hh - shift) mod H  (torch.roll(x, -shift): shifted[h'] = x[(h'+shift)%H]),
// window id = (h'/8)*(W/8) + w'/8, token in window = (h'%8)*8 + w'%8.
__device__ __forceinline__ int window_slot(int hh, int ww, int Hres, int Wres, int shift) {
    int hs = hh - shift; if (hs < 0) hs += Hres;
    int ws = ww - shift; if (ws < 0) ws += Wres;
    return ((hs >> 3) * (Wres >> 3) + (ws >> 3)) * 64 + (hs & 7) * 8 + (ws & 7);
}
