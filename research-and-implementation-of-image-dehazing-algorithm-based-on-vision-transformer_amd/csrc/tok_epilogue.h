// K4 inside the token-Linear GEMMs: the epilogue of the out-projection / linear2 product IS the block's residual step
//     out[dst(m)] = res[dst(m)] + scale[img(m)] * (acc[m] + bias)                                  (M1:859-873, M1:873)
// with dst(m) = m (token order: the LeFF / Mlp branch, and the per-image factor of a backward-data product) or the
// window-reverse + un-roll position of window slot m (the attention branch: M1:859-868) - the GEMM result never exists in HBM in
// window order, the separate reverse_residual pass and its two extra trips over [T, C] are gone.
// Shared by csrc/split6_gemm.hip (both kernels) and csrc/linear_split.hip (gemm_split_kernel).
#pragma once
#include "common.h"

struct TokEpi {
    const void* res;       // shortcut, [images * HW] rows of ldc elements (fp32, or bf16 in the bf16 kernels) in TOKEN order, or null
    const float* scale;    // per-image factor (DropPath), or null (1)
    int HW;                // tokens per image (a multiple of 64 whenever scale or win is set)
    int Hres, Wres, shift; // map geometry of the window layout (win != 0)
    int win;               // GEMM rows are window slots (dhz_ln_partition_fwd's order): store at the token-order position
};

// Rows mb + 16 a + i16 (a < WM) of one wave's accumulator block -> destination rows and the block's factor.  mb is a multiple of
// 16 * WM <= 64 that is WAVE-UNIFORM (tile origin + wave row block): the rows lie inside one 64-token window of one image, so both
// integer divisions are scalar.
template <int WM>
__device__ __forceinline__ void tok_epi_rows(const TokEpi& e, int mb, int i16, int (&dst)[WM], float& sc) {
    const int img = e.HW > 0 ? mb / e.HW : 0;
    sc = e.scale ? e.scale[img] : 1.f;
    if (!e.win) {
#pragma unroll
        for (int a = 0; a < WM; ++a) dst[a] = mb + 16 * a + i16;
        return;
    }
    const int r = mb - img * e.HW;
    const int nww = e.Wres >> 3;
    const int wi = r >> 6, wy = wi / nww, wx = wi - wy * nww;
#pragma unroll
    for (int a = 0; a < WM; ++a) {
        const int tok = (r & 63) + 16 * a + i16;
        int hh = 8 * wy + (tok >> 3) + e.shift, ww = 8 * wx + (tok & 7) + e.shift;      // torch.roll(+shift) of the reversed map
        if (hh >= e.Hres) hh -= e.Hres;
        if (ww >= e.Wres) ww -= e.Wres;
        dst[a] = img * e.HW + hh * e.Wres + ww;
    }
}

// bf16 kernels: a lane's 8 consecutive features of one row: out = bf16(res + sc * v)   (v = acc + bias in fp32)
__device__ __forceinline__ void tok_epi_store8_bf16(const TokEpi& e, uint16_t* C, size_t off, float sc, const f32x4 v0, const f32x4 v1) {
    typedef uint32_t u32x4_ __attribute__((ext_vector_type(4)));
    f32x4 r0 = sc * v0, r1 = sc * v1;
    if (e.res) {
        const u32x4_ q = *reinterpret_cast<const u32x4_*>(reinterpret_cast<const uint16_t*>(e.res) + off);
        r0 += f32x4{__uint_as_float(q[0] << 16), __uint_as_float(q[0] & 0xffff0000u), __uint_as_float(q[1] << 16), __uint_as_float(q[1] & 0xffff0000u)};
        r1 += f32x4{__uint_as_float(q[2] << 16), __uint_as_float(q[2] & 0xffff0000u), __uint_as_float(q[3] << 16), __uint_as_float(q[3] & 0xffff0000u)};
    }
    u32x4_ r;
    r[0] = (uint32_t)f32_to_bf16(r0[0]) | ((uint32_t)f32_to_bf16(r0[1]) << 16);
    r[1] = (uint32_t)f32_to_bf16(r0[2]) | ((uint32_t)f32_to_bf16(r0[3]) << 16);
    r[2] = (uint32_t)f32_to_bf16(r1[0]) | ((uint32_t)f32_to_bf16(r1[1]) << 16);
    r[3] = (uint32_t)f32_to_bf16(r1[2]) | ((uint32_t)f32_to_bf16(r1[3]) << 16);
    *reinterpret_cast<u32x4_*>(C + off) = r;
}

// host-side argument check shared by the entry points; returns nullptr or the complaint
inline const char* tok_epi_check(const TokEpi& e, int T) {
    if (e.win || e.scale) {
        if (e.HW <= 0 || e.HW % 64 || T % e.HW) return "tokens per image must be a positive multiple of 64 that divides T";
    }
    if (e.win) {
        if (e.Hres <= 0 || e.Wres <= 0 || e.Hres % 8 || e.Wres % 8 || e.Hres * e.Wres != e.HW) return "map must be Hres x Wres = HW with multiples of 8";
        if (e.shift < 0 || e.shift >= 8) return "shift must be in [0, 8)";
    }
    return nullptr;
}
