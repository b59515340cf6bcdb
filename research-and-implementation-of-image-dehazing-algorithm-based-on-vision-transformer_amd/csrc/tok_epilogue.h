// K4 inside the token-Linear GEMMs: the epilogue of the out-projection / linear2 product IS the block's residual step
//     out[dst(m)] = res[dst(m)] + scale[img(m)] * (acc[m] + bias)                                  (M1:859-873, M1:873)
// with dst(m) = m (token order: the LeFF / Mlp branch, and the per-image factor of a backward-data product) or the
// window-reverse + un-roll position of window slot m (the attention branch: M1:859-868) - the GEMM result never exists in HBM in
// window order, the separate reverse_residual pass and its two extra trips over [T, C] are gone.
// Shared by csrc/split6_gemm.hip (both kernels) and csrc/linear_split.hip (gemm_split_kernel).
#pragma once
#include "common.h"

struct TokEpi {
    const float* res;      // shortcut, [images * HW] rows of ldc floats in TOKEN order, or null (nothing added)
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
