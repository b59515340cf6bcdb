// BASELINE config 4 (bf16 activations): the 4x4 / stride-2 / pad-1 down-sampling convolution (Downsample, M1:606-622) as
// GEMMs on the bf16 matrix pipe.  The three products run on the token-Linear kernels of csrc/linear_bf16.hip
// (v_mfma_f32_16x16x32_bf16, fp32 accumulation) over an explicit tap-major patch matrix
//     col[b, ho, wo][(ky, kx, ci)] = x[b, 2 ho + ky - 1, 2 wo + kx - 1, ci]        (zero outside the image)
// which is cheap in this layout: a tap of an output pixel is ONE contiguous run of Cin bf16 of a token, so im2col / col2im are
// pure 16-byte-per-lane streaming copies (no transposition, no arithmetic besides the 4-term sum of col2im):
//     forward        y    = col . Wp^T + b        Wp[co][(ky, kx, ci)]                 dhz_linear_fwd_bf16
//     backward-data  dcol = dy . Wp  ->  dx[b, h, w] = sum over the 4 (output pixel, tap) pairs that read it   dhz_linear_dgrad_bf16
//     weight grad    dWp += dy^T . col, db += column sums of dy                        dhz_linear_wgrad_bf16
// At bf16 matrix rates these GEMMs are HBM-bound anyway (K = 16 Cin >= 1024); the patch matrix costs 4x the input in bf16 =
// 2x the input in fp32 terms, written once and read by two of the GEMMs.
#include "common.h"

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// one thread per 16-byte chunk (8 channels) of the patch matrix
__global__ __launch_bounds__(256) void im2col_k4s2_bf16_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ col, int B,
                                                               int H, int W, int Cin, long long nchunk) {
    const int c8n = Cin >> 3;
    const int Ho = H >> 1, Wo = W >> 1;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < nchunk; e += (long long)gridDim.x * 256) {
        const int c8 = (int)(e % c8n);
        long long r = e / c8n;
        const int tap = (int)(r & 15);
        r >>= 4;
        const int wo = (int)(r % Wo);
        r /= Wo;
        const int ho = (int)(r % Ho), b = (int)(r / Ho);
        const int hi = 2 * ho + (tap >> 2) - 1, wi = 2 * wo + (tap & 3) - 1;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (hi >= 0 && hi < H && wi >= 0 && wi < W)
            v = *reinterpret_cast<const u32x4*>(x + (((size_t)b * H + hi) * W + wi) * Cin + 8 * c8);
        *reinterpret_cast<u32x4*>(col + e * 8) = v;
    }
}

__device__ __forceinline__ void acc8(float (&s)[8], u32x4 v) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        s[2 * i] += __uint_as_float(v[i] << 16);
        s[2 * i + 1] += __uint_as_float(v[i] & 0xffff0000u);
    }
}

// one thread per 16-byte chunk of dx: input pixel (h, w) is read by output rows ho = (h + 1 - ky) / 2 for the two ky of the
// parity of h + 1 (and likewise along w): 4 (output pixel, tap) pairs, summed in fp32, rounded once
__global__ __launch_bounds__(256) void col2im_k4s2_bf16_kernel(const uint16_t* __restrict__ dcol, uint16_t* __restrict__ dx, int B,
                                                               int H, int W, int Cin, long long nchunk) {
    const int c8n = Cin >> 3;
    const int Ho = H >> 1, Wo = W >> 1;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < nchunk; e += (long long)gridDim.x * 256) {
        const int c8 = (int)(e % c8n);
        long long r = e / c8n;
        const int w = (int)(r % W);
        r /= W;
        const int h = (int)(r % H), b = (int)(r / H);
        float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int ky = ((h + 1) & 1) + 2 * a, ho = (h + 1 - ky) >> 1;
            if (h + 1 - ky < 0 || ho >= Ho) continue;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int kx = ((w + 1) & 1) + 2 * c, wo = (w + 1 - kx) >> 1;
                if (w + 1 - kx < 0 || wo >= Wo) continue;
                const size_t row = ((size_t)b * Ho + ho) * Wo + wo;
                acc8(s, *reinterpret_cast<const u32x4*>(dcol + (row * 16 + ky * 4 + kx) * Cin + 8 * c8));
            }
        }
        u32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = (uint32_t)f32_to_bf16(s[2 * i]) | ((uint32_t)f32_to_bf16(s[2 * i + 1]) << 16);
        *reinterpret_cast<u32x4*>(dx + e * 8) = o;
    }
}

int check(const char* who, const void* a, const void* b, int B, int H, int W, int Cin) {
    DHZ_REQUIRE(a && b, "%s: null pointer", who);
    DHZ_REQUIRE(B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && Cin > 0 && Cin % 8 == 0,
                "%s: bad shape B=%d %dx%d Cin=%d (even map, Cin a multiple of 8)", who, B, H, W, Cin);
    DHZ_REQUIRE((((uintptr_t)a | (uintptr_t)b) & 15) == 0, "%s: buffers must be 16-byte aligned", who);
    return DHZ_OK;
}

}  // namespace

extern "C" int dhz_im2col_k4s2_bf16(const void* x, void* col, int B, int H, int W, int Cin, void* stream) {
    if (int rc = check("dhz_im2col_k4s2_bf16", x, col, B, H, W, Cin)) return rc;
    const long long nchunk = (long long)B * (H / 2) * (W / 2) * 16 * (Cin / 8);
    const long long blocks = (nchunk + 255) / 256;
    const int grid = (int)(blocks < 16LL * dhz_num_cus() ? blocks : 16LL * dhz_num_cus());
    hipLaunchKernelGGL(im2col_k4s2_bf16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x, (uint16_t*)col, B, H,
                       W, Cin, nchunk);
    DHZ_CHECK_LAUNCH("dhz_im2col_k4s2_bf16");
    return DHZ_OK;
}

extern "C" int dhz_col2im_k4s2_bf16(const void* dcol, void* dx, int B, int H, int W, int Cin, void* stream) {
    if (int rc = check("dhz_col2im_k4s2_bf16", dcol, dx, B, H, W, Cin)) return rc;
    const long long nchunk = (long long)B * H * W * (Cin / 8);
    const long long blocks = (nchunk + 255) / 256;
    const int grid = (int)(blocks < 16LL * dhz_num_cus() ? blocks : 16LL * dhz_num_cus());
    hipLaunchKernelGGL(col2im_k4s2_bf16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)dcol, (uint16_t*)dx, B,
                       H, W, Cin, nchunk);
    DHZ_CHECK_LAUNCH("dhz_col2im_k4s2_bf16");
    return DHZ_OK;
}
