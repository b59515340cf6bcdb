// F2 - the training feed of dataset.py:17-77 on the device: for each batch item (patch id, r, c, augmentation k) gather the
// ps x ps crop of the uint8 HWC patch pair held in HBM, apply one of the 8 rotate/flip maps of
// utils/dataset_utils.py:6-40 and emit float32 CHW in [0,1] (value / 255, the same IEEE division load_img performs).
// HBM-bound and tiny (2 x n x 3 x ps^2 floats out); one thread per output pixel, stores coalesced along the row.
#include <stdint.h>
#include "common.h"

namespace {

__global__ void crop_augment_pair_kernel(const uint8_t* __restrict__ gt, const uint8_t* __restrict__ hazy,
                                         const int* __restrict__ table, float* __restrict__ out_gt,
                                         float* __restrict__ out_hazy, int n, int Hs, int Ws, int ps) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t per = (size_t)ps * ps;
    if (e >= (size_t)n * per) return;
    const int item = (int)(e / per);
    const int pix = (int)(e % per);
    const int i = pix / ps, j = pix % ps, m = ps - 1;
    const int id = table[item * 4 + 0], r = table[item * 4 + 1], c = table[item * 4 + 2], k = table[item * 4 + 3] & 7;
    int si, sj;                                     // source position inside the crop for output (i, j)
    switch (k) {
        case 0: si = i;     sj = j;     break;      // identity
        case 1: si = m - j; sj = i;     break;      // rot90(k=1, dims=[-1,-2])
        case 2: si = m - i; sj = m - j; break;      // rot90(k=2)
        case 3: si = j;     sj = m - i; break;      // rot90(k=3)
        case 4: si = m - i; sj = j;     break;      // flip(-2)
        case 5: si = m - j; sj = m - i; break;      // rot90(k=1).flip(-2)
        case 6: si = i;     sj = m - j; break;      // rot90(k=2).flip(-2)
        default: si = j;    sj = i;     break;      // rot90(k=3).flip(-2)
    }
    const size_t src = (((size_t)id * Hs + r + si) * Ws + c + sj) * 3;
    const size_t dst = (size_t)item * 3 * per + pix;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        out_gt[dst + ch * per] = (float)gt[src + ch] / 255.f;
        out_hazy[dst + ch * per] = (float)hazy[src + ch] / 255.f;
    }
}

}  // namespace

extern "C" int dhz_crop_augment_pair(const uint8_t* gt, const uint8_t* hazy, const int* table, float* out_gt,
                                     float* out_hazy, int n, int Hs, int Ws, int ps, void* stream) {
    DHZ_REQUIRE(gt && hazy && table && out_gt && out_hazy, "dhz_crop_augment_pair: null pointer");
    DHZ_REQUIRE(n > 0 && ps > 0 && ps <= Hs && ps <= Ws, "dhz_crop_augment_pair: bad sizes n=%d ps=%d H=%d W=%d", n, ps, Hs, Ws);
    const size_t total = (size_t)n * ps * ps;
    hipLaunchKernelGGL(crop_augment_pair_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, gt,
                       hazy, table, out_gt, out_hazy, n, Hs, Ws, ps);
    DHZ_CHECK_LAUNCH("dhz_crop_augment_pair");
    return DHZ_OK;
}
