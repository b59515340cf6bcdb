// K9b - the output projection of the U-shaped net (My_model_1.py:696-723): Conv2d(C -> 3, 3x3, pad 1) from the token layout
// [B, H*W, C] to an NCHW image [B, 3, H, W], its backward-data and its weight/bias gradient.
//
// The library serves this "thin" convolution (3 output channels) with implicit-GEMM kernels built for wide outputs:
// 235 us forward and 340 us backward at B = 32, 128 x 128, C = 64 - 0.6 TB/s on a problem whose whole traffic is the
// 134 MB token tensor.  Round 1-2 ran it as VALU kernels (lane <-> pixel, weights as LDS broadcasts): 128 / 150 us, bound by
// the four LDS reads per 12 FMAs.  Round 3: the matrix pipe after all, by splitting the 3 x 3 window off the contraction:
//   forward : Z[q][(o, tap)] = sum_c x[q][c] w[o][c][tap] is a plain GEMM over the 10 x 18 HALO positions q of an 8 x 16
//             output tile (K = C contiguous in the token row, N = 27 -> 32: two 16-column blocks of v_mfma_f32_16x16x4_f32),
//             and y[o][p] = sum_tap Z[p + off(tap)][(o, tap)] is a 9-term gather of that 180 x 32 matrix from LDS.  1.4 x
//             the FLOPs of the direct form (halo positions), all of them on the matrix pipe.
//   wgrad   : dW^T[c][(o, tap)] = sum_q x[q][c] G[q][(o, tap)] with G[q][(o, tap)] = dy[o][q - off(tap)] (zero outside the
//             tile's own output pixels) - a GEMM with the halo positions as the contraction (K = 180 = 45 MFMA steps); G is
//             built in LDS from the 3 x 128 dy values of the tile.  Persistent workgroups, accumulators in registers over all
//             tiles, one atomic per (o, c, tap) per workgroup at the end.
//   dgrad   : (VALU) lane <-> channel (the token row is the contiguous axis of the output), the 27 weights of the lane's channel
//             live in registers, the 3-channel gradient tile is read as LDS broadcasts.
// Channels are processed in chunks of 64 (C = 128: two passes over the tile) so two workgroups fit a CU's LDS.
#include "common.h"

#ifndef TW_ABL
#define TW_ABL 0        // timing diagnostics of the weight-gradient kernel (tools/variants.sh): 1 no MFMA loop, 2 no G build, 4 no token fetch / put, 8 no atomics
#endif

namespace {

constexpr int TW = 16, TH = 8;                 // output pixels per workgroup
constexpr int HW_ = TW + 2, HH_ = TH + 2;      // halo tile
constexpr int NPOS = HH_ * HW_;                // 180

constexpr int CC = 64;                         // channel chunk
constexpr int XS = CC + 4;                     // row stride of the token halo tile (floats): b128 row reads conflict-free
constexpr int ZS = 36;                         // row stride of Z (forward)
constexpr int GS = 33;                         // row stride of G (wgrad)
constexpr int NRB = (NPOS + 15) / 16;          // 12 row blocks of 16 halo positions (the last one runs 12 rows past the tile)

struct ThinFwdSmem {
    float x[NPOS * XS];                        // token halo tile of one channel chunk (rows 180 .. 191 of the last block: whatever
    float z[NRB * 16 * ZS];                    // follows - their Z rows are never read);  Z, and before the MFMAs the weights
};
struct ThinWgradSmem {
    float x[NPOS * XS];
    float g[NPOS * GS];
    float dy[3 * TH * TW];
    float red[4];
};

// stage channels [cb, cb + 64) of the 10 x 18 halo tile (zero outside the image) of a token tensor [B, H*W, C] or (BLOCKED) of a
// channel-blocked NCHW8c map [B, C/8, H, W, 8] (the layout of the fp32 VGG feature engine)
// All requests of a thread are issued before the first LDS write (clamped addresses + select instead of a divergent guard): as a
// loop of (load, write) pairs the tile cost one global round trip per iteration - 12 of them.
constexpr int TOK_NIT = (NPOS * (CC / 4) + 255) / 256;           // 16-byte requests per thread per (tile, channel chunk): 12
struct TokRegs {
    f32x4 v[TOK_NIT];
    unsigned ok;                                                 // bit i: request i lies inside the image
};
template <int C, bool BLOCKED = false, typename T = float>
__device__ __forceinline__ void fetch_tokens(TokRegs& r, const T* __restrict__ x, int bimg, int ty, int tx, int H, int W, int cb) {
    constexpr int C4 = CC / 4;
    const int t = threadIdx.x;
    const size_t ib = (size_t)bimg * H * W;
    r.ok = 0u;
#pragma unroll
    for (int i = 0; i < TOK_NIT; ++i) {
        const int e = min(t + 256 * i, NPOS * C4 - 1);
        const int pos = BLOCKED ? e % NPOS : e / C4, c4 = BLOCKED ? e / NPOS : e % C4;     // consecutive lanes walk the contiguous axis
        const int yy = ty * TH - 1 + pos / HW_, xx = tx * TW - 1 + pos % HW_;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) r.ok |= 1u << i;
        const int yc = min(max(yy, 0), H - 1), xc = min(max(xx, 0), W - 1);
        const int c = cb + c4 * 4;
        const size_t o = BLOCKED ? (((size_t)bimg * (C / 8) + c / 8) * H * W + (size_t)yc * W + xc) * 8 + (c & 7)
                                 : (ib + (size_t)yc * W + xc) * C + c;
        r.v[i] = ld4v(x + o);
    }
}
template <bool BLOCKED = false>
__device__ __forceinline__ void put_tokens(float* xs, const TokRegs& r) {
    constexpr int C4 = CC / 4;
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < TOK_NIT; ++i) {
        const int e = t + 256 * i;
        if (e < NPOS * C4) {
            const int pos = BLOCKED ? e % NPOS : e / C4, c4 = BLOCKED ? e / NPOS : e % C4;
            *reinterpret_cast<f32x4*>(&xs[pos * XS + c4 * 4]) = ((r.ok >> i) & 1u) ? r.v[i] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
}
// All requests of a thread are issued before the first LDS write (clamped addresses + select instead of a divergent guard): as a
// loop of (load, write) pairs the tile cost one global round trip per iteration - 12 of them.
template <int C, bool BLOCKED = false, typename T = float>
__device__ __forceinline__ void stage_tokens(float* xs, const T* __restrict__ x, int bimg, int ty, int tx, int H, int W, int cb) {
    TokRegs r;
    fetch_tokens<C, BLOCKED, T>(r, x, bimg, ty, tx, H, W, cb);
    put_tokens<BLOCKED>(xs, r);
}

// TRANSPOSED: w is a [C, 3, 3, 3] tensor (a 3 -> C convolution's weight) and the kernel computes that layer's
// backward-data: y[b, o, p] = sum_{c, ky, kx} w[c][o][2 - ky][2 - kx] x[b, c, p + (ky - 1, kx - 1)]
template <int C, bool BLOCKED, bool TRANSPOSED, typename T = float>
__global__ __launch_bounds__(256, 2) void thin_conv_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                               const float* __restrict__ bias, float* __restrict__ y, int H,
                                                               int W, int tiles_x, int tiles_y, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    ThinFwdSmem& sm = *reinterpret_cast<ThinFwdSmem*>(smem_raw);
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int i16 = lane & 15, g = lane >> 4;
    constexpr int NCH = C / CC;
    // Persistent workgroups.  The weights go through LDS once (coalesced reads of w, re-ordered to wl[c][n = o * 9 + tap], 32
    // columns, 27 used; the slot is Z's) into registers in the MFMA operand layout and stay there for all tiles.
    float* wl = sm.z;
    static_assert(C * 32 <= NRB * 16 * ZS, "weight staging must fit the Z slot");
    {
        constexpr int NW = (27 * C + 255) / 256;
        float wv_[NW];
#pragma unroll
        for (int i = 0; i < NW; ++i) wv_[i] = w[min(t + 256 * i, 27 * C - 1)];
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int e = t + 256 * i;
            if (e < 27 * C) {
                int o, c, tap;
                if (TRANSPOSED) { c = e / 27; o = (e / 9) % 3; tap = 8 - e % 9; }      // w[c][o][ky][kx] acts at tap (2-ky, 2-kx)
                else { o = e / (9 * C); c = (e / 9) % C; tap = e % 9; }
                wl[c * 32 + o * 9 + tap] = wv_[i];
            }
        }
    }
    for (int e = t; e < 5 * C; e += 256) wl[(e / 5) * 32 + 27 + e % 5] = 0.f;
    __syncthreads();
    // contraction order inside a 16-channel group: MFMA step j of group s4 takes channel 16 s4 + 4 g + j from lane group g, so a
    // lane's four steps read ONE b128 of its token row (any order of the contraction is as good as another)
    float wr[C / 16][4][2];
#pragma unroll
    for (int s4 = 0; s4 < C / 16; ++s4)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) wr[s4][j][nb] = wl[(16 * s4 + 4 * g + j) * 32 + 16 * nb + i16];
    const float b0 = bias ? bias[0] : 0.f, b1 = bias ? bias[1] : 0.f, b2 = bias ? bias[2] : 0.f;
    __syncthreads();                                                          // the slot becomes Z

    // Z^T block = W^T (rows n) . X^T (columns q): the accumulator of lane (q = i16) holds n = 4 g .. 4 g + 3 -> one b128 store
    // into Z[q][.].  Wave wv owns row blocks 3 wv .. 3 wv + 2 (halo positions) x both column blocks.
    // The token requests of step s + 1 (next channel chunk, or the next tile's first) are in flight while step s multiplies,
    // gathers and stores - before that, every tile paid one exposed global round trip with two workgroups per CU to hide it.
    TokRegs regs;
    int tile = blockIdx.x;
    auto coords = [&](int tl, int& tx, int& ty, int& bimg) { tx = tl % tiles_x; ty = (tl / tiles_x) % tiles_y; bimg = tl / (tiles_x * tiles_y); };
    {
        int tx, ty, bimg;
        coords(tile, tx, ty, bimg);
        fetch_tokens<C, BLOCKED, T>(regs, x, bimg, ty, tx, H, W, 0);
    }
    for (; tile < ntiles; tile += gridDim.x) {
        int tx, ty, bimg;
        coords(tile, tx, ty, bimg);
        f32x4 acc[3][2];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) acc[r][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            put_tokens<BLOCKED>(sm.x, regs);
            __syncthreads();
            if (ch + 1 < NCH) fetch_tokens<C, BLOCKED, T>(regs, x, bimg, ty, tx, H, W, (ch + 1) * CC);
            else if (tile + (int)gridDim.x < ntiles) {
                int nx, ny, nb_;
                coords(tile + gridDim.x, nx, ny, nb_);
                fetch_tokens<C, BLOCKED, T>(regs, x, nb_, ny, nx, H, W, 0);
            }
#pragma unroll
            for (int s4 = 0; s4 < CC / 16; ++s4) {
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const f32x4 xv = *reinterpret_cast<const f32x4*>(&sm.x[(16 * (3 * wv + r) + i16) * XS + 16 * s4 + 4 * g]);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int nb = 0; nb < 2; ++nb)
                            acc[r][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[ch * (CC / 16) + s4][j][nb], xv[j], acc[r][nb], 0, 0, 0);
                }
            }
            if (ch + 1 < NCH) __syncthreads();                                // this chunk's tile is read; the next one may land
        }
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
                *reinterpret_cast<f32x4*>(&sm.z[(16 * (3 * wv + r) + i16) * ZS + 16 * nb + 4 * g]) = acc[r][nb];
        __syncthreads();                                                      // Z complete; also: every wave is done with sm.x
        for (int e = t; e < 3 * TH * TW; e += 256) {
            const int o = e / (TH * TW), p = e % (TH * TW);
            const int py = p / TW, px = p % TW;
            const int yy = ty * TH + py, xx = tx * TW + px;
            float a = o == 0 ? b0 : (o == 1 ? b1 : b2);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) a += sm.z[((py + tap / 3) * HW_ + px + tap % 3) * ZS + o * 9 + tap];
            if (yy < H && xx < W) y[(((size_t)bimg * 3 + o) * H + yy) * W + xx] = a;
        }
        // (the next tile's Z stores come after its own put + barrier: the gather above is finished by then)
    }
}

// dy halo tile [NPOS][4] (3 used), zero outside the image
__device__ __forceinline__ void stage_dy(float* ds, const float* __restrict__ dy, int bimg, int ty, int tx, int H, int W) {
    for (int e = threadIdx.x; e < NPOS * 3; e += 256) {
        const int o = e / NPOS, pos = e % NPOS;
        const int yy = ty * TH - 1 + pos / HW_, xx = tx * TW - 1 + pos % HW_;
        float v = 0.f;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) v = dy[(((size_t)bimg * 3 + o) * H + yy) * W + xx];
        ds[pos * 4 + o] = v;
    }
}

// dx[b, p, c] = sum_{o, ky, kx} w[o][c][ky][kx] * dy[b, o, p + (1 - ky, 1 - kx)]
template <int C, typename T = float>
__global__ __launch_bounds__(256) void thin_conv_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                              T* __restrict__ dx, int H, int W, int tiles_x, int tiles_y) {
    __shared__ __attribute__((aligned(16))) float ds[NPOS * 4];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int tx = blockIdx.x % tiles_x, ty = (blockIdx.x / tiles_x) % tiles_y, bimg = blockIdx.x / (tiles_x * tiles_y);
    stage_dy(ds, dy, bimg, ty, tx, H, W);
    __syncthreads();
    const size_t ib = (size_t)bimg * H * W;
#pragma unroll 1
    for (int cb = 0; cb < C; cb += 64) {
        const int c = cb + lane;
        float wr[3][9];
#pragma unroll
        for (int o = 0; o < 3; ++o)
#pragma unroll
            for (int k = 0; k < 9; ++k) wr[o][k] = w[(o * C + c) * 9 + k];
#pragma unroll 2
        for (int i = 0; i < TH * TW / 4; ++i) {
            const int pix = wv * (TH * TW / 4) + i;
            const int py = pix / TW, px = pix % TW;
            const int yy = ty * TH + py, xx = tx * TW + px;
            float acc = 0.f;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const f32x4 g = *reinterpret_cast<const f32x4*>(&ds[((py + 2 - ky) * HW_ + px + 2 - kx) * 4]);   // broadcast
                    acc += wr[0][ky * 3 + kx] * g[0] + wr[1][ky * 3 + kx] * g[1] + wr[2][ky * 3 + kx] * g[2];
                }
            if (yy < H && xx < W) st1(dx + (ib + (size_t)yy * W + xx) * C + c, acc);
        }
    }
}

// dw[o][c][ky][kx] += sum_p dy[o][p] x[p + (ky - 1, kx - 1)][c] ; db[o] += sum_p dy[o][p]
template <int C, typename T = float>
__global__ __launch_bounds__(256, 2) void thin_conv_wgrad_kernel(const float* __restrict__ dy, const T* __restrict__ x,
                                                                 float* __restrict__ dw, float* __restrict__ db, int B, int H,
                                                                 int W, int tiles_x, int tiles_y) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    ThinWgradSmem& sm = *reinterpret_cast<ThinWgradSmem*>(smem_raw);
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int i16 = lane & 15, g = lane >> 4;
    constexpr int NCH = C / CC;
    // dW^T block [16 channels 16 (4 ch + wv) ..][16 columns of (o, tap)]: A = x^T (row c = i16, step q = g), B = G (column n = i16)
    f32x4 acc[NCH][2];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) acc[ch][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    float accb0 = 0.f, accb1 = 0.f;                                           // db: dy sums of output t / 128, and (t < 128) of output 2
    for (int e = t; e < NPOS * GS; e += 256) sm.g[e] = 0.f;                   // columns 27 .. 32 stay zero
    const int ntiles = B * tiles_x * tiles_y;
    // the token requests (and the 3 x 128 dy values) of the next step - next channel chunk, or the next tile - are in flight while
    // this one multiplies
    TokRegs regs;
    float dyr[2];
    auto coords = [&](int tl, int& tx, int& ty, int& bimg) { tx = tl % tiles_x; ty = (tl / tiles_x) % tiles_y; bimg = tl / (tiles_x * tiles_y); };
    auto fetch_dy = [&](int tl) {
        int tx, ty, bimg;
        coords(tl, tx, ty, bimg);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e = min(t + 256 * i, 3 * TH * TW - 1);
            const int o = e / (TH * TW), p = e % (TH * TW);
            const int yy = ty * TH + p / TW, xx = tx * TW + p % TW;
            const float v = dy[(((size_t)bimg * 3 + o) * H + min(yy, H - 1)) * W + min(xx, W - 1)];
            dyr[i] = (yy < H && xx < W) ? v : 0.f;
        }
    };
    if ((int)blockIdx.x < ntiles) {
        int tx, ty, bimg;
        coords(blockIdx.x, tx, ty, bimg);
        fetch_tokens<C, false, T>(regs, x, bimg, ty, tx, H, W, 0);
        fetch_dy(blockIdx.x);
    }
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int tx, ty, bimg;
        coords(tile, tx, ty, bimg);
        const bool has_next = tile + (int)gridDim.x < ntiles;
        __syncthreads();                                                      // previous tile's reads done
        sm.dy[t] = dyr[0];
        accb0 += dyr[0];
        if (t < 3 * TH * TW - 256) { sm.dy[t + 256] = dyr[1]; accb1 += dyr[1]; }
        if (!(TW_ABL & 4)) put_tokens<false>(sm.x, regs);
        __syncthreads();
        if (TW_ABL & 4) {} else if (NCH > 1) fetch_tokens<C, false, T>(regs, x, bimg, ty, tx, H, W, CC);
        else if (has_next) {
            int nx, ny, nb_;
            coords(tile + gridDim.x, nx, ny, nb_);
            fetch_tokens<C, false, T>(regs, x, nb_, ny, nx, H, W, 0);
            fetch_dy(tile + gridDim.x);
        }
        // G[q][(o, tap)] = dy[o][q - (ky, kx)] in tile coordinates (q: halo position, its origin one pixel up / left)
        // (no loop with a back edge between a fetch and its put: hipcc drains vmcnt in front of one, which would wait for the
        // prefetch right here)
#pragma unroll
        for (int it = 0; it < (NPOS * 3 + 255) / 256; ++it) {                 // one (halo position, output) per trip: 9 taps
            const int e = t + 256 * it;
            if (e < NPOS * 3 && !(TW_ABL & 2)) {
                const int q = e % NPOS, o = e / NPOS;
                const int qy = q / HW_, qx = q % HW_;
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int py = qy - tap / 3, px = qx - tap % 3;
                    const bool in = (unsigned)py < (unsigned)TH && (unsigned)px < (unsigned)TW;
                    sm.g[q * GS + o * 9 + tap] = in ? sm.dy[o * TH * TW + (in ? py * TW + px : 0)] : 0.f;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            if (ch) {
                __syncthreads();
                put_tokens<false>(sm.x, regs);
                __syncthreads();
                if (ch + 1 < NCH) fetch_tokens<C, false, T>(regs, x, bimg, ty, tx, H, W, (ch + 1) * CC);
                else if (has_next) {
                    int nx, ny, nb_;
                    coords(tile + gridDim.x, nx, ny, nb_);
                    fetch_tokens<C, false, T>(regs, x, nb_, ny, nx, H, W, 0);
                    fetch_dy(tile + gridDim.x);
                }
            }
#pragma unroll
            for (int ks = 0; ks < ((TW_ABL & 1) ? 1 : NPOS / 4); ++ks) {
                const int q = 4 * ks + g;
                const float xv = sm.x[q * XS + 16 * wv + i16];
                const float g0 = sm.g[q * GS + i16], g1 = sm.g[q * GS + 16 + i16];
                acc[ch][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv, g0, acc[ch][0], 0, 0, 0);
                acc[ch][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv, g1, acc[ch][1], 0, 0, 0);
            }
        }
    }
    // acc[ch][nb][r] = dW^T[c = 64 ch + 16 wv + 4 g + r][n = 16 nb + i16]
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const int n = 16 * nb + i16;
            if (n < 27 && !(TW_ABL & 8)) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = CC * ch + 16 * wv + 4 * g + r;
                    atomicAdd(dw + ((n / 9) * C + c) * 9 + n % 9, acc[ch][nb][r]);
                }
            }
        }
    if (db) {
        __syncthreads();
        if (t < 4) sm.red[t] = 0.f;
        __syncthreads();
        atomicAdd(&sm.red[t >> 7], accb0);
        if (t < 128) atomicAdd(&sm.red[2], accb1);
        __syncthreads();
        if (t < 3) atomicAdd(db + t, sm.red[t]);
    }
}

// First VGG19 layer (My_CR.py:65, features[0..1]): Conv2d(3 -> 64, 3x3, pad 1) + bias + ReLU from an NCHW image
// [B, 3, H, W] straight into the channel-blocked layout [B, 8, H, W, 8] of the Winograd stack - the library needs a
// convolution, a bias pass, a ReLU pass and a layout change for this.  Thin on the INPUT side: 27 MACs per output, lane <->
// pixel, wave pair <-> 32 of the 64 output channels, weights as LDS broadcasts, 512-byte runs per (channel group, row) out.
__global__ __launch_bounds__(256) void conv3x3_in3_blocked_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                  const float* __restrict__ bias, float* __restrict__ y,
                                                                  int H, int W, int tiles_x, int tiles_y, int relu) {
    __shared__ __attribute__((aligned(16))) float xs[3 * NPOS];        // [c][10 x 18]
    __shared__ __attribute__((aligned(16))) float ws[27 * 64];         // [c * 9 + tap][k]
    const int t = threadIdx.x;
    const int tx = blockIdx.x % tiles_x, ty = (blockIdx.x / tiles_x) % tiles_y, bimg = blockIdx.x / (tiles_x * tiles_y);
    for (int e = t; e < 27 * 64; e += 256) ws[(e % 27) * 64 + e / 27] = w[e];          // w[k][c][tap] -> ws[c*9+tap][k]
    for (int e = t; e < 3 * NPOS; e += 256) {
        const int c = e / NPOS, pos = e % NPOS;
        const int yy = ty * TH - 1 + pos / HW_, xx = tx * TW - 1 + pos % HW_;
        xs[e] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? x[(((size_t)bimg * 3 + c) * H + yy) * W + xx] : 0.f;
    }
    __syncthreads();
    const int half = t >> 7, pix = t & 127;
    const int py = pix / TW, px = pix % TW;
    f32x4 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = *reinterpret_cast<const f32x4*>(bias + half * 32 + 4 * j);
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const float xv = xs[c * NPOS + (py + tap / 3) * HW_ + px + tap % 3];
            const float* wr = ws + (c * 9 + tap) * 64 + half * 32;                    // wave-uniform: broadcast reads
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const f32x4 wv = *reinterpret_cast<const f32x4*>(wr + 4 * j);
                acc[j] += xv * wv;
            }
        }
    const int yy = ty * TH + py, xx = tx * TW + px;
    if (yy < H && xx < W) {
        const float lo = relu ? 0.f : -__builtin_inff();
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            f32x4 v = acc[j];
            v[0] = fmaxf(v[0], lo); v[1] = fmaxf(v[1], lo); v[2] = fmaxf(v[2], lo); v[3] = fmaxf(v[3], lo);
            const int k = half * 32 + 4 * j;                                           // channel group k / 8, offset k % 8
            *reinterpret_cast<f32x4*>(y + ((((size_t)bimg * 8 + k / 8) * H + yy) * W + xx) * 8 + (k & 7)) = v;
        }
    }
}

template <int C, typename T>
int launch_all(int which, const void* a, const float* b, const float* c, void* d, float* e, int B, int H, int W, hipStream_t s) {
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH, ntiles = B * tiles_x * tiles_y;
    if (which == 0) {
        const size_t smem = sizeof(ThinFwdSmem);            // a = tokens x (T), b = w, c = bias, d = image y (float)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&thin_conv_fwd_kernel<C, false, false, T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        const int cap = 2 * dhz_num_cus();                   // two resident workgroups per CU (LDS), persistent over tiles
        hipLaunchKernelGGL((thin_conv_fwd_kernel<C, false, false, T>), dim3(ntiles < cap ? ntiles : cap), dim3(256), smem, s, (const T*)a, b, c, (float*)d, H, W, tiles_x, tiles_y, ntiles);
    } else if (which == 1) {     // a = image gradient dy (float), b = w, d = token gradient dx (T)
        hipLaunchKernelGGL((thin_conv_dgrad_kernel<C, T>), dim3(ntiles), dim3(256), 0, s, (const float*)a, b, (T*)d, H, W, tiles_x, tiles_y);
    } else {                     // a = dy (float), b -> tokens x (T) passed through c's slot: see dispatch
        const size_t smem = sizeof(ThinWgradSmem);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&thin_conv_wgrad_kernel<C, T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        const int cap = 2 * dhz_num_cus();                   // two resident workgroups per CU (LDS)
        const int grid = ntiles < cap ? ntiles : cap;
        hipLaunchKernelGGL((thin_conv_wgrad_kernel<C, T>), dim3(grid), dim3(256), smem, s, (const float*)a, (const T*)c, (float*)d, e, B, H, W, tiles_x, tiles_y);
    }
    return 0;
}

// which = 0: (x tokens, w, bias) -> y;  1: (dy, w) -> dx tokens;  2: (dy, x tokens via `c`) -> dw (d), db (e)
int dispatch(int which, int C, int dtype, const void* a, const float* b, const void* c, void* d, float* e, int B, int H, int W, hipStream_t s) {
    if (dtype == DHZ_F32) {
        if (C == 64) return launch_all<64, float>(which, a, b, (const float*)c, d, e, B, H, W, s);
        if (C == 128) return launch_all<128, float>(which, a, b, (const float*)c, d, e, B, H, W, s);
    } else if (dtype == DHZ_BF16) {
        if (C == 64) return launch_all<64, bf16s>(which, a, b, (const float*)c, d, e, B, H, W, s);
        if (C == 128) return launch_all<128, bf16s>(which, a, b, (const float*)c, d, e, B, H, W, s);
    }
    return 1;
}

}  // namespace

extern "C" int dhz_thin_conv3x3_fwd_dt(const void* x, const float* w, const float* bias, float* y, int B, int H, int W, int C,
                                       int dtype, void* stream) {
    DHZ_REQUIRE(x && w && y && B > 0 && H > 0 && W > 0, "dhz_thin_conv3x3_fwd: bad arguments");
    DHZ_REQUIRE(C == 64 || C == 128, "dhz_thin_conv3x3_fwd: C=%d unsupported (64, 128)", C);
    DHZ_REQUIRE(dtype == DHZ_F32 || dtype == DHZ_BF16, "dhz_thin_conv3x3_fwd: unknown dtype %d", dtype);
    dispatch(0, C, dtype, x, w, bias, y, nullptr, B, H, W, (hipStream_t)stream);
    DHZ_CHECK_LAUNCH("dhz_thin_conv3x3_fwd");
    return DHZ_OK;
}
extern "C" int dhz_thin_conv3x3_fwd(const float* x, const float* w, const float* bias, float* y, int B, int H, int W, int C,
                                    void* stream) {
    return dhz_thin_conv3x3_fwd_dt(x, w, bias, y, B, H, W, C, DHZ_F32, stream);
}

extern "C" int dhz_thin_conv3x3_dgrad_dt(const float* dy, const float* w, void* dx, int B, int H, int W, int C, int dtype,
                                         void* stream) {
    DHZ_REQUIRE(dy && w && dx && B > 0 && H > 0 && W > 0, "dhz_thin_conv3x3_dgrad: bad arguments");
    DHZ_REQUIRE(C == 64 || C == 128, "dhz_thin_conv3x3_dgrad: C=%d unsupported (64, 128)", C);
    DHZ_REQUIRE(dtype == DHZ_F32 || dtype == DHZ_BF16, "dhz_thin_conv3x3_dgrad: unknown dtype %d", dtype);
    dispatch(1, C, dtype, dy, w, nullptr, dx, nullptr, B, H, W, (hipStream_t)stream);
    DHZ_CHECK_LAUNCH("dhz_thin_conv3x3_dgrad");
    return DHZ_OK;
}
extern "C" int dhz_thin_conv3x3_dgrad(const float* dy, const float* w, float* dx, int B, int H, int W, int C, void* stream) {
    return dhz_thin_conv3x3_dgrad_dt(dy, w, dx, B, H, W, C, DHZ_F32, stream);
}

extern "C" int dhz_thin_conv3x3_wgrad_dt(const float* dy, const void* x, float* dw, float* db, int B, int H, int W, int C,
                                         int dtype, void* stream) {
    DHZ_REQUIRE(dy && x && dw && B > 0 && H > 0 && W > 0, "dhz_thin_conv3x3_wgrad: bad arguments");
    DHZ_REQUIRE(C == 64 || C == 128, "dhz_thin_conv3x3_wgrad: C=%d unsupported (64, 128)", C);
    DHZ_REQUIRE(dtype == DHZ_F32 || dtype == DHZ_BF16, "dhz_thin_conv3x3_wgrad: unknown dtype %d", dtype);
    dispatch(2, C, dtype, dy, nullptr, x, dw, db, B, H, W, (hipStream_t)stream);
    DHZ_CHECK_LAUNCH("dhz_thin_conv3x3_wgrad");
    return DHZ_OK;
}
extern "C" int dhz_thin_conv3x3_wgrad(const float* dy, const float* x, float* dw, float* db, int B, int H, int W, int C,
                                      void* stream) {
    return dhz_thin_conv3x3_wgrad_dt(dy, x, dw, db, B, H, W, C, DHZ_F32, stream);
}

extern "C" int dhz_thin_conv3x3_dgrad_blocked(const float* gb, const float* w, float* dx, int B, int H, int W, int C,
                                              void* stream) {
    DHZ_REQUIRE(gb && w && dx && B > 0 && H > 0 && W > 0, "dhz_thin_conv3x3_dgrad_blocked: bad arguments");
    DHZ_REQUIRE(C == 64, "dhz_thin_conv3x3_dgrad_blocked: C=%d unsupported (64)", C);
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH, ntiles = B * tiles_x * tiles_y;
    const size_t smem = sizeof(ThinFwdSmem);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&thin_conv_fwd_kernel<64, true, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    const int cap = 2 * dhz_num_cus();
    hipLaunchKernelGGL((thin_conv_fwd_kernel<64, true, true>), dim3(ntiles < cap ? ntiles : cap), dim3(256), smem, (hipStream_t)stream, gb,
                       w, nullptr, dx, H, W, tiles_x, tiles_y, ntiles);
    DHZ_CHECK_LAUNCH("dhz_thin_conv3x3_dgrad_blocked");
    return DHZ_OK;
}

extern "C" int dhz_conv3x3_in3_blocked(const float* x, const float* w, const float* bias, float* y, int B, int H, int W,
                                       int K, int relu, void* stream) {
    DHZ_REQUIRE(x && w && bias && y && B > 0 && H > 0 && W > 0, "dhz_conv3x3_in3_blocked: bad arguments");
    DHZ_REQUIRE(K == 64, "dhz_conv3x3_in3_blocked: K=%d unsupported (64)", K);
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    hipLaunchKernelGGL(conv3x3_in3_blocked_kernel, dim3(B * tiles_x * tiles_y), dim3(256), 0, (hipStream_t)stream, x, w, bias, y,
                       H, W, tiles_x, tiles_y, relu);
    DHZ_CHECK_LAUNCH("dhz_conv3x3_in3_blocked");
    return DHZ_OK;
}
