// BASELINE config 4 (bf16 activations): the frozen VGG19[:30] feature stack of the contrastive loss (My_CR.py:56-86) with bf16
// feature maps in NHWC ("token") layout - what torch.autocast makes of the reference's F.conv2d calls, laid out for the bf16
// matrix pipe.  A 3x3 / pad-1 convolution is an IMPLICIT GEMM on v_mfma_f32_16x16x32_bf16 (fp32 accumulation):
//     y[m = (n, h, w)][co] = sum over k = (tap, ci) of  x[n, h + dy(tap), w + dx(tap), ci] . Wp[co][k]
// with Wp = the filter repacked tap-major ([Cout][9 Cin] bf16, dhz_vgg_prepack_bf16).  A stage of the contraction (64 elements)
// lies inside ONE tap because Cin is a multiple of 64, so the A operand of a stage is the same 128-byte row gather as a token
// Linear's - shifted by (dy W + dx) Cin elements and zero-filled outside the image - and no patch matrix is ever written
// (an explicit im2col would cost 9x the feature map per layer: 1.2 GB at relu1_2 of a 24 x 256 x 256 batch).  Both operands are
// contraction-contiguous: fragments by ds_read_b128 from XOR-swizzled [row][128 B] LDS images (as csrc/linear_bf16.hip).
// Backward-data is the same kernel on the gradient map with the flipped / transposed filter (prepack with `transpose`); the
// weight gradient is never formed (the reference freezes the VGG, My_CR.py:75-77).  The epilogue fuses
//     forward        y = max(acc + bias, 0)
//     backward-data  dx = (acc + addend) * [act > 0]      (addend: the gradient arriving at the same map from its L1 tap;
//                                                           act: the saved post-ReLU map that is this layer's input)
// Also here: the 2x2 max pooling between stages (forward, and backward fused with the ReLU below it) and the two L1 distances of
// one feature tap, on bf16 NHWC maps.
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 64;

__device__ __forceinline__ f32x4 mfma_bf16(s16x8 a, s16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ int off_row(int row, int ch) { return row * 128 + 16 * (ch ^ ((row >> 1) & 7)); }
// the filter (B) image is read in the row order of csrc/linear_bf16.hip::perm_row, so that a lane owns 8 consecutive output channels
// per pair of fragment blocks (16-byte stores / mask / addend loads in 64-byte runs); its swizzle key follows that order
__device__ __forceinline__ int off_rowp(int row, int ch) { return row * 128 + 16 * (ch ^ ((((row >> 3) & 3) << 1) | ((row >> 1) & 1))); }
__device__ __forceinline__ int perm_row(int b, int r) { return 32 * (b >> 1) + 8 * (r >> 2) + 4 * (b & 1) + (r & 3); }

__device__ __forceinline__ f32x4 unpack4(uint32_t x, uint32_t y) {
    return f32x4{__uint_as_float(x << 16), __uint_as_float(x & 0xffff0000u), __uint_as_float(y << 16), __uint_as_float(y & 0xffff0000u)};
}

// persistent workgroups over [BM = 32 WM pixels] x [BN = 32 WN output channels] tiles; 4 waves as 2 x 2, wave tile 16 WM x 16 WN
template <int WM, int WN>
__global__ __launch_bounds__(256, 2) void conv3_bf16_kernel(const uint16_t* __restrict__ X, const uint16_t* __restrict__ Wp,
                                                         const float* __restrict__ bias, int relu,
                                                         const uint16_t* __restrict__ act, const uint16_t* __restrict__ addend,
                                                         uint16_t* __restrict__ Y, int M, int H, int W, int lgC, int Cout,
                                                         int tiles_n, int ntiles) {
    constexpr int BM = 32 * WM, BN = 32 * WN;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;
    constexpr int STAGE = A_BYTES + B_BYTES;
    constexpr int NA = WM, NB = WN;                              // 16-byte chunks per thread per stage
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int i16 = lane & 15, g = lane >> 4;
    const int wm = w >> 1, wn = w & 1;
    const int Cin = 1 << lgC, K = 9 << lgC;
    const int nst = K / BK;
    const int grid = gridDim.x;
    auto tile_of = [&](int i) -> int {
        const int lin = blockIdx.x + i * grid;
        if (lin >= ntiles) return -1;
        if ((grid & 7) == 0 && (ntiles & 7) == 0) return (lin & 7) * (ntiles >> 3) + (lin >> 3);     // one XCD: neighbouring tiles
        return lin;
    };
    f32x4 acc[WM][WN];
#pragma unroll
    for (int a = 0; a < WM; ++a)
#pragma unroll
        for (int b = 0; b < WN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Operand staging runs TWO stages ahead of the matrix pipe through two register sets: stage s + 2 is requested at the top of
    // stage s and written to LDS at the end of stage s + 1, so a request has two stages of MFMAs (of both resident workgroups)
    // to come back from L2 / HBM - one stage of look-ahead left the pipe waiting on every stage.  The request stream
    // is its own cursor over (tile, stage) and runs across tile boundaries; past the end of this workgroup's tiles it keeps
    // re-requesting its last position (unconditional loads keep the compiler's vmcnt accounting exact).
    u32x4 ra[2][NA], rb[2][NB];
    const uint16_t* pa[NA];
    int pyx[NA];                                                 // (row << 16) | column of the pixel whose chunk this thread stages
    const uint16_t* pb[NB];
    auto set_tile = [&](int tile) {
        const int tn = tile % tiles_n, tm = tile / tiles_n;
        const int m0 = tm * BM, n0 = tn * BN;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int e = t + 256 * i;
            const int m = min(m0 + (e >> 3), M - 1);
            const int px = m % W, py = (m / W) % H;
            pyx[i] = (py << 16) | px;
            pa[i] = X + ((size_t)m << lgC) + 8 * (e & 7);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int e = t + 256 * i;
            pb[i] = Wp + (size_t)(n0 + (e >> 3)) * K + 8 * (e & 7);
        }
    };
    auto gload = [&](int k0, u32x4 (&A)[NA], u32x4 (&B)[NB]) {
        const int tap = k0 >> lgC, c0 = k0 & (Cin - 1);
        const int dy = ((tap * 11) >> 5) - 1, dx = tap - 3 * (dy + 1) - 1;
        const int off = ((dy * W + dx) << lgC) + c0;             // elements; negative for the taps above / left
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int py = pyx[i] >> 16, px = pyx[i] & 0xffff;
            const bool ok = (unsigned)(py + dy) < (unsigned)H && (unsigned)(px + dx) < (unsigned)W;
            const u32x4 v = *reinterpret_cast<const u32x4*>(ok ? pa[i] + off : pa[i]);
            A[i] = ok ? v : u32x4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) B[i] = *reinterpret_cast<const u32x4*>(pb[i] + k0);
    };
    auto swrite = [&](int buf, const u32x4 (&A)[NA], const u32x4 (&B)[NB]) {
        unsigned char* As = smem + buf * STAGE;
        unsigned char* Bs = As + A_BYTES;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int e = t + 256 * i;
            *reinterpret_cast<u32x4*>(As + off_row(e >> 3, e & 7)) = A[i];
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int e = t + 256 * i;
            *reinterpret_cast<u32x4*>(Bs + off_rowp(e >> 3, e & 7)) = B[i];
        }
    };

    int tile = tile_of(0);
    if (tile < 0) return;
    const int mytiles = (ntiles - (int)blockIdx.x + grid - 1) / grid;
    const int total = mytiles * nst;                             // stages of this workgroup over all its tiles
    int l_ti = 0, l_st = 0;                                      // request cursor
    set_tile(tile);
    auto request = [&](u32x4 (&A)[NA], u32x4 (&B)[NB]) {
        gload(l_st * BK, A, B);
        if (++l_st == nst) {
            const int nt = tile_of(l_ti + 1);
            if (nt >= 0) { ++l_ti; l_st = 0; set_tile(nt); }
            else l_st = nst - 1;                                 // end of the stream: stay on a valid position
        }
    };
    request(ra[0], rb[0]);
    request(ra[1], rb[1]);
    swrite(0, ra[0], rb[0]);
    __syncthreads();
    const int sw = (i16 >> 1) & 7;
    int buf = 0, st = 0, ti = 0;
    // one stage: request stage gs + 2 into the set stage gs came from, multiply stage gs, stage gs + 1 into the other LDS buffer
    auto stage = [&](int gs, u32x4 (&Aq)[NA], u32x4 (&Bq)[NB], const u32x4 (&Aw)[NA], const u32x4 (&Bw)[NB]) {
        request(Aq, Bq);
        const unsigned char* As = smem + buf * STAGE;
        const unsigned char* Bs = As + A_BYTES;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            s16x8 af[WM], bf[WN];
#pragma unroll
            for (int a = 0; a < WM; ++a)
                af[a] = *reinterpret_cast<const s16x8*>(As + (wm * WM * 16 + a * 16 + i16) * 128 + 16 * ((4 * s + g) ^ sw));
#pragma unroll
            for (int b = 0; b < WN; ++b)
                bf[b] = *reinterpret_cast<const s16x8*>(Bs + off_rowp(wn * WN * 16 + perm_row(b, i16), 4 * s + g));
#pragma unroll
            for (int a = 0; a < WM; ++a)
#pragma unroll
                for (int b = 0; b < WN; ++b) acc[a][b] = mfma_bf16(bf[b], af[a], acc[a][b]);     // D = C^T block (epilogue)
        }
        if (gs + 1 < total) {
            swrite(buf ^ 1, Aw, Bw);
            __syncthreads();
            buf ^= 1;
        }
        if (++st < nst) return;
        st = 0;
        {   // acc[a][b][j] = C[pixel 16 a + i16][channel perm_row(b, 4 g + j)]: blocks 2 h, 2 h + 1 of a lane are the 8 consecutive
            // channels 32 h + 8 g .. + 7 of one pixel - one 16-byte store (and mask / addend load) per pair
            const int tn = tile % tiles_n, tm = tile / tiles_n;
            const int m0 = tm * BM + wm * WM * 16 + i16, n0 = tn * BN + wn * WN * 16 + 8 * g;
            const size_t o0 = (size_t)m0 * Cout + n0;
            f32x4 bv[WN];                                        // bias in the lane layout (loaded here: 16 registers less held
#pragma unroll                                                   // through the stages; one wait per tile of >= 9 stages)
            for (int b = 0; b < WN; ++b)
                bv[b] = bias ? *reinterpret_cast<const f32x4*>(bias + n0 + 32 * (b >> 1) + 4 * (b & 1)) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int a = 0; a < WM; ++a) {
                if (m0 + 16 * a < M) {
#pragma unroll
                    for (int h = 0; h < WN / 2; ++h) {
                        const size_t o = o0 + (size_t)(16 * a) * Cout + 32 * h;
                        f32x4 v0 = acc[a][2 * h] + bv[2 * h], v1 = acc[a][2 * h + 1] + bv[2 * h + 1];
                        if (relu) {
                            v0 = f32x4{fmaxf(v0[0], 0.f), fmaxf(v0[1], 0.f), fmaxf(v0[2], 0.f), fmaxf(v0[3], 0.f)};
                            v1 = f32x4{fmaxf(v1[0], 0.f), fmaxf(v1[1], 0.f), fmaxf(v1[2], 0.f), fmaxf(v1[3], 0.f)};
                        }
                        if (act) {
                            if (addend) {
                                const u32x4 ad = *reinterpret_cast<const u32x4*>(addend + o);
                                v0 += unpack4(ad[0], ad[1]);
                                v1 += unpack4(ad[2], ad[3]);
                            }
                            const u32x4 mk = *reinterpret_cast<const u32x4*>(act + o);
                            const f32x4 m0v = unpack4(mk[0], mk[1]), m1v = unpack4(mk[2], mk[3]);
                            v0 = f32x4{m0v[0] > 0.f ? v0[0] : 0.f, m0v[1] > 0.f ? v0[1] : 0.f, m0v[2] > 0.f ? v0[2] : 0.f, m0v[3] > 0.f ? v0[3] : 0.f};
                            v1 = f32x4{m1v[0] > 0.f ? v1[0] : 0.f, m1v[1] > 0.f ? v1[1] : 0.f, m1v[2] > 0.f ? v1[2] : 0.f, m1v[3] > 0.f ? v1[3] : 0.f};
                        }
                        u32x4 r;
                        r[0] = (uint32_t)f32_to_bf16(v0[0]) | ((uint32_t)f32_to_bf16(v0[1]) << 16);
                        r[1] = (uint32_t)f32_to_bf16(v0[2]) | ((uint32_t)f32_to_bf16(v0[3]) << 16);
                        r[2] = (uint32_t)f32_to_bf16(v1[0]) | ((uint32_t)f32_to_bf16(v1[1]) << 16);
                        r[3] = (uint32_t)f32_to_bf16(v1[2]) | ((uint32_t)f32_to_bf16(v1[3]) << 16);
                        *reinterpret_cast<u32x4*>(Y + o) = r;
                    }
                }
            }
#pragma unroll
            for (int a = 0; a < WM; ++a)
#pragma unroll
                for (int b = 0; b < WN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        ++ti;
        const int nt = tile_of(ti);
        if (nt >= 0) tile = nt;
    };
    for (int gs = 0; gs < total; gs += 2) {
        stage(gs, ra[0], rb[0], ra[1], rb[1]);
        if (gs + 1 >= total) break;
        stage(gs + 1, ra[1], rb[1], ra[0], rb[0]);
    }
}

template <int WM, int WN>
void launch_conv(const uint16_t* X, const uint16_t* Wp, const float* bias, int relu, const uint16_t* act, const uint16_t* addend,
                 uint16_t* Y, int M, int H, int W, int lgC, int Cout, hipStream_t s) {
    constexpr int BM = 32 * WM, BN = 32 * WN;
    constexpr size_t smem = 2 * (size_t)(BM * 128 + BN * 128);
    const int tiles_n = Cout / BN, tiles_m = (M + BM - 1) / BM;
    const int ntiles = tiles_n * tiles_m;
    const int slots = 2 * dhz_num_cus();                          // two resident workgroups per CU
    const int grid = ntiles < slots ? ntiles : slots;
    if (smem > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_bf16_kernel<WM, WN>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL((conv3_bf16_kernel<WM, WN>), dim3(grid), dim3(256), smem, s, X, Wp, bias, relu, act, addend, Y, M, H, W, lgC,
                       Cout, tiles_n, ntiles);
}

// filter [K][C][3][3] fp32 -> bf16 [K][9 C] (tap-major), or with `transpose` the backward-data filter [C][9 K]:
//     out[c][tap * K + k] = w[k][c][8 - tap]      (the 180-degree rotation)
__global__ __launch_bounds__(256) void vgg_prepack_bf16_kernel(const float* __restrict__ w, uint16_t* __restrict__ out, int K, int C,
                                                               int transpose) {
    const int total = K * C * 9;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        if (!transpose) {
            const int c = e % C, tap = (e / C) % 9, k = e / (9 * C);
            out[e] = f32_to_bf16(w[((size_t)k * C + c) * 9 + tap]);
        } else {
            const int k = e % K, tap = (e / K) % 9, c = e / (9 * K);
            out[e] = f32_to_bf16(w[((size_t)k * C + c) * 9 + (8 - tap)]);
        }
    }
}

// ---- 2x2 / stride-2 max pooling of NHWC bf16 maps; one thread = 8 channels (16 bytes) of one pooled pixel
__device__ __forceinline__ void unpack8(u32x4 v, float (&f)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = __uint_as_float(v[i] << 16);
        f[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u);
    }
}
__device__ __forceinline__ u32x4 pack8(const float (&f)[8]) {
    u32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = (uint32_t)f32_to_bf16(f[2 * i]) | ((uint32_t)f32_to_bf16(f[2 * i + 1]) << 16);
    return v;
}

__global__ __launch_bounds__(256) void maxpool2x2_nhwc_bf16_fwd_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y,
                                                                       long long total, int Ho, int Wo, int C8) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;      // over [n][ho][wo][C / 8]
    if (e >= total) return;
    const int c8 = (int)(e % C8);
    long long r = e / C8;
    const int wo = (int)(r % Wo);
    r /= Wo;
    const int ho = (int)(r % Ho);
    const long long n = r / Ho;
    const size_t rowb = (size_t)2 * Wo * C8 * 8;                         // elements per input row
    const uint16_t* src = x + ((size_t)(n * 2 * Ho + 2 * ho)) * rowb + (size_t)(2 * wo) * C8 * 8 + 8 * c8;
    float a[8], b[8], c[8], d[8], m[8];
    unpack8(*reinterpret_cast<const u32x4*>(src), a);
    unpack8(*reinterpret_cast<const u32x4*>(src + C8 * 8), b);
    unpack8(*reinterpret_cast<const u32x4*>(src + rowb), c);
    unpack8(*reinterpret_cast<const u32x4*>(src + rowb + C8 * 8), d);
#pragma unroll
    for (int i = 0; i < 8; ++i) m[i] = fmaxf(fmaxf(a[i], b[i]), fmaxf(c[i], d[i]));
    *reinterpret_cast<u32x4*>(y + e * 8) = pack8(m);
}

// Backward of the pooling of a post-ReLU map `act`: the gradient goes to the FIRST maximum of each window in scan order (the
// library's tie rule - ties between positive bf16 values are common) and, fused, through the ReLU below it (act > 0).
__global__ __launch_bounds__(256) void maxpool2x2_nhwc_bf16_bwd_kernel(const uint16_t* __restrict__ gy,
                                                                       const uint16_t* __restrict__ act, uint16_t* __restrict__ gx,
                                                                       long long total, int Ho, int Wo, int C8) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int c8 = (int)(e % C8);
    long long r = e / C8;
    const int wo = (int)(r % Wo);
    r /= Wo;
    const int ho = (int)(r % Ho);
    const long long n = r / Ho;
    const size_t rowb = (size_t)2 * Wo * C8 * 8;
    const size_t o = ((size_t)(n * 2 * Ho + 2 * ho)) * rowb + (size_t)(2 * wo) * C8 * 8 + 8 * c8;
    float a[8], b[8], c[8], d[8], g[8], ra[8], rb[8], rc[8], rd[8];
    unpack8(*reinterpret_cast<const u32x4*>(act + o), a);
    unpack8(*reinterpret_cast<const u32x4*>(act + o + C8 * 8), b);
    unpack8(*reinterpret_cast<const u32x4*>(act + o + rowb), c);
    unpack8(*reinterpret_cast<const u32x4*>(act + o + rowb + C8 * 8), d);
    unpack8(*reinterpret_cast<const u32x4*>(gy + e * 8), g);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float m = fmaxf(fmaxf(a[i], b[i]), fmaxf(c[i], d[i]));
        const float gv = m > 0.f ? g[i] : 0.f;
        const bool ia = a[i] == m, ib = !ia && b[i] == m, ic = !ia && !ib && c[i] == m;
        ra[i] = ia ? gv : 0.f;
        rb[i] = ib ? gv : 0.f;
        rc[i] = ic ? gv : 0.f;
        rd[i] = (!ia && !ib && !ic) ? gv : 0.f;
    }
    *reinterpret_cast<u32x4*>(gx + o) = pack8(ra);
    *reinterpret_cast<u32x4*>(gx + o + C8 * 8) = pack8(rb);
    *reinterpret_cast<u32x4*>(gx + o + rowb) = pack8(rc);
    *reinterpret_cast<u32x4*>(gx + o + rowb + C8 * 8) = pack8(rd);
}

// ---- the two L1 distances of one feature tap (My_CR.py:108-112) on bf16 maps: fp32 differences and sums, 8 elements per lane
__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__global__ __launch_bounds__(256) void l1_pair_bf16_fwd_kernel(const uint16_t* __restrict__ a, const uint16_t* __restrict__ p,
                                                               const uint16_t* __restrict__ n, float* __restrict__ sums,
                                                               int64_t n8) {
    __shared__ float part[2][4];
    float sp = 0.f, sn = 0.f;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n8; e += (int64_t)gridDim.x * 256) {
        float av[8], pv[8], nv[8];
        unpack8(reinterpret_cast<const u32x4*>(a)[e], av);
        unpack8(reinterpret_cast<const u32x4*>(p)[e], pv);
#pragma unroll
        for (int i = 0; i < 8; ++i) sp += fabsf(av[i] - pv[i]);
        if (n) {
            unpack8(reinterpret_cast<const u32x4*>(n)[e], nv);
#pragma unroll
            for (int i = 0; i < 8; ++i) sn += fabsf(av[i] - nv[i]);
        }
    }
    sp = wsum(sp);
    sn = wsum(sn);
    if ((threadIdx.x & 63) == 0) { part[0][threadIdx.x >> 6] = sp; part[1][threadIdx.x >> 6] = sn; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(sums, part[0][0] + part[0][1] + part[0][2] + part[0][3]);
        if (n) atomicAdd(sums + 1, part[1][0] + part[1][1] + part[1][2] + part[1][3]);
    }
}

__global__ __launch_bounds__(256) void l1_pair_bf16_bwd_kernel(const uint16_t* __restrict__ a, const uint16_t* __restrict__ p,
                                                               const uint16_t* __restrict__ n, const float* __restrict__ g,
                                                               float inv_n, uint16_t* __restrict__ da, int64_t n8) {
    const float cp = g[0] * inv_n, cn = n ? g[1] * inv_n : 0.f;
    auto sgn = [](float d) { return d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f); };
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n8; e += (int64_t)gridDim.x * 256) {
        float av[8], pv[8], nv[8], r[8];
        unpack8(reinterpret_cast<const u32x4*>(a)[e], av);
        unpack8(reinterpret_cast<const u32x4*>(p)[e], pv);
#pragma unroll
        for (int i = 0; i < 8; ++i) r[i] = cp * sgn(av[i] - pv[i]);
        if (n) {
            unpack8(reinterpret_cast<const u32x4*>(n)[e], nv);
#pragma unroll
            for (int i = 0; i < 8; ++i) r[i] += cn * sgn(av[i] - nv[i]);
        }
        reinterpret_cast<u32x4*>(da)[e] = pack8(r);
    }
}

int grid_for8(int64_t n, int cap) {
    const int64_t b = (n + 255) / 256;
    return (int)(b < cap ? b : cap);
}

}  // namespace

extern "C" int dhz_vgg_prepack_bf16(const float* w, void* out, int K, int C, int transpose, void* stream) {
    DHZ_REQUIRE(w && out && K > 0 && C > 0, "dhz_vgg_prepack_bf16: bad arguments");
    const int total = K * C * 9;
    hipLaunchKernelGGL(vgg_prepack_bf16_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, (uint16_t*)out, K, C,
                       transpose);
    DHZ_CHECK_LAUNCH("dhz_vgg_prepack_bf16");
    return DHZ_OK;
}

extern "C" int dhz_vgg_conv3x3_bf16(const void* x, const void* wp, const float* bias, int relu, const void* act, const void* addend,
                                    void* y, int N, int H, int W, int Cin, int Cout, void* stream) {
    DHZ_REQUIRE(x && wp && y, "dhz_vgg_conv3x3_bf16: null pointer");
    DHZ_REQUIRE(N > 0 && H > 0 && W > 0 && H < 32768 && W < 32768, "dhz_vgg_conv3x3_bf16: N=%d H=%d W=%d", N, H, W);
    DHZ_REQUIRE(Cin >= 64 && (Cin & (Cin - 1)) == 0 && Cout >= 64 && Cout % 64 == 0,
                "dhz_vgg_conv3x3_bf16: Cin=%d (power of two >= 64) Cout=%d (multiple of 64)", Cin, Cout);
    DHZ_REQUIRE(act || !addend, "dhz_vgg_conv3x3_bf16: addend without act");
    DHZ_REQUIRE((long long)N * H * W < (1ll << 31) / 2, "dhz_vgg_conv3x3_bf16: too many pixels");
    DHZ_REQUIRE((((uintptr_t)x | (uintptr_t)wp | (uintptr_t)y | (uintptr_t)act | (uintptr_t)addend) & 15) == 0,
                "dhz_vgg_conv3x3_bf16: operands must be 16-byte aligned");
    int lgC = 0;
    while ((1 << lgC) < Cin) ++lgC;
    const int M = N * H * W;
    const int wn = Cout % 128 == 0 ? 4 : 2;
    const long blocks128 = (long)((M + 127) / 128) * (Cout / (32 * wn));
    const int wm = blocks128 >= dhz_num_cus() ? 4 : 2;
    hipStream_t s = (hipStream_t)stream;
#define CASE(a, b) \
    if (wm == a && wn == b) launch_conv<a, b>((const uint16_t*)x, (const uint16_t*)wp, bias, relu, (const uint16_t*)act, (const uint16_t*)addend, (uint16_t*)y, M, H, W, lgC, Cout, s);
    CASE(4, 4) CASE(4, 2) CASE(2, 4) CASE(2, 2)
#undef CASE
    DHZ_CHECK_LAUNCH("dhz_vgg_conv3x3_bf16");
    return DHZ_OK;
}

extern "C" int dhz_maxpool2x2_nhwc_bf16_fwd(const void* x, void* y, int N, int H, int W, int C, void* stream) {
    DHZ_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && C > 0 && C % 8 == 0,
                "dhz_maxpool2x2_nhwc_bf16_fwd: bad arguments");
    const long long total = (long long)N * (H / 2) * (W / 2) * (C / 8);
    hipLaunchKernelGGL(maxpool2x2_nhwc_bf16_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t*)x, (uint16_t*)y, total, H / 2, W / 2, C / 8);
    DHZ_CHECK_LAUNCH("dhz_maxpool2x2_nhwc_bf16_fwd");
    return DHZ_OK;
}

extern "C" int dhz_maxpool2x2_nhwc_bf16_bwd(const void* gy, const void* act, void* gx, int N, int H, int W, int C, void* stream) {
    DHZ_REQUIRE(gy && act && gx && N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && C > 0 && C % 8 == 0,
                "dhz_maxpool2x2_nhwc_bf16_bwd: bad arguments");
    const long long total = (long long)N * (H / 2) * (W / 2) * (C / 8);
    hipLaunchKernelGGL(maxpool2x2_nhwc_bf16_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t*)gy, (const uint16_t*)act, (uint16_t*)gx, total, H / 2, W / 2, C / 8);
    DHZ_CHECK_LAUNCH("dhz_maxpool2x2_nhwc_bf16_bwd");
    return DHZ_OK;
}

extern "C" int dhz_l1_pair_fwd_bf16(const void* a, const void* p, const void* n, float* sums, int64_t count, void* stream) {
    DHZ_REQUIRE(a && p && sums && count > 0 && count % 8 == 0, "dhz_l1_pair_fwd_bf16: bad arguments (count must be a multiple of 8)");
    hipLaunchKernelGGL(l1_pair_bf16_fwd_kernel, dim3(grid_for8(count / 8, 3 * dhz_num_cus())), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t*)a, (const uint16_t*)p, (const uint16_t*)n, sums, count / 8);
    DHZ_CHECK_LAUNCH("dhz_l1_pair_fwd_bf16");
    return DHZ_OK;
}

extern "C" int dhz_l1_pair_bwd_bf16(const void* a, const void* p, const void* n, const float* g, void* da, int64_t count,
                                    void* stream) {
    DHZ_REQUIRE(a && p && g && da && count > 0 && count % 8 == 0, "dhz_l1_pair_bwd_bf16: bad arguments (count must be a multiple of 8)");
    hipLaunchKernelGGL(l1_pair_bf16_bwd_kernel, dim3(grid_for8(count / 8, 8 * dhz_num_cus())), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t*)a, (const uint16_t*)p, (const uint16_t*)n, g, 1.0f / (float)count, (uint16_t*)da, count / 8);
    DHZ_CHECK_LAUNCH("dhz_l1_pair_bwd_bf16");
    return DHZ_OK;
}
