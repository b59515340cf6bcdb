// K5 fused: the whole LeFF branch of a LeWin block (M1:873 + M1:496-534) for C = 32 / 64 / 128 in ONE forward kernel, so
// that the 4C-wide hidden tensors cross HBM only as saves for the backward:
//
//   forward   out = x + s[b] * ( gelu(dwconv3x3(gelu(LN(x) W1^T + b1)) + bd) W2^T + b2 )
//
// (A backward-data twin - dz, dt, the transposed stencil, du, dxn in one persistent kernel - was built and tested in round 2
// and measured 1.3-1.7x SLOWER than the backward kernel chain at every width; it left the tree in
// round 3 with its entry point, last present in commit aa2dfe3.)
//
// One workgroup owns an 8 x 16 pixel tile of one image.  The LayerNorm-ed tile with its 1-pixel halo (180 tokens, padded
// to 12 MFMA row tiles) stays in LDS for the whole kernel; the hidden dimension is walked in chunks of HC channels:
//   P1  u[192 x HC]  = xn[192 x C] . W1[chunk]^T          v_mfma_f32_16x16x4_f32, operands by conflict-free ds_read_b128
//   P2  u + b1 -> (saved) -> GELU -> G[180 x HC] in LDS   (zero outside the image = Conv2d padding)
//   P3  t = dwconv3x3(G) + bd ; z = gelu(t) -> Z[128 x HC] in LDS (+ saved z, gelu'(t))
//   P4  y[128 x C] += Z . W2[:, chunk]^T                  accumulators live across the chunks
// with two barriers per chunk; the weight slices (W1 of the next chunk, W2 of this one) are fetched into registers at the top
// of a chunk and written to their single LDS images behind the first barrier.  The halo makes P1 compute 192 rows for 128 outputs (1.5x
// on one of the two GEMMs) - the price of never writing u or z for the consumer to read back.
// LDS images of MFMA operands are [row][k] with the 16-byte k-quads XOR-swizzled (swz below) so that a lane's four
// consecutive k come from one ds_read_b128; MFMA j of a 16-deep step contracts over k = 16 s + 4 (lane >> 4) + j.
#include "common.h"

namespace {

constexpr int TH = 8, TW = 16, HWID = TW + 2, HHGT = TH + 2;
constexpr int NPOS = HWID * HHGT;      // 180 tokens with halo
constexpr int NINT = TH * TW;          // 128 interior tokens

// float index of element (r, k) in a swizzled [rows][R] image (R = 16, 32, 64 or 128 floats per row)
template <int R>
__device__ __forceinline__ int swz(int r, int k) {
    if (R == 16) {
        const int f = (0x6C >> (2 * ((r >> 2) & 3))) & 3;          // [0, 3, 2, 1][(r >> 2) & 3]
        return r * 16 + 4 * (((k >> 2) & 3) ^ f) + (k & 3);
    }
    int kb = k >> 5;
    if (R >= 64) kb ^= (r & 1);
    return r * R + 32 * kb + 4 * (((k >> 2) & 7) ^ ((r >> 1) & 7)) + (k & 3);
}

// L6: the two weight products as six-term products on the bf16 matrix pipe (common.h; fp32-class results).  The weights arrive as bf16
// planes in MFMA fragment order (dhz_leff_prepack6) straight from L1 / L2 into registers - a wave needs 3 (C/32) KiB of W1 and 3 (C/16) KiB
// of W2 per 32-channel chunk, no LDS image; the LayerNorm output is split ONCE into the wave's A fragments (registers, reused by every chunk);
// P3 writes z as three bf16 piece images, so P4 issues no vector instruction at all.  Beside a bf16 MFMA the vector work of the SIMD's other
// wave (GELUs, stencil) issues half of the time; beside an fp32 MFMA it does not.
template <int C, int HC, int NW, bool L6 = false>
struct FwdCfg {
    static constexpr int NTHR = 64 * NW;
    static constexpr int Ch = 4 * C;
    static constexpr int NCHUNK = Ch / HC;
    static constexpr int XN_F = L6 ? (C / 32) * 3 * 192 * 16      // L6: bf16 piece images [k-block][piece][192 rows][64 bytes] (rows 180 .. 191: junk, their results are dropped)
                                   : NPOS * C;             // rows 180..191 read by the MFMAs fall into the next region
    static constexpr int W1_F = HC * C, W2_F = C * HC;
    static constexpr int G_F = NPOS * HC, Z_F = NINT * HC;
    static constexpr int Z_IMG_F = L6 ? 3 * NINT * 16 : Z_F;      // L6: three bf16 piece images of 64-byte rows (HC = 32)
    static constexpr int OFF_W1 = XN_F, OFF_W2 = OFF_W1 + (L6 ? 0 : W1_F), OFF_G = OFF_W2 + (L6 ? 0 : W2_F), OFF_Z = OFF_G + G_F;
    static constexpr int OFF_U = OFF_Z + Z_IMG_F;         // raw u of the interior tokens (training): stored by P3 as full lines
    static constexpr int TOTAL_F = OFF_U + Z_F;
    static constexpr size_t SMEM = (size_t)TOTAL_F * sizeof(float);
    static_assert(NINT * (C + 4) <= XN_F + (L6 ? G_F : 0), "epilogue staging must fit in the xn image (L6: + the dead G image behind it)");
    static_assert(12 * 16 * C <= OFF_U, "padded rows of the xn image must stay inside the allocation");
    static_assert(!L6 || HC == 32, "L6: one 32-deep k-block of the hidden dimension per chunk");
};

template <int C, int HC, int NW, bool L6 = false>
__global__ __launch_bounds__(64 * NW) void leff_fused_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ W1, const float* __restrict__ b1, const float* __restrict__ wd, const float* __restrict__ bd,
    const float* __restrict__ W2, const float* __restrict__ b2, const float* __restrict__ scale, float* __restrict__ out,
    float* __restrict__ xn_save, float* __restrict__ stats_save, float* __restrict__ u_save, float* __restrict__ tp_save,
    float* __restrict__ z_save, int Hres, int Wres, int tiles_x, int tiles_y) {
    // L6: W1 points at the planes of dhz_leff_prepack6 (both weights), W2 is not read
    using Cfg = FwdCfg<C, HC, NW, L6>;
    constexpr int NTHR = Cfg::NTHR, Ch = Cfg::Ch, NCHUNK = Cfg::NCHUNK;
    constexpr int NCT1 = HC / 16;                 // column tiles of P1
    static_assert(NW / NCT1 == 4, "P1: four wave groups of three row tiles");
    constexpr int RT2 = 8 / NW, CT2 = C / 16;     // P4: row tiles per wave, column tiles (all of them)
    constexpr int Q = HC / 4;                     // channel quads per chunk
    constexpr int NPS = NTHR / Q;                 // pixel slots of P3
    static_assert(NINT % NPS == 0, "P3 pixel loop");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const XN = smem;
    float* const GS = smem + Cfg::OFF_G;
    float* const ZS = smem + Cfg::OFF_Z;
    float* const US = smem + Cfg::OFF_U;

    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int i16 = lane & 15, g = lane >> 4;
    int bid = blockIdx.x;
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int bimg = bid / tiles_y;
    const int x0 = tx * TW - 1, y0 = ty * TH - 1;
    const size_t tokbase = (size_t)bimg * Hres * Wres;
    const bool train = u_save != nullptr;

    // ---- weight slices: single LDS images.  W1[chunk + 1] (read by P1 only) and W2[:, chunk] (read by P4 only) are fetched into
    //      registers at the top of a chunk and written behind the chunk's first barrier, when no wave can still be reading them.
    constexpr int NWV = (HC * C / 4 + NTHR - 1) / NTHR;     // float4 per thread per weight slice
    f32x4 rw1[NWV], rw2[NWV];
    float* const W1S = smem + Cfg::OFF_W1;
    float* const W2S = smem + Cfg::OFF_W2;
    auto wload1 = [&](int hc0) {
#pragma unroll
        for (int i = 0; i < NWV; ++i) {
            const int e = t + NTHR * i;
            if ((HC * C / 4) % NTHR == 0 || e < HC * C / 4)
                rw1[i] = *reinterpret_cast<const f32x4*>(W1 + (size_t)(hc0 + e / (C / 4)) * C + 4 * (e % (C / 4)));
        }
    };
    auto wload2 = [&](int hc0) {
#pragma unroll
        for (int i = 0; i < NWV; ++i) {
            const int e = t + NTHR * i;
            if ((HC * C / 4) % NTHR == 0 || e < HC * C / 4)
                rw2[i] = *reinterpret_cast<const f32x4*>(W2 + (size_t)(e / Q) * Ch + hc0 + 4 * (e % Q));
        }
    };
    auto wwrite1 = [&]() {
#pragma unroll
        for (int i = 0; i < NWV; ++i) {
            const int e = t + NTHR * i;
            if ((HC * C / 4) % NTHR == 0 || e < HC * C / 4) *reinterpret_cast<f32x4*>(&W1S[swz<C>(e / (C / 4), 4 * (e % (C / 4)))]) = rw1[i];
        }
    };
    auto wwrite2 = [&]() {
#pragma unroll
        for (int i = 0; i < NWV; ++i) {
            const int e = t + NTHR * i;
            if ((HC * C / 4) % NTHR == 0 || e < HC * C / 4) *reinterpret_cast<f32x4*>(&W2S[swz<HC>(e / Q, 4 * (e % Q))]) = rw2[i];
        }
    };
    if constexpr (!L6) wload1(0);

    // ---- LayerNorm (norm2) of the 180 tokens of tile + halo -> XN (swizzled); out-of-image tokens are clamped copies whose
    //      hidden activations are zeroed in P2
    {
        constexpr int LPT = C / 4;                 // lanes per token, one float4 each
        constexpr int TPP = NTHR / LPT;            // tokens per pass
        const int li = t % LPT, sub = t / LPT;
        const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + 4 * li);
        const f32x4 bt = *reinterpret_cast<const f32x4*>(beta + 4 * li);
        constexpr float invC = 1.0f / (float)C;
#pragma unroll 2
        for (int r0 = 0; r0 < NPOS; r0 += TPP) {
            const int r = r0 + sub;
            const int rc = r < NPOS ? r : NPOS - 1;
            const int hy = rc / HWID, hx = rc % HWID;
            const int yy = y0 + hy, xx = x0 + hx;
            const int yc = min(max(yy, 0), Hres - 1), xc = min(max(xx, 0), Wres - 1);
            const size_t tok = tokbase + (size_t)yc * Wres + xc;
            const f32x4 xv = *reinterpret_cast<const f32x4*>(x + tok * C + 4 * li);
            float s = xv[0] + xv[1] + xv[2] + xv[3];
#pragma unroll
            for (int o = 1; o < LPT; o <<= 1) s += __shfl_xor(s, o);
            const float mean = s * invC;
            const f32x4 dv = xv - mean;
            float var = dv[0] * dv[0] + dv[1] * dv[1] + dv[2] * dv[2] + dv[3] * dv[3];
#pragma unroll
            for (int o = 1; o < LPT; o <<= 1) var += __shfl_xor(var, o);
            const float rstd = rsqrtf(var * invC + 1e-5f);
            f32x4 y;
#pragma unroll
            for (int c = 0; c < 4; ++c) y[c] = dv[c] * rstd * gm[c] + bt[c];
            if (r < NPOS) {
                if constexpr (L6) {
                    // the token's 4 channels as bf16 pieces: k-block (4 li) / 32, 16-byte chunk ((4 li) % 32) / 8, half li & 1
                    uint32_t h0, m0, l0, h1, m1, l1;
                    dhz_split2x3(y[0], y[1], h0, m0, l0);
                    dhz_split2x3(y[2], y[3], h1, m1, l1);
                    unsigned char* xp = reinterpret_cast<unsigned char*>(XN) + ((4 * li) >> 5) * (3 * 192 * 64) + dhz_off64(r, ((4 * li) & 31) >> 3) + 8 * (li & 1);
                    *reinterpret_cast<uint2*>(xp) = make_uint2(h0, h1);
                    *reinterpret_cast<uint2*>(xp + 192 * 64) = make_uint2(m0, m1);
                    *reinterpret_cast<uint2*>(xp + 2 * 192 * 64) = make_uint2(l0, l1);
                } else
                *reinterpret_cast<f32x4*>(&XN[swz<C>(r, 4 * li)]) = y;
                const bool interior = hy >= 1 && hy <= TH && hx >= 1 && hx <= TW;
                if (train && interior) {
                    *reinterpret_cast<f32x4*>(xn_save + tok * C + 4 * li) = y;
                    if (li == 0) *reinterpret_cast<float2*>(stats_save + 2 * tok) = make_float2(mean, rstd);
                }
            }
        }
    }
    if constexpr (!L6) wwrite1();

    // ---- per-lane bookkeeping of the P1 rows this lane owns in the accumulator layout: row = 16 (rt0 + a) + 4 g + j
    const int rt0 = 3 * (w & 3), ct1 = w >> 2;
    int tokoff[3][4];          // interior rows: index of the token inside the 8 x 16 tile (0..127), -1 otherwise
    unsigned inmask = 0;       // bit (4 a + j): row is inside the image (and < 180)
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = 16 * (rt0 + a) + 4 * g + j;
            const int hy = r / HWID, hx = r % HWID;
            const int yy = y0 + hy, xx = x0 + hx;
            const bool in = r < NPOS && yy >= 0 && yy < Hres && xx >= 0 && xx < Wres;
            const bool interior = hy >= 1 && hy <= TH && hx >= 1 && hx <= TW;
            if (in) inmask |= 1u << (4 * a + j);
            tokoff[a][j] = (in && interior) ? (hy - 1) * TW + (hx - 1) : -1;
        }

    f32x4 yacc[RT2][CT2];
#pragma unroll
    for (int a = 0; a < RT2; ++a)
#pragma unroll
        for (int b = 0; b < CT2; ++b) yacc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    __syncthreads();

    // L6: the LayerNorm output lies in LDS as bf16 piece images (written once, read by every chunk's P1); k-block kb <-> channels 32 kb + 8 g + e
    constexpr int KB1 = C / 32;
    const uint16_t* const w6 = reinterpret_cast<const uint16_t*>(W1);
    constexpr size_t W2P_OFF = (size_t)(4 * C / 16) * KB1 * 3 * 512;         // elements of the W1 planes; the W2 planes follow
    const unsigned char* const XN6 = reinterpret_cast<const unsigned char*>(XN);
    unsigned char* const Z6 = reinterpret_cast<unsigned char*>(ZS);            // L6: hi | mid | lo images of z, 128 x 64 bytes each
    // W1 fragments of (hidden 16-row tile ht, k-block kb, piece): run ((ht KB1 + kb) 3 + piece) of 1 KiB; W2 fragments of (hidden k-block hb,
    // column tile b, piece): run ((hb C/16 + b) 3 + piece) behind them
    dhz_u32x4 w1f[L6 ? KB1 : 1][3], w2f[L6 ? CT2 : 1][3];
    auto load_w1f = [&](int hc0) {
        const int ht = (hc0 >> 4) + ct1;
#pragma unroll
        for (int kb = 0; kb < KB1; ++kb)
#pragma unroll
            for (int pc = 0; pc < 3; ++pc)
                w1f[kb][pc] = *reinterpret_cast<const dhz_u32x4*>(w6 + ((size_t)(ht * KB1 + kb) * 3 + pc) * 512 + lane * 8);
    };
    auto load_w2f = [&](int hc0) {
        const int hb = hc0 >> 5;
#pragma unroll
        for (int b = 0; b < CT2; ++b)
#pragma unroll
            for (int pc = 0; pc < 3; ++pc)
                w2f[b][pc] = *reinterpret_cast<const dhz_u32x4*>(w6 + W2P_OFF + ((size_t)(hb * CT2 + b) * 3 + pc) * 512 + lane * 8);
    };
    if constexpr (L6) load_w1f(0);

    const int c4 = t % Q, ps = t / Q;       // P3: channel quad and pixel slot of this thread
#pragma unroll 1
    for (int ck = 0; ck < NCHUNK; ++ck) {
        const int hc0 = ck * HC;
        const bool more = ck + 1 < NCHUNK;
        if constexpr (!L6) {
            wload2(hc0);
            if (more) wload1(hc0 + HC);
        }
        // depthwise weights / bias of this thread's 4 channels (L2-resident, consumed in P3)
        f32x4 wkv[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) wkv[i] = *reinterpret_cast<const f32x4*>(wd + (size_t)(hc0 + 4 * c4) * 9 + 4 * i);
        const f32x4 bdv = *reinterpret_cast<const f32x4*>(bd + hc0 + 4 * c4);
        const float b1v = b1[hc0 + 16 * ct1 + i16];

        // ---- P1: u = xn . W1[chunk]^T for row tiles rt0..rt0+2, column tile ct1
        f32x4 acc[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (L6) {
            constexpr int TA[6] = {2, 0, 1, 0, 1, 0}, TB[6] = {0, 2, 1, 1, 0, 0};          // (token piece, weight piece): lh hl mm hm mh hh
#pragma unroll
            for (int kb = 0; kb < KB1; ++kb) {
                dhz_u32x4 a1[3][3];
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int pc = 0; pc < 3; ++pc)
                        a1[a][pc] = *reinterpret_cast<const dhz_u32x4*>(XN6 + (kb * 3 + pc) * (192 * 64) + dhz_off64(16 * (rt0 + a) + i16, g));
#pragma unroll
                for (int term = 0; term < 6; ++term)
#pragma unroll
                    for (int a = 0; a < 3; ++a) acc[a] = dhz_mfma_bf16(a1[a][TA[term]], w1f[kb][TB[term]], acc[a]);
            }
            load_w2f(hc0);                          // this chunk's W2 fragments: in flight during P2 / P3
            if (more) load_w1f(hc0 + HC);           // the next chunk's W1 fragments
        } else
#pragma unroll
        for (int s = 0; s < C / 16; ++s) {
            const f32x4 bf = *reinterpret_cast<const f32x4*>(&W1S[swz<C>(16 * ct1 + i16, 16 * s + 4 * g)]);
            f32x4 af[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) af[a] = *reinterpret_cast<const f32x4*>(&XN[swz<C>(16 * (rt0 + a) + i16, 16 * s + 4 * g)]);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int a = 0; a < 3; ++a) acc[a] = mfma16(af[a][j], bf[j], acc[a]);
        }
        // ---- P2: + b1, save u, GELU -> G
        {
            const int hcol = 16 * ct1 + i16;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const f32x4 uv = acc[a] + b1v;
                const f32x4 gq = gelu_f4(uv);                                            // four rows of this column at once (packed math)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r = 16 * (rt0 + a) + 4 * g + j;
                    if (train && tokoff[a][j] >= 0) US[tokoff[a][j] * HC + hcol] = uv[j];     // -> P3 stores it with z and gelu'(t)
                    const float gv = ((inmask >> (4 * a + j)) & 1) ? gq[j] : 0.f;
                    if (r < NPOS) GS[r * HC + hcol] = gv;
                }
            }
        }
        __syncthreads();
        if constexpr (!L6) {
            wwrite2();
            if (more) wwrite1();
        }
        // ---- P3: t = dwconv3x3(G) + bd, z = gelu(t) -> Z (+ saves)
#pragma unroll
        for (int it = 0; it < NINT / NPS; ++it) {
            const int p = ps + NPS * it;
            const int py = p / TW, px = p % TW;
            f32x4 tacc = bdv;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const f32x4 gv = *reinterpret_cast<const f32x4*>(&GS[((py + ky) * HWID + px + kx) * HC + 4 * c4]);
                    const int k = ky * 3 + kx;
                    // wkv holds the 4 channels' 9 taps back to back: channel c, tap k = element 9 c + k
#pragma unroll
                    for (int c = 0; c < 4; ++c) tacc[c] += wkv[(9 * c + k) >> 2][(9 * c + k) & 3] * gv[c];
                }
            f32x4 zz, zp;
            gelu_both4(tacc, zz, zp);
            if constexpr (L6) {
                // z of the thread's 4 channels as bf16 pieces: 8 bytes per image, row p, 16-byte chunk c4 >> 1, half c4 & 1
                uint32_t h0, m0, l0, h1, m1, l1;
                dhz_split2x3(zz[0], zz[1], h0, m0, l0);
                dhz_split2x3(zz[2], zz[3], h1, m1, l1);
                unsigned char* zp8 = Z6 + dhz_off64(p, c4 >> 1) + 8 * (c4 & 1);
                *reinterpret_cast<uint2*>(zp8) = make_uint2(h0, h1);
                *reinterpret_cast<uint2*>(zp8 + NINT * 64) = make_uint2(m0, m1);
                *reinterpret_cast<uint2*>(zp8 + 2 * NINT * 64) = make_uint2(l0, l1);
            } else
            *reinterpret_cast<f32x4*>(&ZS[swz<HC>(p, 4 * c4)]) = zz;
            if (train) {
                const size_t o = (tokbase + (size_t)(y0 + 1 + py) * Wres + (x0 + 1 + px)) * Ch + hc0 + 4 * c4;
                *reinterpret_cast<f32x4*>(z_save + o) = zz;
                *reinterpret_cast<f32x4*>(tp_save + o) = zp;
                *reinterpret_cast<f32x4*>(u_save + o) = *reinterpret_cast<const f32x4*>(&US[p * HC + 4 * c4]);
            }
        }
        __syncthreads();
        // ---- P4: y += Z . W2[:, chunk]^T
        if constexpr (L6) {
            constexpr int TA[6] = {2, 0, 1, 0, 1, 0}, TB[6] = {0, 2, 1, 1, 0, 0};
            dhz_u32x4 zf[RT2][3];
#pragma unroll
            for (int a = 0; a < RT2; ++a)
#pragma unroll
                for (int pc = 0; pc < 3; ++pc)
                    zf[a][pc] = *reinterpret_cast<const dhz_u32x4*>(Z6 + pc * NINT * 64 + dhz_off64(16 * (RT2 * w + a) + i16, g));
#pragma unroll
            for (int term = 0; term < 6; ++term)
#pragma unroll
                for (int a = 0; a < RT2; ++a)
#pragma unroll
                    for (int b = 0; b < CT2; ++b) yacc[a][b] = dhz_mfma_bf16(zf[a][TA[term]], w2f[b][TB[term]], yacc[a][b]);
        } else
#pragma unroll
        for (int s = 0; s < HC / 16; ++s) {
            f32x4 af[RT2], bf[CT2];
#pragma unroll
            for (int a = 0; a < RT2; ++a) af[a] = *reinterpret_cast<const f32x4*>(&ZS[swz<HC>(16 * (RT2 * w + a) + i16, 16 * s + 4 * g)]);
#pragma unroll
            for (int b = 0; b < CT2; ++b) bf[b] = *reinterpret_cast<const f32x4*>(&W2S[swz<HC>(16 * b + i16, 16 * s + 4 * g)]);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int a = 0; a < RT2; ++a)
#pragma unroll
                    for (int b = 0; b < CT2; ++b) yacc[a][b] = mfma16(af[a][j], bf[b][j], yacc[a][b]);
        }
    }

    // ---- epilogue: y -> LDS (the xn image is dead) -> out = x + s[b] (y + b2), full-line stores
    constexpr int SO = C + 4;
    float* OS = smem;
#pragma unroll
    for (int a = 0; a < RT2; ++a)
#pragma unroll
        for (int b = 0; b < CT2; ++b)
#pragma unroll
            for (int j = 0; j < 4; ++j) OS[(16 * (RT2 * w + a) + 4 * g + j) * SO + 16 * b + i16] = yacc[a][b][j];
    __syncthreads();
    const float sc = scale ? scale[bimg] : 1.0f;
    for (int e = t; e < NINT * (C / 4); e += NTHR) {
        const int p = e / (C / 4), cq = e % (C / 4);
        const int py = p / TW, px = p % TW;
        const size_t o = (tokbase + (size_t)(y0 + 1 + py) * Wres + (x0 + 1 + px)) * C + 4 * cq;
        const f32x4 yv = *reinterpret_cast<const f32x4*>(&OS[p * SO + 4 * cq]);
        const f32x4 bv = *reinterpret_cast<const f32x4*>(b2 + 4 * cq);
        const f32x4 xv = *reinterpret_cast<const f32x4*>(x + o);
        *reinterpret_cast<f32x4*>(out + o) = xv + sc * (yv + bv);
    }
}

template <int C, int HC, int NW, bool L6 = false>
int launch_fwd(const float* x, const float* gamma, const float* beta, const float* W1, const float* b1, const float* wd,
               const float* bd, const float* W2, const float* b2, const float* scale, float* out, float* xn_save,
               float* stats_save, float* u_save, float* tp_save, float* z_save, int B, int Hres, int Wres, hipStream_t s) {
    using Cfg = FwdCfg<C, HC, NW, L6>;
    const int tiles_x = Wres / TW, tiles_y = Hres / TH;
    auto kern = &leff_fused_fwd_kernel<C, HC, NW, L6>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::SMEM);
    // the raw-u staging image is the last LDS region and only exists in training mode (two inference workgroups fit a CU without it)
    const size_t smem = u_save ? Cfg::SMEM : Cfg::SMEM - (size_t)Cfg::Z_F * sizeof(float);
    hipLaunchKernelGGL(kern, dim3(B * tiles_x * tiles_y), dim3(Cfg::NTHR), smem, s, x, gamma, beta, W1, b1, wd, bd, W2, b2,
                       scale, out, xn_save, stats_save, u_save, tp_save, z_save, Hres, Wres, tiles_x, tiles_y);
    return 0;
}


// six-term planes of W1 [4C][C] and W2 [C][4C] in the fragment order of the L6 kernel (1 KiB runs = 64 lanes x 8 bf16):
//   W1: run ((ht C/32 + kb) 3 + piece), element (lane = 16 g + i16, e) = piece of W1[16 ht + i16][32 kb + 8 g + e]      (ht: 16-row tile of the hidden dimension)
//   W2: behind them, run ((hb C/16 + b) 3 + piece), element = piece of W2[16 b + i16][32 hb + 8 g + e]                   (hb: 32-deep k-block of the hidden dimension)
__global__ void leff_prepack6_kernel(const float* __restrict__ w1, const float* __restrict__ w2, uint16_t* __restrict__ out, int C) {
    const int n1 = (4 * C / 16) * (C / 32) * 512, n2 = (4 * C / 32) * (C / 16) * 512;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n1 + n2) return;
    float x;
    uint16_t* o;
    if (t < n1) {
        const int e = t & 7, lane = (t >> 3) & 63, rest = t >> 9;
        const int kb = rest % (C / 32), ht = rest / (C / 32);
        x = w1[(size_t)(16 * ht + (lane & 15)) * C + 32 * kb + 8 * (lane >> 4) + e];
        o = out + ((size_t)(ht * (C / 32) + kb) * 3) * 512 + lane * 8 + e;
    } else {
        const int f = t - n1;
        const int e = f & 7, lane = (f >> 3) & 63, rest = f >> 9;
        const int b = rest % (C / 16), hb = rest / (C / 16);
        x = w2[(size_t)(16 * b + (lane & 15)) * (4 * C) + 32 * hb + 8 * (lane >> 4) + e];
        o = out + (size_t)3 * n1 + ((size_t)(hb * (C / 16) + b) * 3) * 512 + lane * 8 + e;
    }
    const float hi = __uint_as_float(__float_as_uint(x) & 0xffff0000u);
    const float r1 = x - hi;
    const float mid = __uint_as_float(__float_as_uint(r1) & 0xffff0000u);
    const float r2 = r1 - mid;
    o[0] = (uint16_t)(__float_as_uint(x) >> 16);
    o[512] = (uint16_t)(__float_as_uint(r1) >> 16);
    o[1024] = (uint16_t)(__float_as_uint(r2) >> 16);
}

}  // namespace

extern "C" int dhz_leff_prepack6(const float* w1, const float* w2, void* w6, int C, void* stream) {
    DHZ_REQUIRE(w1 && w2 && w6, "dhz_leff_prepack6: null pointer");
    DHZ_REQUIRE(C == 32 || C == 64, "dhz_leff_prepack6: C=%d unsupported (32, 64)", C);
    const int n = (4 * C / 16) * (C / 32) * 512 + (4 * C / 32) * (C / 16) * 512;
    hipLaunchKernelGGL(leff_prepack6_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, w1, w2, (uint16_t*)w6, C);
    DHZ_CHECK_LAUNCH("dhz_leff_prepack6");
    return DHZ_OK;
}

// dhz_leff_fused_fwd with the two weight products as six-term products on the bf16 matrix pipe: w6 = the planes of dhz_leff_prepack6
extern "C" int dhz_leff_fused_fwd6(const float* x, const float* gamma, const float* beta, const void* w6, const float* b1,
                                   const float* wd, const float* bd, const float* b2, const float* drop_scale,
                                   float* out, float* xn_save, float* stats_save, float* u_save, float* tp_save, float* z_save,
                                   int B, int Hres, int Wres, int C, void* stream) {
    DHZ_REQUIRE(x && gamma && beta && w6 && b1 && wd && bd && b2 && out, "dhz_leff_fused_fwd6: null pointer");
    DHZ_REQUIRE(C == 32 || C == 64, "dhz_leff_fused_fwd6: C=%d (supported: 32, 64)", C);
    DHZ_REQUIRE(B > 0 && Hres > 0 && Wres > 0 && Hres % TH == 0 && Wres % TW == 0,
                "dhz_leff_fused_fwd6: map %dx%d must be a multiple of the %dx%d tile", Hres, Wres, TH, TW);
    DHZ_REQUIRE(((uintptr_t)w6 & 15) == 0, "dhz_leff_fused_fwd6: the planes must be 16-byte aligned");
    const bool all = xn_save && stats_save && u_save && tp_save && z_save;
    const bool none = !xn_save && !stats_save && !u_save && !tp_save && !z_save;
    DHZ_REQUIRE(all || none, "dhz_leff_fused_fwd6: the five save pointers must be all set (training) or all NULL (inference)");
    hipStream_t s = (hipStream_t)stream;
    const float* w = reinterpret_cast<const float*>(w6);
    if (C == 32) launch_fwd<32, 32, 8, true>(x, gamma, beta, w, b1, wd, bd, w, b2, drop_scale, out, xn_save, stats_save, u_save, tp_save, z_save, B, Hres, Wres, s);
    else launch_fwd<64, 32, 8, true>(x, gamma, beta, w, b1, wd, bd, w, b2, drop_scale, out, xn_save, stats_save, u_save, tp_save, z_save, B, Hres, Wres, s);
    DHZ_CHECK_LAUNCH("dhz_leff_fused_fwd6");
    return DHZ_OK;
}

extern "C" int dhz_leff_fused_fwd(const float* x, const float* gamma, const float* beta, const float* w1, const float* b1,
                                  const float* wd, const float* bd, const float* w2, const float* b2, const float* drop_scale,
                                  float* out, float* xn_save, float* stats_save, float* u_save, float* tp_save, float* z_save,
                                  int B, int Hres, int Wres, int C, void* stream) {
    DHZ_REQUIRE(x && gamma && beta && w1 && b1 && wd && bd && w2 && b2 && out, "dhz_leff_fused_fwd: null pointer");
    DHZ_REQUIRE(C == 32 || C == 64 || C == 128, "dhz_leff_fused_fwd: C=%d (supported: 32, 64, 128)", C);
    DHZ_REQUIRE(B > 0 && Hres > 0 && Wres > 0 && Hres % TH == 0 && Wres % TW == 0,
                "dhz_leff_fused_fwd: map %dx%d must be a multiple of the %dx%d tile", Hres, Wres, TH, TW);
    const bool all = xn_save && stats_save && u_save && tp_save && z_save;
    const bool none = !xn_save && !stats_save && !u_save && !tp_save && !z_save;
    DHZ_REQUIRE(all || none, "dhz_leff_fused_fwd: the five save pointers must be all set (training) or all NULL (inference)");
    hipStream_t s = (hipStream_t)stream;
    // inference (no saves): two independent 4-wave workgroups per CU walk 16-channel chunks (measured 12-25 % faster than the
    // 8-wave / 32-channel variant, which in turn is the faster one when the saves have to be written: tools/bench_leff.py)
    const int cfg = none ? 1 : 0;
    if (C == 32 && cfg == 1) launch_fwd<32, 16, 4>(x, gamma, beta, w1, b1, wd, bd, w2, b2, drop_scale, out, xn_save, stats_save, u_save, tp_save, z_save, B, Hres, Wres, s);
    else if (C == 64 && cfg == 1) launch_fwd<64, 16, 4>(x, gamma, beta, w1, b1, wd, bd, w2, b2, drop_scale, out, xn_save, stats_save, u_save, tp_save, z_save, B, Hres, Wres, s);
    else if (C == 32) launch_fwd<32, 32, 8>(x, gamma, beta, w1, b1, wd, bd, w2, b2, drop_scale, out, xn_save, stats_save, u_save, tp_save, z_save, B, Hres, Wres, s);
    else if (C == 64) launch_fwd<64, 32, 8>(x, gamma, beta, w1, b1, wd, bd, w2, b2, drop_scale, out, xn_save, stats_save, u_save, tp_save, z_save, B, Hres, Wres, s);
    else launch_fwd<128, 16, 4>(x, gamma, beta, w1, b1, wd, bd, w2, b2, drop_scale, out, xn_save, stats_save, u_save, tp_save, z_save, B, Hres, Wres, s);
    DHZ_CHECK_LAUNCH("dhz_leff_fused_fwd");
    return DHZ_OK;
}
