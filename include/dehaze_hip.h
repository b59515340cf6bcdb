/*
 * dehaze_hip.h - C-ABI of the MI355X (gfx950) kernels behind the Uformer_ProbSparse training path.
 *
 * The reference (xin-fight/...Image-Dehazing...Vision-Transformer) is pure Python/PyTorch: it has no
 * FFI layer of its own, so every entry point below replaces a *sequence of ATen ops* in the reference;
 * the file:line each one replaces is cited (M1 = Uformer_ProbSparse/My_model_1.py, M0 = My_model.py,
 * ATT = Uformer_ProbSparse/ProbSparse/attn.py, TR = My_train.py).  The reference-side binding a
 * maintainer would add is a ctypes stub - see INTEGRATION.md.
 *
 * Conventions
 *   - plain C: raw DEVICE pointers, sizes, an explicit hipStream_t passed as void*; no torch types.
 *   - the caller owns every buffer (allocate through the framework's caching allocator); no entry
 *     point allocates, frees or synchronises; all work is enqueued on `stream`.
 *   - return 0 on success, a negative DHZ_E* code otherwise; dhz_last_error() gives the message of
 *     the last failure on the calling thread.
 *   - all floating-point tensors are fp32, contiguous unless a leading dimension `ld*` is given.
 *   - re-entrant across streams/devices; no hidden global state.
 */
#ifndef DEHAZE_HIP_H
#define DEHAZE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DHZ_OK 0
#define DHZ_EINVAL (-22)  /* bad argument (shape / null pointer / unsupported head_dim) */
#define DHZ_ELAUNCH (-5)  /* hipLaunchKernel reported an error */

/* Storage type of token tensors for the dhz_*_dt entry points (BASELINE config 4: bf16 activations in HBM, fp32 arithmetic
 * and accumulation inside every kernel, fp32 parameters / statistics / parameter gradients). */
#define DHZ_F32 0
#define DHZ_BF16 1

#define DHZ_NTOK 64 /* tokens per 8x8 window */
#define DHZ_NTOP 25 /* u = U_part = 5*ceil(ln 64)  (ATT:310-315) */

int dhz_abi_version(void);
const char* dhz_last_error(void);
/* 16 hex digits: content hash of the sources (every .hip file under csrc/, common.h, this header) the loaded library was built from.  The PMC
 * passes stamp it into profiles/pmc_traffic.json; bench.py reports `roofline.traffic` only when the stamp matches. */
const char* dhz_build_id(void);
/* Compute units the persistent grids of this library leave free (default 0 = none).  One process per GPU with the gradient exchange
 * overlapping the backward pass (replaces nn.DataParallel, My_train.py:97): RCCL's collective kernels need CUs beside grids that are sized
 * "resident workgroups per CU x CUs"; dhz_set_reserved_cus(k) sizes every such grid for (CUs - k).  Results do not depend on it.
 * dhz_grid_cus() = the CU count the grids are sized for (physical - reserved). */
int dhz_set_reserved_cus(int k);
int dhz_get_reserved_cus(void);
int dhz_grid_cus(void);

/* ---------------------------------------------------------------------------------------------
 * K3  ProbSparse window attention core.   Replaces ProbAttention.forward  ATT:287-342
 *     (_prob_QK ATT:71-152, _get_initial_context ATT:154-176, _update_context ATT:178-281).
 *
 * q,k,v : [B_, 64, H, d] fp32 with token stride `ld` floats (element (b,n,h,e) at
 *         ((b*64+n)*ld + h*d + e)); ld = H*d for separate projections, 3*H*d for a packed QKV.
 * idx   : [64, 25] uint8 sampled key ids, shared by all windows/heads (ATT:91).
 * bias  : [H, 64, 64] relative-position bias (M1:408-410) or NULL (options.is_relative_position_bias
 *         False, ATT:227-232).   mask: [nW, 64, 64] (0 / -100) or NULL; window id = b mod nW.
 * out   : [B_, 64, H, d], token stride ldo.
 * rank  : [B_, H, 64] uint8 - for each query its position (0..24, by descending sparsity measure M)
 *         in the top-u set, 255 if not selected.  Saved for backward.
 * d must be 32 or 64.
 */
int dhz_ps_attn_fwd(const float* q, const float* k, const float* v, int ld, const uint8_t* idx,
                    const float* bias, const float* mask, float* out, int ldo, uint8_t* rank,
                    int B_, int H, int nW, int d, void* stream);

/* Backward of the above (autograd of ATT:287-342; gradient flows through steps 6-12 only).
 * dout: [B_,64,H,d] stride ldo.  dq,dk,dv: [B_,64,H,d] stride ldg (every element written).
 * dbias_part: workspace [dhz_ps_attn_bwd_parts_d(B_,H,d), 64, 64] fp32 (written, not accumulated) or
 *             NULL when bias == NULL.  Row p holds the partial bias gradient of head (p % H);
 *             reduce with dhz_bias_table_grad.
 */
int dhz_ps_attn_bwd_parts(int B_, int H);            /* dhz_dense_attn_bwd */
int dhz_ps_attn_bwd_parts_d(int B_, int H, int d);   /* dhz_ps_attn_bwd: its persistent workgroups (two per CU) */
int dhz_ps_attn_bwd(const float* q, const float* k, const float* v, int ld, const float* bias,
                    const float* mask, const uint8_t* rank, const float* dout, int ldo, float* dq,
                    float* dk, float* dv, int ldg, float* dbias_part, int B_, int H, int nW, int d,
                    void* stream);

/* K1+K2+K3+K4  Fused attention branch of a LeWin block, forward, for C = 32, 64, 128 (head_dim 32):
 *     out = x + drop_scale[b] * out_projection(ProbAttention(Q,K,V = projections(window(roll(norm1(x))))))
 *     Replaces M1:839-872 + ATT:385-461 in one kernel (one workgroup per 8x8 window, per-head QKV GEMM and
 *     the out-projection on the fp32 matrix pipe around the LDS-resident ProbSparse core).
 * dhz_fused_attn_prepack: reorders the four [C,C] projection weights into MFMA B-fragment order
 *     (wqkv_p: 3*C*C floats, wo_p: C*C floats); call whenever the weights change.
 * x,out: [B, Hres*Wres, C] tokens.  bqkv = [bq|bk|bv] (3C), bo (C).  idx [64,25] uint8.  bias [H,64,64] or
 *     NULL.  mask [nW,64,64] (shifted blocks) or NULL.  drop_scale [B] or NULL (DropPath keep/keep_prob).
 * Training mode - all five save pointers non-NULL, window-ordered rows r = (b*nW + w)*64 + token:
 *     xn_save [T,C] (LayerNorm output), qkv_save [T,3C], ctx_save [T,C], stats_save [B*HW,2] (mean, rstd by
 *     source token), rank_save [B*nW, H, 64] - exactly what dhz_ps_attn_bwd / dhz_linear_wgrad /
 *     dhz_ln_partition_bwd consume.  Inference mode: pass NULL for all five. */
int dhz_fused_attn_prepack(const float* wq, const float* wk, const float* wv, const float* wo,
                           float* wqkv_p, float* wo_p, int C, void* stream);
/* ... for n <= 16 blocks in ONE launch (host arrays of device pointers; C[i] in {32, 64, 128}).  Same values as n calls of
 * dhz_fused_attn_prepack. */
int dhz_fused_attn_prepack_multi(const float* const* wq, const float* const* wk, const float* const* wv, const float* const* wo,
                                 float* const* wqkv_p, float* const* wo_p, const int* C, int n, void* stream);
int dhz_fused_window_attn_fwd(const float* x, const float* gamma, const float* beta, const float* wqkv_p,
                              const float* bqkv, const float* wo_p, const float* bo, const uint8_t* idx,
                              const float* bias, const float* mask, const float* drop_scale, float* out,
                              float* xn_save, float* qkv_save, float* ctx_save, float* stats_save,
                              uint8_t* rank_save, int B, int Hres, int Wres, int C, int shift,
                              void* stream);
/* The same kernel with the Q / K / V product of every head on the bf16 matrix pipe in the six-term form of dhz_linear_fwd_split6 (three bf16
 * truncation pieces per operand value, products hh hm mh hl lh mm, dropped terms <= 2^-24 relative, fp32 accumulation: fp32-class results).
 * So is the out-projection (its planes in the S tile while P V runs).  wqkv6_p: the planes of the four weights in the kernel's fragment order
 * (dhz_fused_attn_prepack6: (C/32) ((C/64) 36 + (C/16) 3) KiB of bf16), brought ONCE per workgroup and head by LDS-DMA into tiles that are dead
 * at that point - not once per wave through L1.  Every other argument as dhz_fused_window_attn_fwd (wo_p is not read).  C = 64. */
int dhz_fused_window_attn_fwd6(const float* x, const float* gamma, const float* beta, const void* wqkv6_p, const float* bqkv,
                               const float* wo_p, const float* bo, const uint8_t* idx, const float* bias, const float* mask,
                               const float* drop_scale, float* out, float* xn_save, float* qkv_save, float* ctx_save, float* stats_save,
                               uint8_t* rank_save, int B, int Hres, int Wres, int C, int shift, void* stream);
int dhz_fused_attn_prepack6(const float* wq, const float* wk, const float* wv, const float* wo, void* wqkv6_p, int C, void* stream);
/* ... for n <= 16 blocks in ONE launch (host arrays of device pointers). */
int dhz_fused_attn_prepack6_multi(const float* const* wq, const float* const* wk, const float* const* wv, const float* const* wo,
                                  void* const* wqkv6_p, const int* C, int n, void* stream);

/* Fused BACKWARD of the same branch at C = 32 (one head): given d(out) it recomputes LayerNorm, Q/K/V, the selected scores, both
 *     softmaxes and P V per window from x and the 64 selection ranks the forward saved (rank_save of dhz_fused_window_attn_fwd
 *     called with rank_save ALONE), and produces dx (shortcut included) and every parameter gradient on chip.  Replaces, per block,
 *     the autograd chain of M1:839-872: dhz_reverse_residual_bwd, two backward-data GEMMs, three weight-gradient launches,
 *     dhz_ps_attn_bwd, dhz_ln_partition_bwd.
 *     wqkv_p: the forward's prepack (dhz_fused_attn_prepack); wt [4096]: dhz_fused_attn_bwd_prepack(wq, wk, wv, wo).
 *     ACCUMULATED with fp32 atomics (caller zeroes): dwq, dwk, dwv, dwo [32, 32]; dbq, dbk, dbv, dbo [32] (each may be NULL);
 *     dgamma, dbeta [32].  dbias_part [dhz_fused_attn_bwd_parts(nwin)][64][64] (written, not accumulated) iff bias != NULL:
 *     reduce with dhz_bias_table_grad.  dx [B, Hres*Wres, 32] is written. */
int dhz_fused_attn_bwd_parts(int nwin);
int dhz_fused_attn_bwd_prepack(const float* wq, const float* wk, const float* wv, const float* wo, float* wt, int C, void* stream);
int dhz_fused_window_attn_bwd(const float* x, const float* dout, const float* gamma, const float* beta, const float* wqkv_p,
                              const float* bqkv, const float* wt, const float* bias, const float* mask, const float* drop_scale,
                              const uint8_t* rank, float* dx, float* dwq, float* dwk, float* dwv, float* dbq, float* dbk,
                              float* dbv, float* dwo, float* dbo, float* dgamma, float* dbeta, float* dbias_part, int B, int Hres,
                              int Wres, int C, int shift, void* stream);

/* K3-dense  Dense window attention of the My_model.Uformer twin.  Replaces WindowAttention.forward
 *     M0:428-492:  out = softmax(scale * q k^T + bias[h] + mask[b % nW]) v   per (window, head).
 *     Same layouts as dhz_ps_attn_fwd (q,k,v [B_,64,H,d] with token stride ld; bias [H,64,64] or NULL;
 *     mask [nW,64,64] or NULL).  Nothing is saved: the backward recomputes the probabilities.
 *     dbias_part: [dhz_ps_attn_bwd_parts(B_,H), 64, 64] partials, reduced by dhz_bias_table_grad. */
int dhz_dense_attn_fwd(const float* q, const float* k, const float* v, int ld, const float* bias,
                       const float* mask, float* out, int ldo, int B_, int H, int nW, int d,
                       float scale, void* stream);
int dhz_dense_attn_bwd(const float* q, const float* k, const float* v, int ld, const float* bias,
                       const float* mask, const float* dout, int ldo, float* dq, float* dk, float* dv,
                       int ldg, float* dbias_part, int B_, int H, int nW, int d, float scale,
                       void* stream);

/* K7  relative-position bias:  bias[h,i,j] = table[rel_index(i,j), h]   (M1:408-410, win = 8).
 * table: [225, H].  bias: [H,64,64]. */
int dhz_bias_gather(const float* table, float* bias, int H, void* stream);
/* ... for every block of one model forward in ONE launch (host arrays of n <= 32 device pointers / head counts; entry i: biases[i][H_i,64,64]
 * from tables[i][225,H_i]).  Same values as n calls of dhz_bias_gather. */
int dhz_bias_gather_multi(const float* const* tables, float* const* biases, const int* heads, int n, void* stream);
/* dtable[t,h] (+)= sum_p sum_{(i,j): rel_index(i,j)=t} dbias_part[p,i,j] over parts p with p%H==h.
 * dtable: [225,H], overwritten when accumulate == 0. */
int dhz_bias_table_grad(const float* dbias_part, int parts, float* dtable, int H, int accumulate,
                        void* stream);
/* ... for n <= 32 blocks in ONE launch, ACCUMULATING into dtable[i] (host arrays; the accumulate = 1 form of dhz_bias_table_grad per entry). */
int dhz_bias_table_grad_multi(const float* const* dbias_part, const int* parts, float* const* dtable, const int* heads, int n, void* stream);

/* K9  InputProj: Conv2d(3, E, 3x3, padding 1) + LeakyReLU(slope) from the NCHW image into the token layout, M1:659-682
 *     (replaces aten::convolution + aten::leaky_relu_ + the NCHW -> token copy), E = 32 or 64.
 *     dhz_input_proj_fwd: img [B,3,H,W], w [E,3,3,3], bias [E] -> y [B, H*W, E].
 *     dhz_input_proj_bwd: dy, y (the forward output: the LeakyReLU mask is its sign) -> dw [E,3,3,3], db [E] ACCUMULATED (fp32
 *     atomics, caller zeroes).  The image needs no gradient on this path (M1:1169). */
int dhz_input_proj_fwd(const float* img, const float* w, const float* bias, float* y, int B, int H, int W, int E, float slope,
                       void* stream);
int dhz_input_proj_bwd(const float* dy, const float* y, const float* img, float* dw, float* db, int B, int H, int W, int E,
                       float slope, void* stream);
/* the same with the token tensors (y, dy) stored as `dtype` (DHZ_F32 / DHZ_BF16; image, weights and gradients fp32) */
int dhz_input_proj_fwd_dt(const float* img, const float* w, const float* bias, void* y, int B, int H, int W, int E, float slope,
                          int dtype, void* stream);
int dhz_input_proj_bwd_dt(const void* dy, const void* y, const float* img, float* dw, float* db, int B, int H, int W, int E,
                          float slope, int dtype, void* stream);

/* K8  Downsample: Conv2d(Cin, Cout, kernel 4, stride 2, padding 1) on the token layout, M1:606-622, as implicit GEMMs on the fp32
 *     matrix pipe (no im2col matrix: a tap of an output pixel is one contiguous run of Cin floats of a token).
 *     x [B, H*W, Cin] -> y [B, (H/2)*(W/2), Cout].  wp = weight.permute(0,2,3,1) as [Cout, 16*Cin] (taps-major, channels
 *     contiguous), wq = weight.permute(2,3,0,1) as [16*Cout, Cin]; H, W even; Cin, Cout multiples of 32.
 *     dhz_conv4s2_fwd  : replaces aten::convolution (MIOpen implicit GEMM)             bias [Cout] or NULL
 *     dhz_conv4s2_dgrad: dx [B, H*W, Cin] from dy [B, (H/2)*(W/2), Cout] (every element written; four parity-class GEMMs)
 *     dhz_conv4s2_wgrad: dwp [Cout, 16*Cin] += dy^T xcol, db [Cout] += column sums (ACCUMULATED, fp32 atomics; db may be NULL);
 *                        the output map sizes must be powers of two (training patch sizes 128 / 256). */
int dhz_conv4s2_fwd(const float* x, const float* wp, const float* bias, float* y, int B, int H, int W, int Cin, int Cout,
                    void* stream);
int dhz_conv4s2_dgrad(const float* dy, const float* wq, float* dx, int B, int H, int W, int Cin, int Cout, void* stream);
int dhz_conv4s2_wgrad(const float* dy, const float* x, float* dwp, float* db, int B, int H, int W, int Cin, int Cout,
                      void* stream);

/* K8 in bf16 (BASELINE config 4): the same convolution as three token-Linear GEMMs on the bf16 matrix pipe (dhz_linear_fwd_bf16 /
 *     _dgrad_bf16 / _wgrad_bf16 with wp [Cout, 16*Cin]) over an explicit tap-major patch matrix; in the token layout a tap of an
 *     output pixel is one contiguous run of Cin bf16, so both helpers are 16-byte-per-lane streaming copies.  Replaces
 *     aten::convolution / convolution_backward under autocast (MIOpen igemm / CK grouped-conv kernels).
 *     dhz_im2col_k4s2_bf16: x bf16 [B, H*W, Cin] -> col bf16 [B*(H/2)*(W/2), 16*Cin], col[.][(ky, kx, ci)] = x[2ho+ky-1, 2wo+kx-1, ci]
 *     dhz_col2im_k4s2_bf16: dx bf16 [B, H*W, Cin] = for every input pixel the fp32 sum of the 4 dcol entries that read it.
 *     H, W even; Cin a multiple of 8; 16-byte aligned buffers. */
int dhz_im2col_k4s2_bf16(const void* x, void* col, int B, int H, int W, int Cin, void* stream);
int dhz_col2im_k4s2_bf16(const void* dcol, void* dx, int B, int H, int W, int Cin, void* stream);

/* K5 fused  The whole LeFF branch of a LeWin block for C = 32, 64, 128 (hidden width 4C), forward, in one kernel:
 *     out = x + drop_scale[b] * linear2(gelu(dwconv3x3(gelu(linear1(norm2(x))))))          replaces M1:873 + M1:496-534
 *     (LayerNorm, both Linears on the fp32 matrix pipe around an LDS-resident 8x16-pixel tile + halo, both GELUs, the
 *     depthwise 3x3 convolution, bias adds, DropPath scale and the residual - the 4C-wide hidden tensors never travel
 *     to HBM except as the saves below).
 * x, out: [B, Hres*Wres, C] tokens; gamma, beta [C]; w1 [4C, C], b1 [4C]; wd [4C, 3, 3] (Conv2d groups = 4C), bd [4C];
 * w2 [C, 4C], b2 [C]; drop_scale [B] or NULL.  Hres % 8 == 0, Wres % 16 == 0.
 * Training mode - the five save pointers all non-NULL (else all NULL): xn_save [T, C] (norm2 output), stats_save [T, 2]
 * (mean, rstd), u_save [T, 4C] (linear1 output before GELU), tp_save [T, 4C] (gelu'(t), t = dwconv output + bd),
 * z_save [T, 4C] (gelu(t), the input of linear2) - what dhz_leff_dwconv_bwd and dhz_linear_wgrad consume. */
int dhz_leff_fused_fwd(const float* x, const float* gamma, const float* beta, const float* w1, const float* b1,
                       const float* wd, const float* bd, const float* w2, const float* b2, const float* drop_scale,
                       float* out, float* xn_save, float* stats_save, float* u_save, float* tp_save, float* z_save,
                       int B, int Hres, int Wres, int C, void* stream);
/* The same kernel with its two weight products (linear1, linear2) as six-term products on the bf16 matrix pipe (the arithmetic of
 * dhz_linear_fwd_split6: fp32-class results): w6 = the planes of both weights in the kernel's MFMA fragment order (dhz_leff_prepack6:
 * 24 C^2 bf16), read by every wave straight from L1 / L2; the LayerNorm output and z live in LDS as bf16 piece images.  C = 32, 64. */
int dhz_leff_fused_fwd6(const float* x, const float* gamma, const float* beta, const void* w6, const float* b1, const float* wd,
                        const float* bd, const float* b2, const float* drop_scale, float* out, float* xn_save, float* stats_save,
                        float* u_save, float* tp_save, float* z_save, int B, int Hres, int Wres, int C, void* stream);
int dhz_leff_prepack6(const float* w1, const float* w2, void* w6, int C, void* stream);

/* K2/K4/K5  forward and backward-data GEMMs of every token-major nn.Linear on the path (query/key/value/out
 *     projections ATT:420-422,454-458; LeFF linear1/linear2 M1:487-492,508,529; the 2x2/stride-2 transposed convolution
 *     of Upsample M1:633-648 in its token-Linear form), on the fp32 matrix pipe (v_mfma_f32_16x16x4_f32):
 *     dhz_linear_fwd  : y[T,N]  = x[T,K] . w[N,K]^T + bias[N]       replaces aten::addmm / F.linear   (bias may be NULL)
 *     dhz_linear_dgrad: dx[T,K] = dy[T,N] . w[N,K]                   replaces the aten::mm of Linear's backward
 *     x / y / dy / dx are token-major with row strides ldx / ldy (floats, multiples of 4; packed QKV buffers pass 3C);
 *     w is contiguous [N,K].  N % 32 == 0, K % 32 == 0, any T >= 1; all pointers 16-byte aligned. */
int dhz_linear_fwd(const float* x, int ldx, const float* w, const float* bias, float* y, int ldy, int T, int N, int K,
                   void* stream);
int dhz_linear_dgrad(const float* dy, int ldy, const float* w, float* dx, int ldx, int T, int N, int K, void* stream);

/* EXPERIMENT, off by default (dehaze_hip.ops.SPLIT_BF16 / bench.py --split-bf16; BASELINE configs[1] never takes it): the same two
 *     GEMMs with fp32 operands and results in HBM but the products on the bf16 matrix pipe, each operand split on its way into LDS
 *     into bf16 head + bf16 remainder and a.b taken as a_hi b_hi + a_hi b_lo + a_lo b_hi (~16 mantissa bits per product, fp32
 *     accumulation) - csrc/linear_split.hip.  terms = 3: as described; terms = 6: three pieces per operand (all 24 mantissa bits)
 *     and the six products down to 2^-16 - the error class of an fp32 GEMM at 6/16 of its matrix time.  N % 64 == 0, K % 64 == 0. */
int dhz_linear_fwd_split(const float* x, int ldx, const float* w, const float* bias, float* y, int ldy, int T, int N, int K,
                         int terms, void* stream);
int dhz_linear_dgrad_split(const float* dy, int ldy, const float* w, float* dx, int ldx, int T, int N, int K, int terms,
                           void* stream);
/* The product's default form of the two GEMMs above (dehaze_hip.ops.SPLIT_BF16 == 6; csrc/split6_gemm.hip): the six-term split with
 *     the WEIGHT operand pre-split - w_hi / w_mid / w_lo are the three bf16 planes of w[N,K] (same layout, truncation pieces:
 *     hi + mid + lo == w exactly), written by dhz_split3_planes (one launch over the flat parameter buffer after every dhz_adamw_step; + dhz_split3_planes_t for W^T).  The
 *     activation operand is split inside the kernel.  fp32 activations, bias, accumulation and results; replaces the same
 *     aten::addmm / aten::mm as dhz_linear_fwd / dhz_linear_dgrad.  K % 32 == 0; N % 32 == 0 (forward), N % 32 == 0 and
 *     K % 64 == 0 (backward-data: its output features are w's K columns); planes 16-byte aligned. */
int dhz_linear_fwd_split6(const float* x, int ldx, const void* w_hi, const void* w_mid, const void* w_lo, const float* bias, float* y,
                          int ldy, int T, int N, int K, void* stream);
int dhz_linear_dgrad_split6(const float* dy, int ldy, const void* w_hi, const void* w_mid, const void* w_lo, float* dx, int ldx, int T,
                            int N, int K, void* stream);
/* K4 INSIDE the GEMM (csrc/tok_epilogue.h): the out-projection / linear2 product whose epilogue is the block's residual step -
 *     out[dst(t), :] = res[dst(t), :] + scale[t / tokens_per_image] * (x[t, :] . w^T + bias)
 *   windowed = 1: row t of x is a window slot (the order dhz_ln_partition_fwd writes: M1:846-852) and dst(t) is its token-order position
 *                 after window_reverse + roll(+shift)  =  attn out-projection + M1:859-868 + `shortcut + drop_path(x)` M1:872;
 *   windowed = 0: dst(t) = t                           =  LeFF linear2 / Mlp fc2 + `x + drop_path(mlp(norm2(x)))` M1:873.
 *   res == NULL adds nothing (with bias == NULL and windowed = 0 this is the per-image factor of a backward-data product:
 *   d(ctx) = scale . (d(out) . W_o), the DropPath factor of M1:872 in the backward pass); scale == NULL means 1.
 *   The GEMM result never exists in HBM in window order: replaces dhz_linear_fwd_split6 + dhz_reverse_residual_fwd.
 *   res / out: rows of ldo floats in token order; tokens_per_image (= Hres * Wres when windowed) a multiple of 64 dividing T whenever
 *   scale != NULL or windowed; Hres, Wres multiples of 8, 0 <= shift < 8.  Same shape contract as dhz_linear_fwd_split6 otherwise.
 *   _split_res / _dgrad_split_scaled: the same epilogue on the kernels of dhz_linear_fwd_split / dhz_linear_dgrad_split (few-tile shapes). */
int dhz_linear_fwd_split6_res(const float* x, int ldx, const void* w_hi, const void* w_mid, const void* w_lo, const float* bias,
                              const float* res, const float* scale, float* out, int ldo, int T, int N, int K, int tokens_per_image,
                              int Hres, int Wres, int shift, int windowed, void* stream);
int dhz_linear_fwd_split_res(const float* x, int ldx, const float* w, const float* bias, const float* res, const float* scale, float* out,
                             int ldo, int T, int N, int K, int tokens_per_image, int Hres, int Wres, int shift, int windowed, int terms,
                             void* stream);
int dhz_linear_dgrad_split_scaled(const float* dy, int ldy, const float* w, const float* scale, float* dx, int ldx, int T, int N, int K,
                                  int tokens_per_image, int terms, void* stream);
/*   ... and on the bf16 kernels of BASELINE config 4 (dhz_linear_fwd_bf16): x, w, res, out in bf16, bias / scale fp32, fp32 accumulation;
 *   out = bf16(res + scale (x . w^T + bias)) - one rounding less than the two-launch form (the product is not rounded to bf16 first). */
int dhz_linear_fwd_bf16_res(const void* x, int ldx, const void* w, const float* bias, const void* res, const float* scale, void* out, int ldo,
                            int T, int N, int K, int tokens_per_image, int Hres, int Wres, int shift, int windowed, void* stream);
/*     hi[i] + mid[i] + lo[i] == src[i] exactly (three bf16 by truncation); n % 8 == 0, 16-byte aligned pointers. */
int dhz_split3_planes(const float* src, int64_t n, void* hi, void* mid, void* lo, void* stream);
/*     ... and the planes of the TRANSPOSES of nmat matrices inside one buffer: desc (device, int[nmat][4]) = {offset, rows R, cols C, index
 *     of the matrix's first 32 x 32 tile}; matrix m ([R][C] at src + offset) is written as [C][R] at the same offset of the planes.
 *     dx = dy . w is then dhz_linear_fwd_split6 on them (no bias): the backward-data GEMM without transposed fragment reads. */
int dhz_split3_planes_t(const float* src, void* hi, void* mid, void* lo, const int* desc, int nmat, int ntiles, void* stream);
/* bf16 (round to nearest even) copies of the TRANSPOSES of the same table of matrices: dst (bf16, as many elements as src) receives
 *     matrix m transposed at its own offset.  dhz_linear_fwd_bf16(dy, ldy, dst + off, NULL, dx, ldx, T, K, N) is then the backward-data
 *     product dx = dy . W of a Linear with weight W [N, K] (M1:487-492 under autocast) on the software-pipelined forward kernel. */
int dhz_bf16_transpose_batched(const float* src, void* dst, const int* desc, int nmat, int ntiles, void* stream);
/*     ... and the weight gradient (contract of dhz_linear_wgrad_multi + the row scale of dhz_linear_wgrad_rs: row_scale may be NULL;
 *     dw / db HOST arrays of nmat device pointers; ACCUMULATED; db exact fp32 column sums).  T % 64 == 0, nper % 64 == 0, K % 64 == 0. */
int dhz_linear_wgrad_split(const float* dy, int ldy, const float* x, int ldx, int T, int nmat, int nper, int K, float* const* dw,
                           float* const* db, const float* row_scale, int rows_per_scale, int terms, void* stream);

/* bf16 forms of the three token-Linear GEMMs (BASELINE config 4: bf16 activations and bf16 weight copies, fp32 accumulation on
 *     v_mfma_f32_16x16x32_bf16; biases and all parameter gradients stay fp32).  Same contracts as dhz_linear_fwd / _dgrad /
 *     _wgrad_multi with bf16 token tensors (x, y, dy, dx) and a bf16 copy of w [N,K]; N % 64 == 0, K % 64 == 0, wgrad T % 64 == 0;
 *     token operands 16-byte aligned, leading dimensions multiples of 8 elements.  The reference runs this path under
 *     torch.cuda.amp.autocast (TR:224). */
int dhz_linear_fwd_bf16(const void* x, int ldx, const void* w, const float* bias, void* y, int ldy, int T, int N, int K,
                        void* stream);
int dhz_linear_dgrad_bf16(const void* dy, int ldy, const void* w, void* dx, int ldx, int T, int N, int K, void* stream);
int dhz_linear_wgrad_bf16(const void* dy, int ldy, const void* x, int ldx, int T, int nmat, int nper, int K,
                          float* const* dw, float* const* db, void* stream);

/* K5b Mlp activation (token_mlp = 'ffn', My_model_1.py:442-468 - the constructor default of Uformer; options.py selects 'leff'): y = GELU(u)
 *     (exact-erf form) over n elements; backward du = dy * GELU'(u) * scale[e / elems_per_scale] (scale NULL = 1: the per-image DropPath
 *     factor of the branch folded in).  Replaces aten::gelu / aten::gelu_backward between the two token Linears. */
int dhz_gelu_fwd_dt(const void* u, void* y, int64_t n, int dtype, void* stream);
int dhz_gelu_bwd_dt(const void* dy, const void* u, void* du, int64_t n, const float* scale, int64_t elems_per_scale, int dtype, void* stream);
/* dtype-generic forms (dtype = DHZ_F32 / DHZ_BF16 storage of the token tensors, fp32 arithmetic inside) of the streaming
 * kernels around the GEMMs; argument meaning as the fp32 entry points of the same name. */
int dhz_ln_partition_fwd_dt(const void* x, const float* gamma, const float* beta, void* xw, float* stats, int B, int Hres,
                            int Wres, int C, int shift, int partition, int dtype, void* stream);
int dhz_ln_partition_bwd_dt(const void* dxw, const void* x, const float* gamma, const float* stats, const void* dres, void* dx,
                            float* dgamma, float* dbeta, int B, int Hres, int Wres, int C, int shift, int partition, int dtype,
                            void* stream);
/* dhz_ln_partition_bwd_dt with the gradient LAYOUTS of a whole LeWin block's backward pass (M1:839-873 under autograd): the LeFF branch's
 * LayerNorm backward (partition = 0) writes dx in the window order of the attention branch (dx_windowed = 1, dx_shift = its shift) -
 * exactly the d(out) operand the out-projection's backward-data / weight-gradient products read, so no dhz_reverse_residual_bwd pass
 * runs - and the attention branch's LayerNorm backward (partition = 1) reads its residual gradient dres in that same order
 * (dres_windowed = 1).  dx_windowed needs Hres, Wres multiples of 8 and dx != dres. */
int dhz_ln_partition_bwd_lay(const void* dxw, const void* x, const float* gamma, const float* stats, const void* dres, void* dx,
                             float* dgamma, float* dbeta, int B, int Hres, int Wres, int C, int shift, int partition, int dres_windowed,
                             int dx_windowed, int dx_shift, int dtype, void* stream);
/* ... with a SECOND output: dx2 (may be NULL) receives scale2[image] * dx in the window order of dx_shift (scale2 NULL = 1) - the scaled,
 * window-ordered d(out) operand of the out-projection's backward products for storage types whose GEMMs take no row factor (bf16). */
int dhz_ln_partition_bwd_lay2(const void* dxw, const void* x, const float* gamma, const float* stats, const void* dres, void* dx,
                              float* dgamma, float* dbeta, int B, int Hres, int Wres, int C, int shift, int partition, int dres_windowed,
                              int dx_windowed, int dx_shift, void* dx2, const float* scale2, int dtype, void* stream);
int dhz_reverse_residual_fwd_dt(const void* yw, const void* shortcut, const float* scale, void* out, int B, int Hres, int Wres,
                                int C, int shift, int partition, int dtype, void* stream);
int dhz_reverse_residual_bwd_dt(const void* dout, const float* scale, void* dyw, int B, int Hres, int Wres, int C, int shift,
                                int partition, int dtype, void* stream);
int dhz_leff_dwconv_fwd_dt(const void* u, const float* w, const float* b, void* t, void* z, int B, int Hres, int Wres, int Ch,
                           int dtype, void* stream);
int dhz_leff_dwconv_bwd_dt(const void* dz, const void* u, const void* t, const float* w, void* du, float* dw, float* db, int B,
                           int Hres, int Wres, int Ch, int dtype, void* stream);
/*      the same with dz multiplied by dz_scale[b] per image on the way in (the DropPath scale of M1:873 folded into this kernel;
 *      dz_scale NULL = 1) */
int dhz_leff_dwconv_bwd_scaled_dt(const void* dz, const void* u, const void* t, const float* w, void* du, float* dw, float* db,
                                  const float* dz_scale, int B, int Hres, int Wres, int Ch, int dtype, void* stream);
int dhz_ps_attn_fwd_dt(const void* q, const void* k, const void* v, int ld, const uint8_t* idx, const float* bias,
                       const float* mask, void* out, int ldo, uint8_t* rank, int B_, int H, int nW, int d, int dtype,
                       void* stream);
int dhz_ps_attn_bwd_dt(const void* q, const void* k, const void* v, int ld, const float* bias, const float* mask,
                       const uint8_t* rank, const void* dout, int ldo, void* dq, void* dk, void* dv, int ldg,
                       float* dbias_part, int B_, int H, int nW, int d, int dtype, void* stream);

/* K2/K4/K5 (backward)  weight + bias gradient of every token-major nn.Linear on the path
 *     (query/key/value/out projections ATT:420-422,454-458; LeFF linear1/linear2 M1:487-492):
 *     dw[N,K] += dy^T[N,T] . x[T,K]        db[N] += sum_t dy[t,:]        (ACCUMULATED: caller zeroes,
 *     which lets the gradients land directly in the optimizer's flat gradient buffer).
 *     dy: [T,N] with row stride ldy, x: [T,K] with row stride ldx; T % 32 == 0, N % 32 == 0, K % 32 == 0.
 *     db may be NULL.  Summation order over T is not deterministic (fp32 atomics). */
int dhz_linear_wgrad(const float* dy, int ldy, const float* x, int ldx, int T, int N, int K,
                     float* dw, float* db, void* stream);
/*      Same contraction for nmat (1..4) parameters that share the input x - the Q / K / V projections of
 *      AttentionLayer.forward (ATT:385-461): columns [i*nper, (i+1)*nper) of dy belong to dw[i] / db[i] (HOST arrays of
 *      nmat device pointers; db may be NULL, or all of its entries NULL).  One launch reads x once instead of nmat times. */
/*      dhz_linear_wgrad with row t of dy multiplied by row_scale[t / rows_per_scale] on the way in (per-image DropPath scale of the
 *      branch output; rows_per_scale a multiple of 32 that divides T; row_scale NULL = plain dhz_linear_wgrad). */
int dhz_linear_wgrad_rs(const float* dy, int ldy, const float* x, int ldx, int T, int N, int K, float* dw, float* db,
                        const float* row_scale, int rows_per_scale, void* stream);
int dhz_linear_wgrad_multi(const float* dy, int ldy, const float* x, int ldx, int T, int nmat, int nper, int K,
                           float* const* dw, float* const* db, void* stream);

/* K11  3x3 / stride 1 / pad 1 convolution of the VGG19 feature stack (My_CR.py:56-86) as Winograd F(2x2,3x3) on the
 *      fp32 matrix pipe.  Tensors are channel-blocked NCHW8c: x[b][c/8][h][w][c%8] (dhz_layout_blocked8 converts).
 *      dhz_winograd_prepack: weight [Kout,Cin,3,3] -> transform-domain filters upack (16*Kout*Cin floats);
 *        transposed_rot != 0 builds the backward-data filters from the forward weight [Cin,Kout,3,3] (the roles of
 *        the two channel counts swap, taps rotate by 180 degrees).  The VGG weights are frozen: prepack once.
 *      dhz_winograd_conv3x3: forward  y = conv(x, W) [+ bias] [ReLU]                (out_mask = out_addend = NULL)
 *                            backward y = out_mask > 0 ? conv(x, W') + out_addend : 0   (bias = NULL, relu = 0)
 *        where W' are the backward-data filters, out_mask (shape of y, may be NULL) is the saved post-ReLU activation
 *        at the OUTPUT positions - the ReLU of the layer below, fused into the store - and out_addend (shape of y, may
 *        be NULL) the gradient reaching that activation from a loss tap.
 *        H % 16 == 0 and W % 16 == 0, or H == W == 8 (the conv5_1 geometry of 128 x 128 patches: four images share one
 *        16 x 16 block of the kernel); C % 8 == 0, K % 32 == 0.
 *      dhz_maxpool2x2_blocked_fwd / _bwd: the 2x2/stride-2 max pooling between VGG stages in the same layout
 *        (N = B*C/8 planes of [H][W][8]); the backward routes gy to the first maximum of each window of the saved
 *        post-ReLU map `act` and applies that map's ReLU (act > 0) in the same pass. */
int dhz_winograd_prepack(const float* weight, float* upack, int Kout, int Cin, int transposed_rot, void* stream);
int dhz_winograd_conv3x3(const float* x, const float* upack, const float* bias, int relu, const float* out_mask,
                         const float* out_addend, float* y, int B, int H, int W, int C, int K, void* stream);
/* K11b The same convolution as Winograd F(4x4,3x3) (round 5; csrc/winograd43_conv.hip): 36 transform-domain products per 16 outputs
 *      instead of 16 per 4.  Same tensors, same epilogues, same argument meaning as the two entry points above; upack holds
 *      36*Kout*Cin floats.  H % 16 == 0, W % 16 == 0 (maps of 16 x 16 and more: the 8 x 8 layer stays on dhz_winograd_conv3x3),
 *      C % 16 == 0, K % 32 == 0 (a workgroup owns 32 output channels; other K: DHZ_EINVAL).  Interpolation points (0, +-3/4, +-3/2, inf),
 *      filters transformed in double.  Input channels are accumulated in the transform domain in chains of 256: C <= 256 is one launch;
 *      C > 256 runs as chained launches over 256-channel slices in which y carries the partial sums (READ and written by every launch
 *      but the first; bias / relu / out_mask / out_addend applied by the last) - ~2x the rounding error of F(2x2,3x3), inside the same
 *      test tolerances. */
int dhz_winograd43_prepack(const float* weight, float* upack, int Kout, int Cin, int transposed_rot, void* stream);
int dhz_winograd43_conv3x3(const float* x, const float* upack, const float* bias, int relu, const float* out_mask,
                           const float* out_addend, float* y, int B, int H, int W, int C, int K, void* stream);
/* ... + bias + ReLU + 2 x 2 / stride-2 max pooling in the same launch (replaces F.relu(conv2d) followed by nn.MaxPool2d(2, 2) of the VGG19
 * slices, My_CR.py:60-86, where the un-pooled map is not a tap): ypool [B][K/8][H/2][W/2][8].  scratch: full-resolution map
 * [B][K/8][H][W][8], needed only for C > 256 (partial sums of the accumulation chains), may be NULL otherwise. */
int dhz_winograd43_conv3x3_pool(const float* x, const float* upack, const float* bias, float* ypool, float* scratch, int B, int H, int W,
                                int C, int K, void* stream);
int dhz_maxpool2x2_blocked_fwd(const float* x, float* y, int N, int H, int W, void* stream);
int dhz_maxpool2x2_blocked_bwd(const float* gy, const float* act, float* gx, int N, int H, int W, void* stream);
int dhz_layout_blocked8(const float* src, float* dst, int B, int C, int HW, int to_blocked, const float* bias, int relu,
                        void* stream);   /* towards the blocked layout optionally dst = max(src + bias[c], 0) */

/* K9b  output projection (My_model_1.py:696-723): Conv2d(C -> 3, 3x3, pad 1) from tokens x[B, H*W, C] to an NCHW image
 *      y[B, 3, H, W] (+ bias[3], may be NULL); backward-data dx[B, H*W, C] from dy[B, 3, H, W]; weight / bias gradient
 *      dw[3, C, 3, 3], db[3] (ACCUMULATED; db may be NULL).  w is the layer's own [3, C, 3, 3] tensor.  C in {64, 128}. */
int dhz_thin_conv3x3_fwd(const float* x, const float* w, const float* bias, float* y, int B, int H, int W, int C, void* stream);
int dhz_thin_conv3x3_dgrad(const float* dy, const float* w, float* dx, int B, int H, int W, int C, void* stream);
int dhz_thin_conv3x3_wgrad(const float* dy, const float* x, float* dw, float* db, int B, int H, int W, int C, void* stream);
/* the same with the token tensors (x, dx) stored as `dtype` (DHZ_F32 / DHZ_BF16; images, weights and gradients fp32) */
int dhz_thin_conv3x3_fwd_dt(const void* x, const float* w, const float* bias, float* y, int B, int H, int W, int C, int dtype,
                            void* stream);
int dhz_thin_conv3x3_dgrad_dt(const float* dy, const float* w, void* dx, int B, int H, int W, int C, int dtype, void* stream);
int dhz_thin_conv3x3_wgrad_dt(const float* dy, const void* x, float* dw, float* db, int B, int H, int W, int C, int dtype,
                              void* stream);
/*      backward-data of a 3 -> C convolution (the first VGG19 layer, My_CR.py:65) from a channel-blocked gradient
 *      gb[B, C/8, H, W, 8] and that layer's weight w[C, 3, 3, 3]: dx[B, 3, H, W].  C = 64. */
/*      the same layer forward: y[B, K/8, H, W, 8] (channel-blocked) = max(conv(x[B, 3, H, W], w[K, 3, 3, 3]) + bias, 0 if relu).
 *      K = 64. */
int dhz_conv3x3_in3_blocked(const float* x, const float* w, const float* bias, float* y, int B, int H, int W, int K, int relu,
                            void* stream);
int dhz_thin_conv3x3_dgrad_blocked(const float* gb, const float* w, float* dx, int B, int H, int W, int C, void* stream);

/* K11b the two L1 distances of one ContrastLoss feature tap (My_CR.py:108-112): sums[0] += sum|a-p|, sums[1] += sum|a-n|
 *      (n may be NULL: ablation, My_CR.py:114-119); backward da = (g[0] sign(a-p) + g[1] sign(a-n)) / count with g the
 *      device-resident gradients of the two MEANS.  count % 4 == 0; any (common) element order. */
int dhz_l1_pair_fwd(const float* a, const float* p, const float* n, float* sums, int64_t count, void* stream);
int dhz_l1_pair_bwd(const float* a, const float* p, const float* n, const float* g, float* da, int64_t count, void* stream);

/* K10b the same feature stack for BASELINE config 4 (bf16 feature maps in NHWC / token layout [N, H, W, C]; what torch.autocast
 *      makes of My_CR.py:65-72's convolutions), csrc/vgg_bf16.hip:
 *      dhz_vgg_prepack_bf16: filter w[K, C, 3, 3] fp32 -> bf16 [K][9 C] tap-major (transpose = 0) or the backward-data filter
 *        [C][9 K] with the taps rotated by 180 degrees (transpose = 1).
 *      dhz_vgg_conv3x3_bf16: y[N, H, W, Cout] = conv3x3(x[N, H, W, Cin], pad 1) as an implicit GEMM on the bf16 matrix pipe
 *        (fp32 accumulation); forward y = max(. + bias, 0 if relu); backward-data (bias NULL, relu 0, wp the transposed pack)
 *        y = act > 0 ? . + addend : 0 with act (may be NULL) the saved post-ReLU map at the OUTPUT positions and addend (may be
 *        NULL; needs act) the gradient reaching that map from its loss tap.  Cin a power of two >= 64, Cout % 64 == 0.
 *      dhz_maxpool2x2_nhwc_bf16_fwd / _bwd: 2x2 / stride-2 max pooling; the backward routes gy to the first maximum of each
 *        window of the saved post-ReLU map `act` and applies act > 0.  C % 8 == 0.
 *      dhz_l1_pair_fwd_bf16 / _bwd_bf16: K11b on bf16 maps (fp32 sums; da in bf16).  count % 8 == 0. */
int dhz_vgg_prepack_bf16(const float* w, void* out, int K, int C, int transpose, void* stream);
int dhz_vgg_conv3x3_bf16(const void* x, const void* wp, const float* bias, int relu, const void* act, const void* addend, void* y,
                         int N, int H, int W, int Cin, int Cout, void* stream);
int dhz_maxpool2x2_nhwc_bf16_fwd(const void* x, void* y, int N, int H, int W, int C, void* stream);
int dhz_maxpool2x2_nhwc_bf16_bwd(const void* gy, const void* act, void* gx, int N, int H, int W, int C, void* stream);
int dhz_l1_pair_fwd_bf16(const void* a, const void* p, const void* n, float* sums, int64_t count, void* stream);
int dhz_l1_pair_bwd_bf16(const void* a, const void* p, const void* n, const float* g, void* da, int64_t count, void* stream);

/* K11c the scalar side of ContrastLoss.forward (My_CR.py:104-123) over k feature taps: d[i] = (ap_i, an_i) = sums[i] * inv_cnt[i];
 *      out = (sum_i w_i ap_i / (an_i + 1e-7)  [ablation: sum_i w_i ap_i],  sum_i ap_i,  sum_i an_i).  Backward: g[i] = gradient of
 *      (out . (g_loss, g_ap, g_an)) w.r.t. (ap_i, an_i); a NULL upstream gradient counts as zero.  Replaces the [k]-vector
 *      mul / div / add / sum chain of torch ops.  1 <= k <= 64; all pointers device memory. */
int dhz_contrast_combine_fwd(const float* sums, const float* inv_cnt, const float* w, int k, int ablation, float* d, float* out,
                             void* stream);
int dhz_contrast_combine_bwd(const float* d, const float* w, int k, int ablation, const float* g_loss, const float* g_ap,
                             const float* g_an, float* g, void* stream);

/* F2  training feed (dataset.py:17-77): batch item t = table[t] = (patch id, r, c, k) -> ps x ps crop at (r, c) of the
 *     uint8 [N,Hs,Ws,3] RGB patch pair in HBM, augmentation k of utils/dataset_utils.py:6-40 (0 id, 1-3 rot90 k with
 *     dims=[-1,-2], 4-7 the same followed by flip(-2)), float32 [n,3,ps,ps] = value / 255. */
int dhz_crop_augment_pair(const uint8_t* gt, const uint8_t* hazy, const int* table, float* out_gt, float* out_hazy,
                          int n, int Hs, int Ws, int ps, void* stream);

/* K6  shift mask builder: mask[nW,64,64] in {0,-100}  (M1:803-836), Hres x Wres map, win 8. */
int dhz_shift_mask(float* mask, int Hres, int Wres, int shift, void* stream);

/* ---------------------------------------------------------------------------------------------
 * K1  LayerNorm + cyclic shift + window partition.  Replaces norm1 M1:839, roll M1:846,
 *     window_partition M1:550-574.   x: [B, Hres*Wres, C] -> xw: [B*nW, 64, C].
 *     stats: [B*Hres*Wres, 2] (mean, rstd) saved for backward.  eps = 1e-5.  C % 4 == 0, C <= 1024.
 *     partition == 0: plain LayerNorm, tokens stay in place (norm2, M1:873).
 */
int dhz_ln_partition_fwd(const float* x, const float* gamma, const float* beta, float* xw,
                         float* stats, int B, int Hres, int Wres, int C, int shift, int partition,
                         void* stream);
/* dxw: [B*nW,64,C] -> dx: [B,HW,C] = LN-backward(dxw) (+ dres when dres != NULL: the gradient that reaches x
 * through the residual shortcut, M1:872 - dx may alias dres);  dgamma,dbeta [C] are ACCUMULATED (caller zeroes). */
int dhz_ln_partition_bwd(const float* dxw, const float* x, const float* gamma, const float* stats,
                         const float* dres, float* dx, float* dgamma, float* dbeta, int B, int Hres,
                         int Wres, int C, int shift, int partition, void* stream);

/* K4 (tail)  window reverse + un-shift + residual with per-sample DropPath scale.
 *     Replaces window_reverse M1:577-601, roll M1:866, shortcut + drop_path(x) M1:872.
 *     out[b,p,:] = shortcut[b,p,:] + scale[b] * yw[window-position(b,p),:]   (scale == NULL -> 1).
 *     partition == 0: yw is already in token order (the LeFF residual x + drop_path(mlp(..)), M1:873).
 */
int dhz_reverse_residual_fwd(const float* yw, const float* shortcut, const float* scale, float* out,
                             int B, int Hres, int Wres, int C, int shift, int partition, void* stream);
/* dyw[window-position] = scale[b] * dout[b,p,:]  (the shortcut gradient is dout itself). */
int dhz_reverse_residual_bwd(const float* dout, const float* scale, float* dyw, int B, int Hres,
                             int Wres, int C, int shift, int partition, void* stream);

/* ---------------------------------------------------------------------------------------------
 * K5 (middle)  LeFF depthwise stage in token (NHWC) layout.  Replaces the NHWC<->NCHW rearranges,
 *     the GELU after linear1, the depthwise 3x3 conv and its GELU  (M1:488,514-520).
 *     u: [B, Hres*Wres, Ch] = linear1 output BEFORE GELU.  w: [Ch,1,3,3] (PyTorch layout), b: [Ch].
 *     z = gelu(dwconv3x3(gelu(u)) + b).   When t != NULL the kernel also stores t = gelu'(pre-activation),
 *     the only thing the backward needs from the second GELU (same bytes as saving the pre-activation,
 *     one exponential fewer per element in the backward).  GELU is the exact-erf form evaluated with the
 *     Abramowitz-Stegun 7.1.26 rational approximation of erf (|error| <= 1.5e-7, i.e. fp32 rounding level).
 */
int dhz_leff_dwconv_fwd(const float* u, const float* w, const float* b, float* t, float* z, int B,
                        int Hres, int Wres, int Ch, void* stream);
/* du (overwritten); dw [Ch*9], db [Ch] ACCUMULATED (caller zeroes). */
int dhz_leff_dwconv_bwd(const float* dz, const float* u, const float* t, const float* w, float* du,
                        float* dw, float* db, int B, int Hres, int Wres, int Ch, void* stream);

/* ---------------------------------------------------------------------------------------------
 * K10  Charbonnier loss on clamp(x,0,1)  (TR:230 + losses.py:48-52).
 *      loss_sum (1 float, ACCUMULATED; caller zeroes) = sum sqrt((clamp(x)-y)^2 + eps^2).
 *      The mean is loss_sum / n.  Backward: dx = gscale * d/sqrt(d^2+eps^2) inside (0,1), 0 outside
 *      (clamp has zero gradient outside [0,1]; boundary convention of torch.clamp: pass-through at
 *      exactly 0 or 1).  `clampd` (optional, may be NULL) receives clamp(x,0,1).
 *      clamp01 == 0: plain CharbonnierLoss.forward(x, y) without the clamp.
 */
int dhz_charbonnier_fwd(const float* x, const float* y, float* clampd, float* loss_sum, int64_t n,
                        float eps, int clamp01, void* stream);
/* gclamp (optional): gradient arriving at clamp(x,0,1) from other consumers (the contrastive loss);
 * dx = inside(x) * (gscale[0]*inv_n * d/sqrt(d^2+eps^2) + gclamp). */
int dhz_charbonnier_bwd(const float* x, const float* y, const float* gscale, const float* gclamp,
                        float* dx, int64_t n, float eps, float inv_n, int clamp01, void* stream);

/* K12  AdamW step over one flat fp32 buffer (torch.optim.AdamW semantics, TR:90-92):
 *      p *= 1 - lr*wd;  m = b1*m + (1-b1)*g;  v = b2*v + (1-b2)*g*g;
 *      p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps).   `step` is t (>= 1).
 *      grad_scale multiplies g first (1/world_size after a sum all-reduce). */
int dhz_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                   float beta2, float eps, float wd, int step, float grad_scale, void* stream);
/*      the same, also writing the updated parameters as bf16 to p16[n] (may be NULL): the weight copy the bf16 GEMMs of BASELINE
 *      config 4 read, produced in the optimizer's own pass instead of a separate cast of the whole buffer */
int dhz_adamw_step_shadow(float* p, const float* g, float* m, float* v, void* p16, int64_t n, float lr, float beta1,
                          float beta2, float eps, float wd, int step, float grad_scale, void* stream);

/* C1   The gradient exchange of the data-parallel path (one process per GPU, batch-axis sharding): thin wrappers over RCCL for a
 *      host that binds only this library - replaces nn.DataParallel's gradient reduction (My_train.py:97).  RCCL is resolved at
 *      run time (dlopen; a copy the process already holds, e.g. PyTorch's, is shared), so single-GPU users never load it.
 *      dhz_comm_unique_id : rank 0 fills id128 (128 bytes) and hands it to the other ranks out of band (a file, MPI, a socket)
 *      dhz_comm_init      : every rank, after hipSetDevice(its GPU); *comm receives the communicator
 *      dhz_comm_allreduce_sum_f32 : in-place SUM all-reduce of buf[n] (a bucket of the flat gradient), enqueued on `stream`; the
 *                           1/world factor belongs to dhz_adamw_step's grad_scale
 *      dhz_comm_destroy   : releases the communicator.
 *      The Python host of this repository reaches the same RCCL through torch.distributed("nccl") (dehaze_hip/train.py::GradReducer). */
int dhz_comm_unique_id(void* id128);
int dhz_comm_init(void** comm, int rank, int nranks, const void* id128);
int dhz_comm_allreduce_sum_f32(void* comm, float* buf, int64_t n, void* stream);
int dhz_comm_destroy(void* comm);

#ifdef __cplusplus
}
#endif
#endif /* DEHAZE_HIP_H */
