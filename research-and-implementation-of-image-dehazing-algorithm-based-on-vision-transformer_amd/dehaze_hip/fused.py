"""Fused attention branch of a LeWin block: ONE forward kernel (dhz_fused_window_attn_fwd) and a hand-sequenced
backward over the existing kernels, wrapped as a single autograd node.

forward  : out = x + drop_scale * OutProj(ProbAttn(QKV(partition(roll(LN(x))))))           (M1:839-872)
backward : reverse_residual_bwd -> out-proj dgrad (library GEMM) + wgrad (dhz_linear_wgrad, in place)
           -> dhz_ps_attn_bwd (+ bias table gradient) -> QKV dgrad + wgrad -> dhz_ln_partition_bwd with the
           shortcut gradient folded in (dx = dout + dLN) - no autograd-side accumulation kernels at all.
"""
import torch
from torch.autograd import Function

from . import _lib, ops
from .ops import NTOK, _p, _require_gpu, _stream

SUPPORTED_C = (32, 64, 128)
ENABLED = True      # set False to force the unfused chain (tests compare the two)
ATTN_FUSED_C128_MAX_HW = 1024   # C = 128 takes the fused attention forward up to this map size (32 x 32), the chain above
# Fused LeFF kernels (csrc/leff_fused.hip).  Measured on MI355X (tools/bench_leff.py, bs 32): the fused forward wins at C = 32 / 64
# (inference 0.76-0.85x of the chain, training 0.87-0.95x) and loses at C = 128; a fused backward-data kernel (round 2, removed
# in round 3) was 1.5-2x slower than the kernel chain in fp32 - fp32-input MFMA and fp32 VALU instructions share the SIMD's
# issue (profiles/r02_coexec_micro.txt), so fusing the GELU / depthwise VALU work into the GEMM kernels ADDS its time to the
# matrix time instead of hiding it behind HBM traffic as the stand-alone streaming kernels do.
# Fused attention BACKWARD (csrc/fused_attn_bwd.hip, C = 32): the forward then saves only the selection ranks.  A/B against the
# backward kernel chain: tools/bench_fused.py, profiles/r03_fused_attn_bwd_ab.txt.
ATTN_FUSED_BWD_C = (32,)
LEFF_FUSED = True           # False forces the kernel chain everywhere
LEFF_FUSED_C = (32, 64)     # widths that take the fused forward
# ... C = 64 only below this many tokens when the chain's GEMMs run in the six-term form (tools/bench_leff.py, bs 32, forward with the
# training saves: 64 x 64 maps 204 us fused / 218 chain, 128 x 128 maps 794 / 765; forward + backward 2058 / 2010)
LEFF_FUSED_C64_MAX_T = 262144


def _wgrad(dy, off, x, w, b, row_scale=None):
    """dW/db of one Linear from dy[:, off:off+N] and x.  In place into .grad when the parameter is a leaf and
    the split-T kernel is the better choice; returns (dw, db) to hand to autograd, or (None, None)."""
    T, K = x.shape
    N = w.shape[0]
    if not w.requires_grad and (b is None or not b.requires_grad):
        return None, None                                   # frozen Linear: nothing to compute
    q = 64 if dy.dtype == ops.BF16 else 16                  # fp32: 16-wide tiles for the embed_dim = 16 model (csrc/linear_wgrad.hip)
    mine = T % 32 == 0 and N % q == 0 and K % q == 0        # (the kernel's shape contract; always true on this model)
    if mine and w.is_leaf and w.requires_grad and (b is None or (b.is_leaf and b.requires_grad)):
        ops._accumulate_param_grads(dy, off, x, [(w, b)], row_scale)
        return None, None
    if mine:
        dw = torch.zeros_like(w, memory_format=torch.contiguous_format)
        db = torch.zeros_like(b) if b is not None else None
        ops.wgrad_into(dy, off, x, N, dw, db, row_scale)
        return dw, db
    raise RuntimeError(f"dehaze_hip: Linear weight gradient for T={T}, N={N}, K={K}: the HIP kernel needs multiples of 16 "
                       "(there is deliberately no library fallback)")


def _wgrad_qkv(dqkv, xn, C, pairs):
    """The three projections' gradients from the packed dqkv [T,3C]: one launch (x read once) when all of them can be
    accumulated in place, else one _wgrad each.  Returns the six autograd slots (g_wq, g_bq, g_wk, g_bk, g_wv, g_bv)."""
    T, K = xn.shape
    q = 64 if dqkv.dtype == ops.BF16 else 16
    if T % 32 == 0 and C % q == 0 and K % q == 0 and all(w.is_leaf and w.requires_grad and (b is None or (b.is_leaf and b.requires_grad))
                                                           for w, b in pairs):
        ops._accumulate_param_grads(dqkv, 0, xn, pairs)
        return (None,) * 6
    out = ()
    for i, (w, b) in enumerate(pairs):
        out += _wgrad(dqkv, i * C, xn, w, b)
    return out


def _grad_buf(p):
    """Zero-initialised, contiguous .grad of a leaf parameter (the optimizer's flat-buffer view when FlatAdamW is in
    use), or None when in-place accumulation is not possible."""
    if not (p.is_leaf and p.requires_grad):
        return None
    if p.grad is None:
        p.grad = torch.zeros_like(p, memory_format=torch.contiguous_format)
    return p.grad if p.grad.is_contiguous() else None


def _ready(*params):
    if ops.GRAD_READY is not None:
        for p in params:
            ops.GRAD_READY(p)


def _ln_backward(dxn, x, gamma_p, beta_p, gamma, stats, dres, B, Hres, Wres, C, shift, partition):
    """LayerNorm backward with the shortcut gradient folded in; d(gamma), d(beta) accumulated in place when the
    parameters are leaves (returns (dx, dgamma, dbeta) with None for gradients already accumulated)."""
    dx = torch.empty_like(x)
    gg, gb = _grad_buf(gamma_p), _grad_buf(beta_p)
    if gg is not None and gb is not None:
        _lib.call("dhz_ln_partition_bwd_dt", _p(dxn), _p(x), _p(gamma), _p(stats), _p(dres), _p(dx), _p(gg), _p(gb),
                  B, Hres, Wres, C, shift, partition, ops._dt(x), _stream())
        _ready(gamma_p, beta_p)
        return dx, None, None
    dgb = torch.zeros((2, C), device=x.device, dtype=torch.float32)
    _lib.call("dhz_ln_partition_bwd_dt", _p(dxn), _p(x), _p(gamma), _p(stats), _p(dres), _p(dx), dgb[0].data_ptr(),
              dgb[1].data_ptr(), B, Hres, Wres, C, shift, partition, ops._dt(x), _stream())
    return dx, dgb[0], dgb[1]


def _table_backward(dpart, parts, table_p, H, dev):
    gt = _grad_buf(table_p)
    if gt is not None:
        _lib.call("dhz_bias_table_grad", _p(dpart), parts, _p(gt), H, 1, _stream())
        _ready(table_p)
        return None
    dtable = torch.empty((225, H), device=dev, dtype=torch.float32)
    _lib.call("dhz_bias_table_grad", _p(dpart), parts, _p(dtable), H, 0, _stream())
    return dtable


class _FusedAttnBranch(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, wq, bq, wk, bk, wv, bv, wo, bo, table, idx, mask, dscale, Hres, Wres, shift, H, grad_mode):
        _require_gpu(x, gamma, beta, wq, wo, table, idx, mask, dscale)
        x = x.contiguous()
        B, L, C = x.shape
        assert L == Hres * Wres and C == 32 * H and C in SUPPORTED_C
        dev = x.device
        T = B * L
        f32 = dict(device=dev, dtype=torch.float32)
        wqkv_p = torch.empty(3 * C * C, **f32)
        wo_p = torch.empty(C * C, **f32)
        _lib.call("dhz_fused_attn_prepack", _p(wq), _p(wk), _p(wv), _p(wo), _p(wqkv_p), _p(wo_p), C, _stream())
        bqkv = ops.cat_rows([bq.detach(), bk.detach(), bv.detach()])
        bias = None
        if table is not None:
            bias = torch.empty((H, NTOK, NTOK), **f32)
            _lib.call("dhz_bias_gather", _p(table.contiguous()), _p(bias), H, _stream())
        out = torch.empty_like(x)
        train = grad_mode and any(ctx.needs_input_grad)
        fused_bwd = train and C in ATTN_FUSED_BWD_C and x.dtype == torch.float32
        xn = qkv = cx = stats = rank = None
        if train:
            if not fused_bwd:
                xn = torch.empty((T, C), **f32)
                qkv = torch.empty((T, 3 * C), **f32)
                cx = torch.empty((T, C), **f32)
                stats = torch.empty((T, 2), **f32)
            rank = torch.empty(((T // NTOK) * H * NTOK,), device=dev, dtype=torch.uint8)
        timing = ops.KERNEL_TIMING.get("dhz_fused_window_attn_fwd") if ops.KERNEL_TIMING is not None else None
        if timing is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        _lib.call("dhz_fused_window_attn_fwd", _p(x), _p(gamma), _p(beta), _p(wqkv_p), _p(bqkv), _p(wo_p), _p(bo), _p(idx),
                  _p(bias), _p(mask), _p(dscale), _p(out), _p(xn), _p(qkv), _p(cx), _p(stats), _p(rank), B, Hres, Wres, C,
                  shift, _stream())
        if timing is not None:
            e1.record()
            timing.append((e0, e1, T // NTOK, C))
        ctx.fused_bwd = fused_bwd
        if fused_bwd:
            wt = torch.empty(4096, **f32)
            _lib.call("dhz_fused_attn_bwd_prepack", _p(wq), _p(wk), _p(wv), _p(wo), _p(wt), C, _stream())
            ctx.save_for_backward(x, gamma, beta, rank, bias, mask, dscale, wqkv_p, bqkv, wt)
            ctx.params = (wq, bq, wk, bk, wv, bv, wo, bo, gamma, beta, table)
            ctx.geom = (B, Hres, Wres, C, shift, H)
        elif train:
            ctx.save_for_backward(x, gamma, stats, xn, qkv, cx, rank, bias, mask, dscale, wq, wk, wv, wo)
            ctx.params = (wq, bq, wk, bk, wv, bv, wo, bo, gamma, beta, table)
            ctx.geom = (B, Hres, Wres, C, shift, H)
        return out

    @staticmethod
    def _backward_fused(ctx, dout):
        """One kernel from d(out) to dx and every parameter gradient (csrc/fused_attn_bwd.hip)."""
        x, gamma, beta, rank, bias, mask, dscale, wqkv_p, bqkv, wt = ctx.saved_tensors
        wq, bq, wk, bk, wv, bv, wo, bo, gamma_p, beta_p, table_p = ctx.params
        B, Hres, Wres, C, shift, H = ctx.geom
        dev = x.device
        f32 = dict(device=dev, dtype=torch.float32)
        plist = (wq, wk, wv, bq, bk, bv, wo, bo, gamma_p, beta_p)
        bufs = [_grad_buf(p) if p is not None else None for p in plist]
        inplace = all(b is not None or p is None for b, p in zip(bufs, plist))
        if not inplace:
            bufs = [torch.zeros_like(p, memory_format=torch.contiguous_format) if p is not None else None for p in plist]
        gwq, gwk, gwv, gbq, gbk, gbv, gwo, gbo, gg, gb = bufs
        nwin = B * (Hres // 8) * (Wres // 8)
        parts = _lib.load().dhz_fused_attn_bwd_parts(nwin)
        dpart = torch.empty((parts, NTOK, NTOK), **f32) if bias is not None else None
        dx = torch.empty_like(x)
        _lib.call("dhz_fused_window_attn_bwd", _p(x), _p(dout.contiguous()), _p(gamma), _p(beta), _p(wqkv_p), _p(bqkv), _p(wt),
                  _p(bias), _p(mask), _p(dscale), _p(rank), _p(dx), _p(gwq), _p(gwk), _p(gwv), _p(gbq), _p(gbk), _p(gbv), _p(gwo),
                  _p(gbo), _p(gg), _p(gb), _p(dpart), B, Hres, Wres, C, shift, _stream())
        dtable = _table_backward(dpart, parts, table_p, H, dev) if bias is not None else None
        if inplace:
            _ready(*[p for p in plist if p is not None])
            return (dx, None, None, None, None, None, None, None, None, None, None, dtable,
                    None, None, None, None, None, None, None, None)
        return (dx, gg, gb, gwq, gbq, gwk, gbk, gwv, gbv, gwo, gbo, dtable, None, None, None, None, None, None, None, None)

    @staticmethod
    def backward(ctx, dout):
        if getattr(ctx, "fused_bwd", False):
            return _FusedAttnBranch._backward_fused(ctx, dout)
        x, gamma, stats, xn, qkv, cx, rank, bias, mask, dscale, wq_, wk_, wv_, wo_ = ctx.saved_tensors
        wq, bq, wk, bk, wv, bv, wo, bo, gamma_p, beta_p, table_p = ctx.params
        B, Hres, Wres, C, shift, H = ctx.geom
        dout = dout.contiguous()
        dev = x.device
        T = B * Hres * Wres
        B_ = T // NTOK
        f32 = dict(device=dev, dtype=torch.float32)
        # (1) gradient of the window-ordered out-projection output
        daw = torch.empty((T, C), **f32)
        _lib.call("dhz_reverse_residual_bwd_dt", _p(dout), _p(dscale), _p(daw), B, Hres, Wres, C, shift, 1, ops._dt(dout), _stream())
        # (2) out-projection
        dctx = ops.gemm_dgrad(daw, wo_)
        g_wo, g_bo = _wgrad(daw, 0, cx, wo, bo)
        # (3) attention core
        dqkv = torch.empty_like(qkv)
        parts = _lib.load().dhz_ps_attn_bwd_parts_d(B_, H, C // H)
        dpart = torch.empty((parts, NTOK, NTOK), **f32) if bias is not None else None
        nW = mask.shape[0] if mask is not None else 1
        base, gb = qkv.data_ptr(), dqkv.data_ptr()
        _lib.call("dhz_ps_attn_bwd", base, base + 4 * C, base + 8 * C, 3 * C, _p(bias), _p(mask), _p(rank), _p(dctx), C,
                  gb, gb + 4 * C, gb + 8 * C, 3 * C, _p(dpart), B_, H, nW, 32, _stream())
        dtable = _table_backward(dpart, parts, table_p, H, dev) if bias is not None else None
        # (4) QKV projection
        dxn = ops.gemm_dgrad(dqkv, ops.cat_rows([wq_.detach(), wk_.detach(), wv_.detach()]))
        g_wq, g_bq, g_wk, g_bk, g_wv, g_bv = _wgrad_qkv(dqkv, xn, C, [(wq, bq), (wk, bk), (wv, bv)])
        # (5) LayerNorm backward + shortcut gradient in one pass
        dx, dgamma, dbeta = _ln_backward(dxn, x, gamma_p, beta_p, gamma, stats, dout, B, Hres, Wres, C, shift, 1)
        return (dx, dgamma, dbeta, g_wq, g_bq, g_wk, g_bk, g_wv, g_bv, g_wo, g_bo, dtable,
                None, None, None, None, None, None, None, None)


def fused_attn_branch(x, norm, layer, table, idx, mask, dscale, Hres, Wres, shift, heads):
    """x: [B,L,C]; norm: nn.LayerNorm; layer: AttentionLayer (query/key/value/out projections)."""
    q, k, v, o = layer.query_projection, layer.key_projection, layer.value_projection, layer.out_projection
    return _FusedAttnBranch.apply(x, norm.weight, norm.bias, q.weight, q.bias, k.weight, k.bias, v.weight, v.bias,
                                  o.weight, o.bias, table, idx, mask, dscale, Hres, Wres, shift, heads, torch.is_grad_enabled())


# ----------------------------------------------------------------------------- unfused forward, same hand-sequenced backward
class _AttnBranchChain(Function):
    """Attention branch for the widths the fused kernel does not cover (C >= 256) or where the kernel chain is
    faster: forward = dhz_ln_partition_fwd -> library GEMM -> dhz_ps_attn_fwd -> library GEMM ->
    dhz_reverse_residual_fwd, as ONE autograd node whose backward is the same sequence as _FusedAttnBranch's
    (shortcut gradient folded into the LayerNorm backward, weight gradients accumulated in place)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, wq, bq, wk, bk, wv, bv, wo, bo, table, idx, mask, dscale, Hres, Wres, shift, H, grad_mode):
        _require_gpu(x, gamma, beta, wq, wo, table, idx, mask, dscale)
        x = x.contiguous()
        B, L, C = x.shape
        d = C // H
        dev = x.device
        T = B * L
        f32 = dict(device=dev, dtype=torch.float32)
        xn = torch.empty((T, C), device=dev, dtype=x.dtype)
        stats = torch.empty((T, 2), **f32)
        _lib.call("dhz_ln_partition_fwd_dt", _p(x), _p(gamma), _p(beta), _p(xn), _p(stats), B, Hres, Wres, C, shift, 1,
                  ops._dt(x), _stream())
        wcat = ops.cat_rows([wq.detach(), wk.detach(), wv.detach()])
        qkv = ops.gemm_fwd(xn, wcat, ops.cat_rows([bq.detach(), bk.detach(), bv.detach()]))
        bias = None
        if table is not None:
            bias = torch.empty((H, NTOK, NTOK), **f32)
            _lib.call("dhz_bias_gather", _p(table.contiguous()), _p(bias), H, _stream())
        cx = torch.empty((T, C), device=dev, dtype=x.dtype)
        rank = torch.empty(((T // NTOK) * H * NTOK,), device=dev, dtype=torch.uint8)
        nW = mask.shape[0] if mask is not None else 1
        base = qkv.data_ptr()
        timing = ops.KERNEL_TIMING.get("dhz_ps_attn_fwd") if ops.KERNEL_TIMING is not None else None
        if timing is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        es = qkv.element_size()
        _lib.call("dhz_ps_attn_fwd_dt", base, base + es * C, base + 2 * es * C, 3 * C, _p(idx), _p(bias), _p(mask), _p(cx), C,
                  _p(rank), T // NTOK, H, nW, d, ops._dt(qkv), _stream())
        if timing is not None:
            e1.record()
            timing.append((e0, e1, (T // NTOK) * H * 4 * NTOK * d * qkv.element_size()))
        aw = ops.gemm_fwd(cx, wo, bo)
        out = torch.empty_like(x)
        _lib.call("dhz_reverse_residual_fwd_dt", _p(aw), _p(x), _p(dscale), _p(out), B, Hres, Wres, C, shift, 1, ops._dt(x), _stream())
        if grad_mode and any(ctx.needs_input_grad):
            ctx.save_for_backward(x, gamma, stats, xn, qkv, cx, rank, bias, mask, dscale, wq, wk, wv, wo)
            ctx.params = (wq, bq, wk, bk, wv, bv, wo, bo, gamma, beta, table)
            ctx.geom = (B, Hres, Wres, C, shift, H)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, gamma, stats, xn, qkv, cx, rank, bias, mask, dscale, wq_, wk_, wv_, wo_ = ctx.saved_tensors
        wq, bq, wk, bk, wv, bv, wo, bo, gamma_p, beta_p, table_p = ctx.params
        B, Hres, Wres, C, shift, H = ctx.geom
        d = C // H
        dout = dout.contiguous()
        dev = x.device
        T = B * Hres * Wres
        B_ = T // NTOK
        f32 = dict(device=dev, dtype=torch.float32)
        daw = torch.empty((T, C), device=dev, dtype=dout.dtype)
        _lib.call("dhz_reverse_residual_bwd_dt", _p(dout), _p(dscale), _p(daw), B, Hres, Wres, C, shift, 1, ops._dt(dout), _stream())
        dctx = ops.gemm_dgrad(daw, wo_)
        g_wo, g_bo = _wgrad(daw, 0, cx, wo, bo)
        dqkv = torch.empty_like(qkv)
        parts = _lib.load().dhz_ps_attn_bwd_parts_d(B_, H, C // H)
        dpart = torch.empty((parts, NTOK, NTOK), **f32) if bias is not None else None
        nW = mask.shape[0] if mask is not None else 1
        base, gb = qkv.data_ptr(), dqkv.data_ptr()
        es = qkv.element_size()
        _lib.call("dhz_ps_attn_bwd_dt", base, base + es * C, base + 2 * es * C, 3 * C, _p(bias), _p(mask), _p(rank), _p(dctx), C,
                  gb, gb + es * C, gb + 2 * es * C, 3 * C, _p(dpart), B_, H, nW, d, ops._dt(qkv), _stream())
        dtable = _table_backward(dpart, parts, table_p, H, dev) if bias is not None else None
        dxn = ops.gemm_dgrad(dqkv, ops.cat_rows([wq_.detach(), wk_.detach(), wv_.detach()]))
        g_wq, g_bq, g_wk, g_bk, g_wv, g_bv = _wgrad_qkv(dqkv, xn, C, [(wq, bq), (wk, bk), (wv, bv)])
        dx, dgamma, dbeta = _ln_backward(dxn, x, gamma_p, beta_p, gamma, stats, dout, B, Hres, Wres, C, shift, 1)
        return (dx, dgamma, dbeta, g_wq, g_bq, g_wk, g_bk, g_wv, g_bv, g_wo, g_bo, dtable,
                None, None, None, None, None, None, None, None)


def attn_branch(x, norm, layer, table, idx, mask, dscale, Hres, Wres, shift, heads):
    """Dispatch: fused kernel where it wins (measured, tools/bench_fused.py), kernel chain elsewhere."""
    C = x.shape[-1]
    q, k, v, o = layer.query_projection, layer.key_projection, layer.value_projection, layer.out_projection
    args = (x, norm.weight, norm.bias, q.weight, q.bias, k.weight, k.bias, v.weight, v.bias, o.weight, o.bias, table, idx,
            mask, dscale, Hres, Wres, shift, heads, torch.is_grad_enabled())
    use_fused = ENABLED and x.dtype == torch.float32 and C == 32 * heads and (C in (32, 64) or (C == 128 and Hres * Wres <= ATTN_FUSED_C128_MAX_HW))
    return (_FusedAttnBranch if use_fused else _AttnBranchChain).apply(*args)


# ----------------------------------------------------------------------------- LeFF branch as one autograd node
class _LeffBranch(Function):
    """out = x + drop_scale * linear2(gelu(dwconv3x3(gelu(linear1(LayerNorm(x))))))       (M1:873 + M1:496-534)
    forward : dhz_ln_partition_fwd (plain LN) -> library GEMM -> dhz_leff_dwconv_fwd -> library GEMM ->
              dhz_reverse_residual_fwd (token order)
    backward: the mirror sequence with in-place weight gradients and the shortcut gradient folded into the
              LayerNorm backward (no autograd accumulation kernels)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, w1, b1, wd, bd, w2, b2, dscale, Hres, Wres, grad_mode):
        _require_gpu(x, gamma, beta, w1, wd, w2, dscale)
        x = x.contiguous()
        B, L, C = x.shape
        Ch = w1.shape[0]
        T = B * L
        dev = x.device
        f32 = dict(device=dev, dtype=torch.float32)
        train = grad_mode and any(ctx.needs_input_grad)      # grad mode reads False inside Function.forward: passed in
        wdc = wd.contiguous()
        out = torch.empty_like(x)
        tiled = Hres % 8 == 0 and Wres % 16 == 0
        fp32 = x.dtype == torch.float32                      # the fused LeFF kernels are fp32-only
        if LEFF_FUSED and C in LEFF_FUSED_C and tiled and fp32 and not (C == 64 and T > LEFF_FUSED_C64_MAX_T and ops.SPLIT_BF16 == 6):
            # one kernel: norm2, linear1, GELU, depthwise 3x3, GELU, linear2, DropPath scale, residual (csrc/leff_fused.hip)
            xn = stats = u = tg = z = None
            if train:
                xn = torch.empty((T, C), **f32)
                stats = torch.empty((T, 2), **f32)
                u = torch.empty((T, Ch), **f32)
                tg = torch.empty((T, Ch), **f32)
                z = torch.empty((T, Ch), **f32)
            _lib.call("dhz_leff_fused_fwd", _p(x), _p(gamma), _p(beta), _p(w1), _p(b1), _p(wdc), _p(bd), _p(w2), _p(b2),
                      _p(dscale), _p(out), _p(xn), _p(stats), _p(u), _p(tg), _p(z), B, Hres, Wres, C, _stream())
        else:
            xn = torch.empty((T, C), device=dev, dtype=x.dtype)
            stats = torch.empty((T, 2), **f32)
            _lib.call("dhz_ln_partition_fwd_dt", _p(x), _p(gamma), _p(beta), _p(xn), _p(stats), B, L, 1, C, 0, 0, ops._dt(x), _stream())
            u = ops.gemm_fwd(xn, w1, b1)
            z = torch.empty_like(u)
            tg = torch.empty_like(u) if train else None
            _lib.call("dhz_leff_dwconv_fwd_dt", _p(u), _p(wdc), _p(bd), _p(tg), _p(z), B, Hres, Wres, Ch, ops._dt(u), _stream())
            y = ops.gemm_fwd(z, w2, b2)
            _lib.call("dhz_reverse_residual_fwd_dt", _p(y), _p(x), _p(dscale), _p(out), B, L, 1, C, 0, 0, ops._dt(x), _stream())
        if train:
            ctx.save_for_backward(x, gamma, stats, xn, u, tg, z, dscale, w1, wdc, w2)
            ctx.params = (w1, b1, wd, bd, w2, b2, gamma, beta)
            ctx.geom = (B, L, C, Ch, Hres, Wres)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, gamma, stats, xn, u, tg, z, dscale, w1_, wdc, w2_ = ctx.saved_tensors
        w1, b1, wd, bd, w2, b2, gamma_p, beta_p = ctx.params
        B, L, C, Ch, Hres, Wres = ctx.geom
        T = B * L
        dout = dout.contiguous()
        dev = x.device
        f32 = dict(device=dev, dtype=torch.float32)
        gwd, gbd = _grad_buf(wd), _grad_buf(bd)
        # DropPath scale of the branch output (per image): in fp32 it is folded into the consumers - linear2's weight gradient scales
        # the rows of dout as it stages them, the depthwise backward scales dz - instead of a scaled copy of dout (one pass over
        # [T, C] per block less); bf16 keeps the copy
        fold = dscale is not None and dout.dtype == torch.float32 and L % 32 == 0
        zscale = dscale if fold else None
        if dscale is not None and not fold:
            dy = torch.empty((T, C), device=dev, dtype=dout.dtype)
            _lib.call("dhz_reverse_residual_bwd_dt", _p(dout), _p(dscale), _p(dy), B, L, 1, C, 0, 0, ops._dt(dout), _stream())
        else:
            dy = dout.view(T, C)
        dz = ops.gemm_dgrad(dy, w2_)
        g_w2, g_b2 = _wgrad(dy, 0, z, w2, b2, (dscale, L) if fold else None)
        du = torch.empty_like(u)
        if gwd is not None and gbd is not None:          # depthwise weight / bias gradients straight into .grad
            _lib.call("dhz_leff_dwconv_bwd_scaled_dt", _p(dz), _p(u), _p(tg), _p(wdc), _p(du), _p(gwd), _p(gbd), _p(zscale), B, Hres,
                      Wres, Ch, ops._dt(u), _stream())
            _ready(wd, bd)
            g_wd = g_bd = None
        else:
            dwb = torch.zeros((Ch * 10,), **f32)
            _lib.call("dhz_leff_dwconv_bwd_scaled_dt", _p(dz), _p(u), _p(tg), _p(wdc), _p(du), dwb.data_ptr(),
                      dwb.data_ptr() + 4 * Ch * 9, _p(zscale), B, Hres, Wres, Ch, ops._dt(u), _stream())
            g_wd, g_bd = dwb[:Ch * 9].view(Ch, 1, 3, 3), dwb[Ch * 9:]
        dxn = ops.gemm_dgrad(du, w1_)
        g_w1, g_b1 = _wgrad(du, 0, xn, w1, b1)
        dx, dgamma, dbeta = _ln_backward(dxn, x, gamma_p, beta_p, gamma, stats, dout, B, L, 1, C, 0, 0)
        return (dx, dgamma, dbeta, g_w1, g_b1, g_wd, g_bd, g_w2, g_b2, None, None, None, None)


class _FfnBranch(Function):
    """out = x + drop_scale * fc2(gelu(fc1(LayerNorm(x))))        (M1:873 with token_mlp = 'ffn': M1:442-468)
    forward : dhz_ln_partition_fwd (plain LN) -> GEMM -> dhz_gelu_fwd -> GEMM -> dhz_reverse_residual_fwd (token order)
    backward: the mirror sequence, weight gradients in place, the DropPath factor folded into fc2's weight gradient (row scale) and
              into dhz_gelu_bwd, the shortcut gradient folded into the LayerNorm backward."""

    @staticmethod
    def forward(ctx, x, gamma, beta, w1, b1, w2, b2, dscale, grad_mode):
        _require_gpu(x, gamma, beta, w1, w2, dscale)
        x = x.contiguous()
        B, L, C = x.shape
        T = B * L
        dev = x.device
        xn = torch.empty((T, C), device=dev, dtype=x.dtype)
        stats = torch.empty((T, 2), device=dev, dtype=torch.float32)
        _lib.call("dhz_ln_partition_fwd_dt", _p(x), _p(gamma), _p(beta), _p(xn), _p(stats), B, L, 1, C, 0, 0, ops._dt(x), _stream())
        u = ops.gemm_fwd(xn, w1, b1)
        z = torch.empty_like(u)
        _lib.call("dhz_gelu_fwd_dt", _p(u), _p(z), u.numel(), ops._dt(u), _stream())
        y = ops.gemm_fwd(z, w2, b2)
        out = torch.empty_like(x)
        _lib.call("dhz_reverse_residual_fwd_dt", _p(y), _p(x), _p(dscale), _p(out), B, L, 1, C, 0, 0, ops._dt(x), _stream())
        if grad_mode and any(ctx.needs_input_grad):
            ctx.save_for_backward(x, gamma, stats, xn, u, z, dscale, w1, w2)
            ctx.params = (w1, b1, w2, b2, gamma, beta)
            ctx.geom = (B, L, C)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, gamma, stats, xn, u, z, dscale, w1_, w2_ = ctx.saved_tensors
        w1, b1, w2, b2, gamma_p, beta_p = ctx.params
        B, L, C = ctx.geom
        T = B * L
        Ch = u.shape[1]
        dout = dout.contiguous()
        fold = dscale is not None and dout.dtype == torch.float32 and L % 32 == 0
        if dscale is not None and not fold:
            dy = torch.empty((T, C), device=x.device, dtype=dout.dtype)
            _lib.call("dhz_reverse_residual_bwd_dt", _p(dout), _p(dscale), _p(dy), B, L, 1, C, 0, 0, ops._dt(dout), _stream())
        else:
            dy = dout.view(T, C)
        dz = ops.gemm_dgrad(dy, w2_)
        g_w2, g_b2 = _wgrad(dy, 0, z, w2, b2, (dscale, L) if fold else None)
        du = torch.empty_like(u)
        _lib.call("dhz_gelu_bwd_dt", _p(dz), _p(u), _p(du), u.numel(), _p(dscale) if fold else None, L * Ch if fold else 0, ops._dt(u),
                  _stream())
        dxn = ops.gemm_dgrad(du, w1_)
        g_w1, g_b1 = _wgrad(du, 0, xn, w1, b1)
        dx, dgamma, dbeta = _ln_backward(dxn, x, gamma_p, beta_p, gamma, stats, dout, B, L, 1, C, 0, 0)
        return (dx, dgamma, dbeta, g_w1, g_b1, g_w2, g_b2, None, None)


def ffn_branch(x, norm, mlp, dscale):
    return _FfnBranch.apply(x, norm.weight, norm.bias, mlp.fc1.weight, mlp.fc1.bias, mlp.fc2.weight, mlp.fc2.bias, dscale,
                            torch.is_grad_enabled())


def leff_branch(x, norm, mlp, dscale, Hres, Wres):
    return _LeffBranch.apply(x, norm.weight, norm.bias, mlp.linear1[0].weight, mlp.linear1[0].bias,
                             mlp.dwconv[0].weight, mlp.dwconv[0].bias, mlp.linear2[0].weight, mlp.linear2[0].bias,
                             dscale, Hres, Wres, torch.is_grad_enabled())
