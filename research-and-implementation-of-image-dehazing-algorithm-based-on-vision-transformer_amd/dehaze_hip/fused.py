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


def _wgrad(dy, off, x, w, b):
    """dW/db of one Linear from dy[:, off:off+N] and x.  In place into .grad when the parameter is a leaf and
    the split-T kernel is the better choice; returns (dw, db) to hand to autograd, or (None, None)."""
    T, K = x.shape
    N = w.shape[0]
    mine = T >= 16384 or N * K < 200000
    if mine and w.is_leaf and (b is None or b.is_leaf):
        ops._accumulate_param_grads(dy, off, x, [(w, b)])
        return None, None
    if mine:
        dw = torch.zeros_like(w, memory_format=torch.contiguous_format)
        db = torch.zeros_like(b) if b is not None else None
        _lib.call("dhz_linear_wgrad", dy.data_ptr() + 4 * off, dy.stride(0), _p(x), x.stride(0), T, N, K, _p(dw), _p(db),
                  _stream())
        return dw, db
    dys = dy[:, off:off + N]
    return dys.t() @ x, (dys.sum(0) if b is not None else None)


class _FusedAttnBranch(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, wq, bq, wk, bk, wv, bv, wo, bo, table, idx, mask, dscale, Hres, Wres, shift, H):
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
        bqkv = torch.cat([bq, bk, bv])
        bias = None
        if table is not None:
            bias = torch.empty((H, NTOK, NTOK), **f32)
            _lib.call("dhz_bias_gather", _p(table.contiguous()), _p(bias), H, _stream())
        out = torch.empty_like(x)
        train = any(ctx.needs_input_grad)
        xn = qkv = cx = stats = rank = None
        if train:
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
        if train:
            ctx.save_for_backward(x, gamma, stats, xn, qkv, cx, rank, bias, mask, dscale, wq, wk, wv, wo)
            ctx.params = (wq, bq, wk, bk, wv, bv, wo, bo)
            ctx.geom = (B, Hres, Wres, C, shift, H)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, gamma, stats, xn, qkv, cx, rank, bias, mask, dscale, wq_, wk_, wv_, wo_ = ctx.saved_tensors
        wq, bq, wk, bk, wv, bv, wo, bo = ctx.params
        B, Hres, Wres, C, shift, H = ctx.geom
        dout = dout.contiguous()
        dev = x.device
        T = B * Hres * Wres
        B_ = T // NTOK
        f32 = dict(device=dev, dtype=torch.float32)
        # (1) gradient of the window-ordered out-projection output
        daw = torch.empty((T, C), **f32)
        _lib.call("dhz_reverse_residual_bwd", _p(dout), _p(dscale), _p(daw), B, Hres, Wres, C, shift, 1, _stream())
        # (2) out-projection
        dctx = daw @ wo_
        g_wo, g_bo = _wgrad(daw, 0, cx, wo, bo)
        # (3) attention core
        dqkv = torch.empty_like(qkv)
        parts = _lib.load().dhz_ps_attn_bwd_parts(B_, H)
        dpart = torch.empty((parts, NTOK, NTOK), **f32) if bias is not None else None
        nW = mask.shape[0] if mask is not None else 1
        base, gb = qkv.data_ptr(), dqkv.data_ptr()
        _lib.call("dhz_ps_attn_bwd", base, base + 4 * C, base + 8 * C, 3 * C, _p(bias), _p(mask), _p(rank), _p(dctx), C,
                  gb, gb + 4 * C, gb + 8 * C, 3 * C, _p(dpart), B_, H, nW, 32, _stream())
        dtable = None
        if bias is not None:
            dtable = torch.empty((225, H), **f32)
            _lib.call("dhz_bias_table_grad", _p(dpart), parts, _p(dtable), H, 0, _stream())
        # (4) QKV projection
        dxn = dqkv @ torch.cat([wq_, wk_, wv_], 0)
        g_wq, g_bq = _wgrad(dqkv, 0, xn, wq, bq)
        g_wk, g_bk = _wgrad(dqkv, C, xn, wk, bk)
        g_wv, g_bv = _wgrad(dqkv, 2 * C, xn, wv, bv)
        # (5) LayerNorm backward + shortcut gradient in one pass
        dx = torch.empty_like(x)
        dgb = torch.zeros((2, C), **f32)
        _lib.call("dhz_ln_partition_bwd", _p(dxn), _p(x), _p(gamma), _p(stats), _p(dout), _p(dx), dgb[0].data_ptr(),
                  dgb[1].data_ptr(), B, Hres, Wres, C, shift, 1, _stream())
        return (dx, dgb[0], dgb[1], g_wq, g_bq, g_wk, g_bk, g_wv, g_bv, g_wo, g_bo, dtable,
                None, None, None, None, None, None, None)


def fused_attn_branch(x, norm, layer, table, idx, mask, dscale, Hres, Wres, shift, heads):
    """x: [B,L,C]; norm: nn.LayerNorm; layer: AttentionLayer (query/key/value/out projections)."""
    q, k, v, o = layer.query_projection, layer.key_projection, layer.value_projection, layer.out_projection
    return _FusedAttnBranch.apply(x, norm.weight, norm.bias, q.weight, q.bias, k.weight, k.bias, v.weight, v.bias,
                                  o.weight, o.bias, table, idx, mask, dscale, Hres, Wres, shift, heads)
