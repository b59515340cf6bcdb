"""torch.autograd.Function wrappers over the C-ABI (include/dehaze_hip.h).

PyTorch is plumbing here: it owns device memory (caching allocator), the current HIP stream and the
autograd tape; the ops below run hand-written gfx950 kernels through ctypes (the token-Linear GEMMs included:
dhz_linear_fwd / dhz_linear_dgrad / dhz_linear_wgrad).  No vendor compute kernel is left in the fp32 or the bf16 training
step; the library is reached only for shapes the kernels do not tile (e.g. the last VGG layer on 24 x 24 maps of 384 x 384
patches, Downsample / projections of unusual channel counts - DESIGN.md section 2).  CPU tensors are rejected - there is no
fallback path.
"""
import collections
import ctypes

import os

import torch
from torch.autograd import Function

from . import _lib

NTOK, NTOP = 64, 25

# bench.py sets this to {"<entry point>": []} to collect (start_event, end_event, units) per launch of
# that kernel, recorded on the stream the kernel is launched on (torch's current stream).
KERNEL_TIMING = None


def _stream():
    return torch.cuda.current_stream().cuda_stream


# The last tensors whose device pointers were handed to a launch: a pointer argument is often taken from a temporary
# (`_p(g.contiguous())`, `_p(w.to(dtype))`), which Python drops as soon as `_p` returns - before the launch is even enqueued.
# Stream order makes a later reuse of that block by the same stream harmless, but not a reuse from another stream (the
# reducer's, a loader's).  Keeping the most recent ones referenced closes the whole class; 64 covers the longest argument list.
_RECENT = collections.deque(maxlen=64)


def _p(t):
    if t is None:
        return None
    _RECENT.append(t)
    return t.data_ptr()


def _require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("dehaze_hip: this op runs only on a HIP device (got a CPU tensor); "
                               "there is deliberately no CPU/PyTorch fallback")
        if t is not None and t.dtype not in (torch.float32, torch.uint8, torch.bfloat16):
            raise RuntimeError(f"dehaze_hip: fp32 (or bf16 token tensors, BASELINE config 4) expected, got {t.dtype}")


BF16 = torch.bfloat16

_WARNED = set()


def warn_library_fallback(what, shape):
    """One warning per (layer kind, shape): a shape the hand-written kernels do not tile went to the library convolution."""
    key = (what, tuple(shape))
    if key not in _WARNED:
        _WARNED.add(key)
        import warnings
        warnings.warn(f"dehaze_hip: {what} with shape {tuple(shape)} is not tiled by the HIP kernels - running the library "
                      "convolution (MIOpen) for it", stacklevel=3)


def _dt(t):
    """dtype code of the dhz_*_dt entry points (include/dehaze_hip.h: DHZ_F32 = 0, DHZ_BF16 = 1)."""
    return 1 if t.dtype == BF16 else 0


# Derived copies of the flat fp32 parameter buffer that FlatAdamW keeps for the GEMMs:
#   BF16_SHADOW  = [flat f32, its bf16 copy]                      (config 4: the bf16 GEMMs' weight operand)
#   SPLIT_SHADOW = [flat f32, hi plane, mid plane, lo plane]      (the six-term split GEMMs' pre-split weight operand, bf16 each)
# A weight that is a view of the flat buffer gets the matching views - one launch per optimizer step for all parameters instead
# of one per Linear and pass.  The optimizer kernel keeps them current.  Writes that bypass it (a loaded checkpoint, a landscape
# probe, `with torch.no_grad(): p.add_(...)`) bump the autograd version counter of the PARAMETER (parameters are views with counters
# of their own: the flat buffer's counter does not see them).  Two guards, both through SHADOW_OWNER (a weak reference to the
# FlatAdamW that registered the copies - weak, so that dropping the optimizer frees its buffers and un-registers the copies):
#   * every lookup (split_planes, split_planes_t, bf16_copy, bf16_copy_t) validates itself: the counters of the parameters that
#     overlap the requested region are compared with the values recorded when the copies were derived (_shadow_fresh); on a
#     mismatch ALL copies are re-derived before the views are handed out.  So any entry that reaches a GEMM - a block, a layer, a
#     bare ops.* call - multiplies with current weights, not only train_step / Uformer.forward;
#   * sync_shadows() (train_step, Uformer.forward) checks all parameters at once.
# NOT seen by either: writes that bump no counter - `p.data.copy_(...)` / `p.data.add_(...)` (a fresh view with a fresh counter) and
# raw-pointer kernels.  After such a write call FlatAdamW.sync_shadows(force=True).
BF16_SHADOW = None
SPLIT_SHADOW = None
SHADOW_OWNER = None          # weakref.ref(FlatAdamW) or None


def _shadow_owner():
    global SHADOW_OWNER
    if SHADOW_OWNER is None:
        return None
    o = SHADOW_OWNER()
    if o is None:                # the optimizer is gone: its copies are meaningless
        SHADOW_OWNER = None
        set_bf16_shadow(None, None)
        set_split_shadow(None, None)
    return o


def sync_shadows(force=False):
    o = _shadow_owner()
    if o is not None:
        o.sync_shadows(force)


def _shadow_fresh(off, n):
    """called by the lookups with the flat-buffer region they are about to hand out views for"""
    o = _shadow_owner()
    if o is not None and not o.region_current(off, n):
        o.sync_shadows()

# Which matrix pipe takes the PRODUCTS of the fp32 path's GEMM-shaped kernels (storage, accumulation, bias / statistics stay fp32):
#   6 (default) - the bf16 pipe by operand splitting into three bf16 pieces (hi + mid + lo = all 24 mantissa bits, exactly) and the
#                 six products down to 2^-16 (hh, hm, mh, hl, lh, mm); what is dropped (ml, lm, ll) is <= 2^-24 relative - the size of
#                 ONE fp32 rounding, i.e. the error class of an fp32 FMA chain (csrc/linear_split.hip; every kernel-level fp32
#                 tolerance of tests/ holds under it).  6 / 16 of the fp32 pipe's matrix time.
#   0           - the fp32 matrix pipe itself (v_mfma_f32_16x16x4_f32, csrc/linear_gemm.hip / linear_wgrad.hip): bench.py times the
#                 same step on it as `fp32_pipe` beside the headline.
#   3           - EXPERIMENT, never a product setting: hi / lo pieces, three products (~16 mantissa bits per product).
# DHZ_SPLIT_BF16 in the environment overrides the default (1 means 3).
def _split_terms(v):
    try:
        return {0: 0, 1: 3, 3: 3, 6: 6}[int(v)]
    except (KeyError, ValueError):
        raise ValueError(f"DHZ_SPLIT_BF16={v!r}: expected 0 (fp32 matrix pipe), 6 (default: six-term bf16 split) or 3 (or 1; experiment)") from None


SPLIT_BF16 = _split_terms(os.environ.get("DHZ_SPLIT_BF16", "6"))
SPLIT_MIN_K = int(os.environ.get("DHZ_SPLIT_MIN_K", "128"))      # smallest contraction the forward / backward-data GEMMs split


def set_bf16_shadow(f32, b16, b16t=None, desc=None, index=None, ntiles=0):
    """b16: bf16 mirror of the flat buffer; b16t (optional): the same with every registered matrix stored TRANSPOSED at its offset
    (desc / ntiles: the table of dhz_bf16_transpose_batched, index: {(offset, rows, cols)})."""
    global BF16_SHADOW
    BF16_SHADOW = None if f32 is None else [f32, b16, b16t, desc, index or set(), int(ntiles)]


def refresh_bf16_shadow_t():
    sh = BF16_SHADOW
    if sh is not None and sh[2] is not None:
        _lib.call("dhz_bf16_transpose_batched", sh[0].data_ptr(), sh[2].data_ptr(), sh[3].data_ptr(), sh[3].shape[0], sh[5], _stream())


def bf16_copy_t(W):
    """bf16 copy of W^T ([K, N] row-major for W [N, K]) when FlatAdamW keeps one for this matrix, else None."""
    sh = BF16_SHADOW
    if sh is None or sh[2] is None or not W.is_contiguous() or W.dim() != 2:
        return None
    o = _view_of(sh[0], W)
    if o < 0 or (o, W.shape[0], W.shape[1]) not in sh[4]:
        return None
    _shadow_fresh(o, W.numel())
    return sh[2][o: o + W.numel()]


def refresh_bf16_shadow():
    sh = BF16_SHADOW
    if sh is not None:
        sh[1].copy_(sh[0])                   # one cast launch for all parameters
        refresh_bf16_shadow_t()


def _view_of(flat, W):
    """element offset of contiguous W inside the flat buffer, or -1"""
    off = W.data_ptr() - flat.data_ptr()
    if 0 <= off < 4 * flat.numel() and W.untyped_storage().data_ptr() == flat.untyped_storage().data_ptr():
        return off // 4
    return -1


def bf16_copy(W):
    """bf16 copy of an fp32 master weight for the bf16 GEMMs (parameters stay fp32)."""
    if W.dtype == BF16:
        return W
    sh = BF16_SHADOW
    if sh is not None and W.is_contiguous():
        o = _view_of(sh[0], W)
        if o >= 0:
            _shadow_fresh(o, W.numel())
            return sh[1][o: o + W.numel()].view(W.shape)
    return W.detach().to(BF16)


def set_split_shadow(f32, planes, planes_t=None, desc=None, index=None, ntiles=0):
    """planes: [3, n] bf16 mirror of the flat buffer; planes_t (optional): the same with every registered matrix stored TRANSPOSED at
    its offset; desc / ntiles: device int32 [nmat, 4] table of dhz_split3_planes_t and its tile count; index: {(offset, rows, cols)} of
    the registered matrices."""
    global SPLIT_SHADOW
    SPLIT_SHADOW = None if f32 is None else [f32, planes[0], planes[1], planes[2], planes_t, desc, index or set(), int(ntiles)]


def refresh_split_shadow():
    sh = SPLIT_SHADOW
    if sh is not None:
        _lib.call("dhz_split3_planes", sh[0].data_ptr(), sh[0].numel(), sh[1].data_ptr(), sh[2].data_ptr(), sh[3].data_ptr(), _stream())
        if sh[4] is not None:
            pt, desc = sh[4], sh[5]
            _lib.call("dhz_split3_planes_t", sh[0].data_ptr(), pt[0].data_ptr(), pt[1].data_ptr(), pt[2].data_ptr(), desc.data_ptr(),
                      desc.shape[0], sh[7], _stream())


def split_planes_t(W):
    """(hi, mid, lo) planes of W^T ([K, N] row-major for W [N, K]) when FlatAdamW keeps them for this matrix (every Linear weight of
    the flat buffer; the adjacent Q / K / V weights as ONE packed [3C, C] matrix), else None."""
    sh = SPLIT_SHADOW
    if sh is None or sh[4] is None or not W.is_contiguous() or W.dim() != 2:
        return None
    o = _view_of(sh[0], W)
    if o < 0 or (o, W.shape[0], W.shape[1]) not in sh[6]:
        return None
    n = W.numel()
    _shadow_fresh(o, n)
    pt = sh[4]
    return pt[0][o: o + n], pt[1][o: o + n], pt[2][o: o + n]


def split_planes(W):
    """(hi, mid, lo): the three bf16 truncation pieces of a contiguous fp32 weight (hi + mid + lo == W exactly) - views of
    FlatAdamW's planes when W lives in its flat buffer, otherwise one dhz_split3_planes launch."""
    assert W.dtype == torch.float32 and W.is_contiguous() and W.numel() % 8 == 0
    sh = SPLIT_SHADOW
    if sh is not None:
        o = _view_of(sh[0], W)
        if o >= 0 and o % 8 == 0:
            n = W.numel()
            _shadow_fresh(o, n)
            return sh[1][o: o + n], sh[2][o: o + n], sh[3][o: o + n]
    pl = torch.empty((3, W.numel()), device=W.device, dtype=BF16)
    _lib.call("dhz_split3_planes", _p(W), W.numel(), pl[0].data_ptr(), pl[1].data_ptr(), pl[2].data_ptr(), _stream())
    _RECENT.append(pl)
    return pl[0], pl[1], pl[2]


_ROUTE6_FORCE = os.environ.get("DHZ_S6_ROUTE", "")
_NO_TPLANES = bool(os.environ.get("DHZ_S6_NO_TPLANES"))       # diagnostics: backward-data through the transposed-read kernel


def _route6(T, contraction, out, dgrad):
    """which kernel takes a six-term GEMM of T tokens (measured per shape on the config-2 step, tools/bench_split6.py):
    'new' = csrc/split6_gemm.hip (pre-split weight planes; 256 x 128 / 128 x 128 tiles), 'old' = csrc/linear_split.hip (both
    operands split in the kernel; 128 x 64 tiles, two workgroups per CU: ahead on narrow outputs and few-tile problems),
    'f32' = the fp32 pipe (HBM-bound shapes where the split buys nothing)."""
    if contraction % 32 or out % 32 or contraction < 64 or out < 64:
        return "f32"
    old_ok = contraction % 64 == 0 and out % 64 == 0
    if _ROUTE6_FORCE == "old" and old_ok and contraction >= 128:          # diagnostics: the round-3 dispatch
        return "old"
    if _ROUTE6_FORCE == "old":
        return "f32"
    new_ok = not dgrad or out % 64 == 0
    if T <= 2048 and out < 2048 and old_ok:
        return "old"
    if dgrad and out == 64 and old_ok:
        return "old"
    if not dgrad and contraction <= 64 and out % 128 and T >= (1 << 19):
        return "f32"
    if new_ok and out <= 2048:
        return "new"
    return "old" if old_ok else "f32"


def _terms():
    """3 or 6 for the split kernels (True counts as 3)"""
    return 6 if SPLIT_BF16 == 6 else 3


def _timed(name):
    """(list, start event) when bench.py collects HIP-event timings for this entry point, else None."""
    lst = KERNEL_TIMING.get(name) if KERNEL_TIMING is not None else None
    if lst is None:
        return None
    e0 = torch.cuda.Event(enable_timing=True)
    e0.record()
    return lst, e0


def _timed_end(ev, units):
    if ev is not None:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        ev[0].append((ev[1], e1, units))


def gemm_fwd(x, W, b=None):
    """y[T,N] = x[T,K] W[N,K]^T + b on the fp32 matrix pipe (dhz_linear_fwd).  x: rows of K contiguous floats (any row
    stride), W contiguous."""
    _require_gpu(x, W, b)
    T, K = x.shape
    N = W.shape[0]
    assert x.stride(1) == 1 and W.shape[1] == K
    W = W if W.is_contiguous() else W.contiguous()
    y = torch.empty((T, N), device=x.device, dtype=x.dtype)
    if x.dtype == BF16:                     # bf16 activations x bf16 weight copy, fp32 accumulate, fp32 bias (config 4)
        Wb = bf16_copy(W)
        ev = _timed("dhz_linear_bf16")
        _lib.call("dhz_linear_fwd_bf16", _p(x), x.stride(0), _p(Wb), _p(b), _p(y), N, T, N, K, _stream())
        _timed_end(ev, 2.0 * T * N * K)
    elif SPLIT_BF16 == 6 and x.stride(0) % 4 == 0 and _route6(T, K, N, False) != "f32":
        if _route6(T, K, N, False) == "new":
            hi, mid, lo = split_planes(W)
            ev = _timed("dhz_linear_split6")
            _lib.call("dhz_linear_fwd_split6", _p(x), x.stride(0), _p(hi), _p(mid), _p(lo), _p(b), _p(y), N, T, N, K, _stream())
        else:
            ev = _timed("dhz_linear_split6")
            _lib.call("dhz_linear_fwd_split", _p(x), x.stride(0), _p(W), _p(b), _p(y), N, T, N, K, 6, _stream())
        _timed_end(ev, 12.0 * T * N * K)                         # ISSUED bf16 FLOPs: six products per multiply-add
    elif SPLIT_BF16 == 3 and K >= SPLIT_MIN_K and K % 64 == 0 and N % 64 == 0:      # experiment
        _lib.call("dhz_linear_fwd_split", _p(x), x.stride(0), _p(W), _p(b), _p(y), N, T, N, K, 3, _stream())
    else:
        _lib.call("dhz_linear_fwd", _p(x), x.stride(0), _p(W), _p(b), _p(y), N, T, N, K, _stream())
    return y


RES_EPILOGUE = not os.environ.get("DHZ_NO_RES_EPILOGUE")        # A/B switch: K4 as a pass of its own (the pre-round-6 chain)


def gemm_fwd_res(x, W, b, res, scale, B, Hres, Wres, shift, windowed):
    """out = res + scale[image] * (x W^T + b) with the rows stored at their token-order position: K4 (window reverse, un-roll,
    DropPath factor, residual; M1:859-873) as the EPILOGUE of the out-projection / linear2 GEMM (dhz_linear_fwd_split6_res,
    csrc/tok_epilogue.h).  x: [T, K] (window order when `windowed`, the order dhz_ln_partition_fwd writes); res: [T, N] in token
    order; scale: [B] or None.  Shapes / arithmetic the epilogue kernels do not cover (bf16 storage, the fp32 matrix pipe) run
    the GEMM and dhz_reverse_residual_fwd as two launches."""
    _require_gpu(x, W, b, res, scale)
    T, K = x.shape
    N = W.shape[0]
    HW = Hres * Wres
    assert x.stride(1) == 1 and W.shape[1] == K and res.is_contiguous() and T == B * HW
    route = "f32"
    if RES_EPILOGUE and x.dtype == torch.float32 and SPLIT_BF16 == 6 and x.stride(0) % 4 == 0 and HW % 64 == 0 and W.is_contiguous():
        route = _route6(T, K, N, False)
    if RES_EPILOGUE and x.dtype == BF16 and res.dtype == BF16 and HW % 64 == 0 and N % 64 == 0 and K % 64 == 0 and x.stride(0) % 8 == 0:
        Wb = bf16_copy(W if W.is_contiguous() else W.contiguous())
        out = torch.empty_like(res)
        ev = _timed("dhz_linear_bf16")
        _lib.call("dhz_linear_fwd_bf16_res", _p(x), x.stride(0), _p(Wb), _p(b), _p(res), _p(scale), _p(out), N, T, N, K, HW, Hres, Wres, shift,
                  1 if windowed else 0, _stream())
        _timed_end(ev, 2.0 * T * N * K)
        return out
    if route == "f32":
        y = gemm_fwd(x, W, b)
        out = torch.empty_like(res)
        _lib.call("dhz_reverse_residual_fwd_dt", _p(y), _p(res), _p(scale), _p(out), B, Hres, Wres, N, shift, 1 if windowed else 0,
                  _dt(res), _stream())
        return out
    out = torch.empty_like(res)
    ev = _timed("dhz_linear_split6")
    if route == "new":
        hi, mid, lo = split_planes(W)
        _lib.call("dhz_linear_fwd_split6_res", _p(x), x.stride(0), _p(hi), _p(mid), _p(lo), _p(b), _p(res), _p(scale), _p(out), N, T, N, K,
                  HW, Hres, Wres, shift, 1 if windowed else 0, _stream())
    else:
        _lib.call("dhz_linear_fwd_split_res", _p(x), x.stride(0), _p(W), _p(b), _p(res), _p(scale), _p(out), N, T, N, K, HW, Hres, Wres,
                  shift, 1 if windowed else 0, 6, _stream())
    _timed_end(ev, 12.0 * T * N * K)
    return out


def gemm_dgrad(dy, W, row_scale=None):
    """dx[T,K] = dy[T,N] W[N,K] (dhz_linear_dgrad).  row_scale = (scale[B], rows_per_image): dx rows carry the per-image factor
    (the DropPath factor of the branch in the backward pass) - in the epilogue of the six-term kernels, a pass of its own elsewhere."""
    _require_gpu(dy, W)
    T, N = dy.shape
    K = W.shape[1]
    assert dy.stride(1) == 1 and W.shape[0] == N
    W = W if W.is_contiguous() else W.contiguous()
    if row_scale is not None:
        sc, rows = row_scale
        assert T % rows == 0
        f32ok = RES_EPILOGUE and dy.dtype == torch.float32 and SPLIT_BF16 == 6 and dy.stride(0) % 4 == 0 and rows % 64 == 0
        if f32ok and _route6(T, N, K, False) == "new" and not _NO_TPLANES and split_planes_t(W) is not None:
            hi, mid, lo = split_planes_t(W)
            dx = torch.empty((T, K), device=dy.device, dtype=dy.dtype)
            ev = _timed("dhz_linear_split6")
            _lib.call("dhz_linear_fwd_split6_res", _p(dy), dy.stride(0), _p(hi), _p(mid), _p(lo), None, None, _p(sc), _p(dx), K, T, K, N,
                      rows, 0, 0, 0, 0, _stream())
            _timed_end(ev, 12.0 * T * N * K)
            return dx
        if f32ok and _route6(T, N, K, True) == "old":
            dx = torch.empty((T, K), device=dy.device, dtype=dy.dtype)
            ev = _timed("dhz_linear_split6")
            _lib.call("dhz_linear_dgrad_split_scaled", _p(dy), dy.stride(0), _p(W), _p(sc), _p(dx), K, T, N, K, rows, 6, _stream())
            _timed_end(ev, 12.0 * T * N * K)
            return dx
        dx = gemm_dgrad(dy, W)
        out = torch.empty_like(dx)
        _lib.call("dhz_reverse_residual_bwd_dt", _p(dx), _p(sc), _p(out), T // rows, rows, 1, K, 0, 0, _dt(dx), _stream())
        return out
    dx = torch.empty((T, K), device=dy.device, dtype=dy.dtype)
    if dy.dtype == BF16:
        Wt = None if _NO_TPLANES else bf16_copy_t(W)
        ev = _timed("dhz_linear_bf16")
        if Wt is not None:
            # dx = dy . W = dy . (W^T)^T: the forward kernel (software-pipelined, csrc/gemm_bf16_pipe.hip) on the optimizer's bf16 copy of W^T
            _lib.call("dhz_linear_fwd_bf16", _p(dy), dy.stride(0), Wt.data_ptr(), None, _p(dx), K, T, K, N, _stream())
        else:
            Wb = bf16_copy(W)
            _lib.call("dhz_linear_dgrad_bf16", _p(dy), dy.stride(0), _p(Wb), _p(dx), K, T, N, K, _stream())
        _timed_end(ev, 2.0 * T * N * K)
    elif SPLIT_BF16 == 6 and dy.stride(0) % 4 == 0 and _route6(T, N, K, False) == "new" and not _NO_TPLANES \
            and split_planes_t(W) is not None:
        # dx = dy . W = dy . (W^T)^T: the FORWARD kernel on the planes of W^T (kept by the optimizer) - no transposed fragment reads
        hi, mid, lo = split_planes_t(W)
        ev = _timed("dhz_linear_split6")
        _lib.call("dhz_linear_fwd_split6", _p(dy), dy.stride(0), _p(hi), _p(mid), _p(lo), None, _p(dx), K, T, K, N, _stream())
        _timed_end(ev, 12.0 * T * N * K)
    elif SPLIT_BF16 == 6 and dy.stride(0) % 4 == 0 and _route6(T, N, K, True) != "f32":
        ev = _timed("dhz_linear_split6")
        if _route6(T, N, K, True) == "new":
            hi, mid, lo = split_planes(W)
            _lib.call("dhz_linear_dgrad_split6", _p(dy), dy.stride(0), _p(hi), _p(mid), _p(lo), _p(dx), K, T, N, K, _stream())
        else:
            _lib.call("dhz_linear_dgrad_split", _p(dy), dy.stride(0), _p(W), _p(dx), K, T, N, K, 6, _stream())
        _timed_end(ev, 12.0 * T * N * K)
    elif SPLIT_BF16 == 3 and N >= SPLIT_MIN_K and N % 64 == 0 and K % 64 == 0:      # experiment
        _lib.call("dhz_linear_dgrad_split", _p(dy), dy.stride(0), _p(W), _p(dx), K, T, N, K, 3, _stream())
    else:
        _lib.call("dhz_linear_dgrad", _p(dy), dy.stride(0), _p(W), _p(dx), K, T, N, K, _stream())
    return dx


# ----------------------------------------------------------------------------- K3 / K7
class _PSWindowAttention(Function):
    """ProbAttention.forward (ATT:287-342) on a packed QKV buffer.

    qkv   : [T, 3C] (T = B_*64 window-ordered tokens; columns [Q | K | V], each [H, d])
    table : [225, H] relative position bias table, or None (options.is_relative_position_bias False)
    idx   : [64, 25] uint8 sampled keys;  mask: [nW, 64, 64] or None
    """

    @staticmethod
    def forward(ctx, qkv, table, idx, mask, H, d):
        _require_gpu(qkv, table, idx, mask)
        T, C3 = qkv.shape
        C = H * d
        assert C3 == 3 * C and T % NTOK == 0 and qkv.is_contiguous()
        B_ = T // NTOK
        out = torch.empty((T, C), device=qkv.device, dtype=qkv.dtype)
        rank = torch.empty((B_ * H * NTOK,), device=qkv.device, dtype=torch.uint8)
        bias = None
        if table is not None:
            bias = torch.empty((H, NTOK, NTOK), device=qkv.device, dtype=torch.float32)
            _lib.call("dhz_bias_gather", _p(table.contiguous()), _p(bias), H, _stream())
        nW = mask.shape[0] if mask is not None else 1
        base, es = qkv.data_ptr(), qkv.element_size()
        timing = KERNEL_TIMING.get("dhz_ps_attn_fwd") if KERNEL_TIMING is not None else None
        if timing is not None:      # HIP events on the launch stream, bracketing exactly this kernel
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        _lib.call("dhz_ps_attn_fwd_dt", base, base + es * C, base + 2 * es * C, 3 * C, _p(idx), _p(bias), _p(mask), _p(out), C,
                  _p(rank), B_, H, nW, d, _dt(qkv), _stream())
        if timing is not None:
            e1.record()
            timing.append((e0, e1, B_ * H * 4 * NTOK * d * qkv.element_size()))
        ctx.save_for_backward(qkv, bias, mask, rank)
        ctx.dims = (B_, H, d, nW)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, bias, mask, rank = ctx.saved_tensors
        B_, H, d, nW = ctx.dims
        C = H * d
        dout = dout.contiguous()
        dqkv = torch.empty_like(qkv)
        lib = _lib.load()
        dpart, dtable = None, None
        parts = lib.dhz_ps_attn_bwd_parts_d(B_, H, d)
        if bias is not None:
            dpart = torch.empty((parts, NTOK, NTOK), device=qkv.device, dtype=torch.float32)
        base, gb, es = qkv.data_ptr(), dqkv.data_ptr(), qkv.element_size()
        _lib.call("dhz_ps_attn_bwd_dt", base, base + es * C, base + 2 * es * C, 3 * C, _p(bias), _p(mask), _p(rank), _p(dout), C,
                  gb, gb + es * C, gb + 2 * es * C, 3 * C, _p(dpart), B_, H, nW, d, _dt(qkv), _stream())
        if bias is not None:
            dtable = torch.empty((225, H), device=qkv.device, dtype=torch.float32)
            _lib.call("dhz_bias_table_grad", _p(dpart), parts, _p(dtable), H, 0, _stream())
        return dqkv, dtable, None, None, None, None


def ps_window_attention(qkv, table, idx, mask, H, d):
    return _PSWindowAttention.apply(qkv, table, idx, mask, H, d)


def ps_window_attention_rank(qkv, table, idx, mask, H, d):
    """Forward only; also returns the saved selection ranks [B_,H,64] (tests / diagnostics)."""
    T = qkv.shape[0]
    C = H * d
    B_ = T // NTOK
    out = torch.empty((T, C), device=qkv.device, dtype=torch.float32)
    rank = torch.empty((B_, H, NTOK), device=qkv.device, dtype=torch.uint8)
    bias = None
    if table is not None:
        bias = torch.empty((H, NTOK, NTOK), device=qkv.device, dtype=torch.float32)
        _lib.call("dhz_bias_gather", _p(table.contiguous()), _p(bias), H, _stream())
    nW = mask.shape[0] if mask is not None else 1
    base = qkv.data_ptr()
    _lib.call("dhz_ps_attn_fwd", base, base + 4 * C, base + 8 * C, 3 * C, _p(idx), _p(bias), _p(mask), _p(out), C,
              _p(rank), B_, H, nW, d, _stream())
    return out, rank


class _DenseWindowAttention(Function):
    """Dense window attention (My_model twin, M0:428-492) on a packed [T,3C] QKV buffer."""

    @staticmethod
    def forward(ctx, qkv, table, mask, H, d, scale):
        _require_gpu(qkv, table, mask)
        T, C3 = qkv.shape
        C = H * d
        assert C3 == 3 * C and T % NTOK == 0 and qkv.is_contiguous()
        B_ = T // NTOK
        out = torch.empty((T, C), device=qkv.device, dtype=torch.float32)
        bias = torch.empty((H, NTOK, NTOK), device=qkv.device, dtype=torch.float32)
        _lib.call("dhz_bias_gather", _p(table.contiguous()), _p(bias), H, _stream())
        nW = mask.shape[0] if mask is not None else 1
        base = qkv.data_ptr()
        _lib.call("dhz_dense_attn_fwd", base, base + 4 * C, base + 8 * C, 3 * C, _p(bias), _p(mask), _p(out), C, B_, H,
                  nW, d, float(scale), _stream())
        ctx.save_for_backward(qkv, bias, mask)
        ctx.dims = (B_, H, d, nW, float(scale))
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, bias, mask = ctx.saved_tensors
        B_, H, d, nW, scale = ctx.dims
        C = H * d
        dout = dout.contiguous()
        dqkv = torch.empty_like(qkv)
        parts = _lib.load().dhz_ps_attn_bwd_parts(B_, H)
        dpart = torch.empty((parts, NTOK, NTOK), device=qkv.device, dtype=torch.float32)
        base, gb = qkv.data_ptr(), dqkv.data_ptr()
        _lib.call("dhz_dense_attn_bwd", base, base + 4 * C, base + 8 * C, 3 * C, _p(bias), _p(mask), _p(dout), C,
                  gb, gb + 4 * C, gb + 8 * C, 3 * C, _p(dpart), B_, H, nW, d, scale, _stream())
        dtable = torch.empty((225, H), device=qkv.device, dtype=torch.float32)
        _lib.call("dhz_bias_table_grad", _p(dpart), parts, _p(dtable), H, 0, _stream())
        return dqkv, dtable, None, None, None, None


def dense_window_attention(qkv, table, mask, H, d, scale):
    return _DenseWindowAttention.apply(qkv, table, mask, H, d, scale)


def shift_mask(Hres, Wres, shift, device):
    """[nW,64,64] 0/-100 mask of M1:803-836 (cached by callers; depends only on the geometry)."""
    m = torch.empty(((Hres // 8) * (Wres // 8), NTOK, NTOK), device=device, dtype=torch.float32)
    _lib.call("dhz_shift_mask", _p(m), Hres, Wres, shift, _stream())
    return m


# ----------------------------------------------------------------------------- token-major Linear (K2 / K4 / K5 GEMMs)
# Called with every parameter whose gradient has just been accumulated IN PLACE by a wgrad kernel (autograd's
# AccumulateGrad - and therefore its post-accumulate hooks - is bypassed for those); the gradient reducer
# installs itself here to keep its bucket bookkeeping.
GRAD_READY = None
# [zeroed fp32 tensor, bump offset, weakref(owner)]: FlatAdamW.zero_grad() zeroes it together with the flat gradient buffer and resets the
# offset; zeros_f32 carves accumulation targets of the backward pass out of it (valid until the next zero_grad)
ZERO_SCRATCH = None


def zeros_f32(shape, device):
    """torch.zeros(shape, fp32) - from the optimizer's pre-zeroed scratch region when one is registered for this device and has room
    (no fill launch), else a fresh allocation.  Only for temporaries of ONE backward pass."""
    n = 1
    for d in shape:
        n *= int(d)
    zs = ZERO_SCRATCH
    if zs is not None and zs[2]() is not None and zs[0].device == device:
        off = (zs[1] + 7) // 8 * 8
        if off + n <= zs[0].numel():
            zs[1] = off + n
            return zs[0][off: off + n].view(shape)
    return torch.zeros(shape, device=device, dtype=torch.float32)


def _accumulate_param_grads(dy, ldy_off, x, params, row_scale=None):
    """dW += dy[:, off:off+N]^T x, db += colsum for every (W, b) pair, straight into .grad (zero-init).  Pairs of equal
    shape (the Q / K / V projections) go through ONE launch that reads x once (dhz_linear_wgrad_multi).
    row_scale = (scale[B], rows_per_scale): row t of dy counts as scale[t // rows_per_scale] * dy[t] (fp32, one pair only)."""
    T, K = x.shape
    for W, b in params:
        for p in (W, b):
            if p is not None and p.grad is None:
                p.grad = torch.zeros_like(p, memory_format=torch.contiguous_format)
        assert W.grad.is_contiguous()
    N = params[0][0].shape[0]
    same = 1 < len(params) <= 4 and all(W.shape[0] == N for W, _ in params) and \
        len({b is None for _, b in params}) == 1
    assert row_scale is None or (dy.dtype == torch.float32 and len(params) == 1)
    if dy.dtype == BF16:
        # bf16 dy / x, fp32 accumulation straight into the fp32 .grad buffers; equal-shaped parameters share one launch
        groups = [params] if (same or len(params) == 1) else [[pr] for pr in params]
        off = ldy_off
        for grp in groups:
            n = len(grp)
            Ng = grp[0][0].shape[0]
            dws = (ctypes.c_void_p * n)(*[W.grad.data_ptr() for W, _ in grp])
            dbs = (ctypes.c_void_p * n)(*[(b.grad.data_ptr() if b is not None else None) for _, b in grp])
            _lib.call("dhz_linear_wgrad_bf16", dy.data_ptr() + 2 * off, dy.stride(0), _p(x), x.stride(0), T, n, Ng, K,
                      ctypes.cast(dws, ctypes.c_void_p), ctypes.cast(dbs, ctypes.c_void_p), _stream())
            off += n * Ng
    elif SPLIT_BF16 and T % 64 == 0 and K % 64 == 0 and all(W.shape[0] % 64 == 0 for W, _ in params):
        # the default fp32-class path (SPLIT_BF16 == 6): contraction over T on the bf16 pipe with split operands (csrc/linear_split.hip)
        groups = [params] if (same or len(params) == 1) else [[pr] for pr in params]
        off = ldy_off
        for grp in groups:
            n = len(grp)
            Ng = grp[0][0].shape[0]
            dws = (ctypes.c_void_p * n)(*[W.grad.data_ptr() for W, _ in grp])
            dbs = (ctypes.c_void_p * n)(*[(b.grad.data_ptr() if b is not None else None) for _, b in grp])
            ev = _timed("dhz_linear_wgrad_split")
            _lib.call("dhz_linear_wgrad_split", dy.data_ptr() + 4 * off, dy.stride(0), _p(x), x.stride(0), T, n, Ng, K,
                      ctypes.cast(dws, ctypes.c_void_p), ctypes.cast(dbs, ctypes.c_void_p),
                      _p(row_scale[0]) if row_scale is not None else None, int(row_scale[1]) if row_scale is not None else 0,
                      _terms(), _stream())
            _timed_end(ev, 2.0 * _terms() * T * n * Ng * K)
            off += n * Ng
    elif same:
        n = len(params)
        dws = (ctypes.c_void_p * n)(*[W.grad.data_ptr() for W, _ in params])
        dbs = (ctypes.c_void_p * n)(*[(b.grad.data_ptr() if b is not None else None) for _, b in params])
        _lib.call("dhz_linear_wgrad_multi", dy.data_ptr() + 4 * ldy_off, dy.stride(0), _p(x), x.stride(0), T, n, N, K,
                  ctypes.cast(dws, ctypes.c_void_p), ctypes.cast(dbs, ctypes.c_void_p), _stream())
    else:
        off = ldy_off
        for W, b in params:
            if row_scale is not None:
                _lib.call("dhz_linear_wgrad_rs", dy.data_ptr() + 4 * off, dy.stride(0), _p(x), x.stride(0), T, W.shape[0], K,
                          _p(W.grad), _p(b.grad) if b is not None else None, _p(row_scale[0]), int(row_scale[1]), _stream())
            else:
                _lib.call("dhz_linear_wgrad", dy.data_ptr() + 4 * off, dy.stride(0), _p(x), x.stride(0), T, W.shape[0], K,
                          _p(W.grad), _p(b.grad) if b is not None else None, _stream())
            off += W.shape[0]
    if GRAD_READY is not None:
        for W, b in params:
            GRAD_READY(W)
            if b is not None:
                GRAD_READY(b)


def wgrad_into(dy, off, x, N, dw, db, row_scale=None):
    """dw[N,K] += dy[:, off:off+N]^T x, db += column sums (fp32 accumulators) for fp32 or bf16 dy / x."""
    T, K = x.shape
    if SPLIT_BF16 and dy.dtype == torch.float32 and T % 64 == 0 and K % 64 == 0 and N % 64 == 0:
        dws = (ctypes.c_void_p * 1)(dw.data_ptr())
        dbs = (ctypes.c_void_p * 1)(db.data_ptr() if db is not None else None)
        _lib.call("dhz_linear_wgrad_split", dy.data_ptr() + 4 * off, dy.stride(0), _p(x), x.stride(0), T, 1, N, K,
                  ctypes.cast(dws, ctypes.c_void_p), ctypes.cast(dbs, ctypes.c_void_p),
                  _p(row_scale[0]) if row_scale is not None else None, int(row_scale[1]) if row_scale is not None else 0, _terms(),
                  _stream())
        return
    if row_scale is not None:
        assert dy.dtype == torch.float32
        _lib.call("dhz_linear_wgrad_rs", dy.data_ptr() + 4 * off, dy.stride(0), _p(x), x.stride(0), T, N, K, _p(dw), _p(db),
                  _p(row_scale[0]), int(row_scale[1]), _stream())
        return
    if dy.dtype == BF16:
        dws = (ctypes.c_void_p * 1)(dw.data_ptr())
        dbs = (ctypes.c_void_p * 1)(db.data_ptr() if db is not None else None)
        _lib.call("dhz_linear_wgrad_bf16", dy.data_ptr() + 2 * off, dy.stride(0), _p(x), x.stride(0), T, 1, N, K,
                  ctypes.cast(dws, ctypes.c_void_p), ctypes.cast(dbs, ctypes.c_void_p), _stream())
    else:
        _lib.call("dhz_linear_wgrad", dy.data_ptr() + 4 * off, dy.stride(0), _p(x), x.stride(0), T, N, K, _p(dw), _p(db), _stream())


_NO_CAT_VIEW = bool(__import__("os").environ.get("DHZ_NO_CAT_VIEW"))     # A/B switch: always copy


def cat_rows(ts):
    """torch.cat(ts, 0) for detached row blocks - WITHOUT a copy when they already sit back to back in memory, which is how
    FlatAdamW lays out the Q / K / V weights (and biases) of an attention layer in its flat parameter buffer.  Only for use
    where autograd does not track the result (inside Function.forward / backward)."""
    t0 = ts[0]
    end = t0.data_ptr() + t0.numel() * t0.element_size()
    base = t0.untyped_storage().data_ptr()          # same allocation, not merely neighbouring ones
    ok = t0.is_contiguous() and not t0.requires_grad and not _NO_CAT_VIEW
    for t in ts[1:]:
        ok = ok and t.is_contiguous() and not t.requires_grad and t.dtype == t0.dtype and t.device == t0.device \
            and t.shape[1:] == t0.shape[1:] and t.data_ptr() == end and t.untyped_storage().data_ptr() == base
        if not ok:
            break
        end += t.numel() * t.element_size()
    if not ok:
        return torch.cat(list(ts), 0)
    rows = sum(t.shape[0] for t in ts)
    shape = (rows,) + tuple(t0.shape[1:])
    return t0.as_strided(shape, t0.stride())


class _LinearTokens(Function):
    """y = x [W_1;..;W_n]^T + [b_1;..;b_n] for token-major x [T,K]: dhz_linear_fwd forward, dhz_linear_dgrad for the
    input gradient, dhz_linear_wgrad for the weight/bias gradients, which are accumulated in place into the parameters'
    .grad (the optimizer's flat gradient buffer)."""

    @staticmethod
    def forward(ctx, x, *wb):
        _require_gpu(x)
        params = [(wb[i], wb[i + 1]) for i in range(0, len(wb), 2)]
        if len(params) == 1:
            W, b = params[0]
        else:
            W = cat_rows([w.detach() for w, _ in params])
            b = cat_rows([b_.detach() for _, b_ in params])
        if x.dtype == BF16:
            W = bf16_copy(W)                # one cast per Linear and step: the copy is saved for the backward-data GEMM
        y = gemm_fwd(x, W, b)
        ctx.save_for_backward(x, W)
        ctx.params = params
        return y

    @staticmethod
    def backward(ctx, dy):
        x, W = ctx.saved_tensors
        dy = dy.contiguous()
        dx = gemm_dgrad(dy, W) if ctx.needs_input_grad[0] else None
        T, K = x.shape
        grads = []
        off = 0
        q = 64 if dy.dtype == BF16 else 16          # fp32: 16-wide tile forms exist for the embed_dim = 16 model (csrc/linear_wgrad.hip)
        tq = 64 if dy.dtype == BF16 else 32         # token rows per stage
        if T % tq == 0 and K % q == 0 and all(w.shape[0] % q == 0 and w.is_leaf and w.requires_grad
                                               and (b is None or (b.is_leaf and b.requires_grad)) for w, b in ctx.params):
            _accumulate_param_grads(dy, 0, x, ctx.params)           # one launch for equal-shaped parameters (Q / K / V)
            return (dx,) + (None, None) * len(ctx.params)
        for w, b in ctx.params:
            N = w.shape[0]
            # measured on MI355X (tools/bench_wgrad.py): the split-T kernel wins 2-18x for T >= 16k tokens; on the deep
            # stages (T <= 8k) it is within 0.9-1.2x of the library's TN GEMM and delivers the bias gradient for free
            # (the library path pays a separate ~20 us column-sum kernel), so it is used everywhere
            mine = T % tq == 0 and N % q == 0 and K % q == 0        # (the kernel's shape contract; always true on this model)
            if not w.requires_grad and (b is None or not b.requires_grad):
                grads += [None, None]                                 # frozen Linear
            elif mine and w.is_leaf and w.requires_grad and (b is None or (b.is_leaf and b.requires_grad)):
                _accumulate_param_grads(dy, off, x, [(w, b)])
                grads += [None, None]
            elif mine:
                fp = w.dtype == torch.float32           # (accumulation targets of this backward pass: from the optimizer's zeroed scratch, no fill launch)
                dw = zeros_f32(tuple(w.shape), w.device) if fp else torch.zeros_like(w, memory_format=torch.contiguous_format)
                db = (zeros_f32(tuple(b.shape), b.device) if b.dtype == torch.float32 else torch.zeros_like(b)) if b is not None else None
                wgrad_into(dy, off, x, N, dw, db)
                grads += [dw, db]
            else:
                raise RuntimeError(f"dehaze_hip: Linear weight gradient for T={T}, N={N}, K={K}: the HIP kernel needs "
                                   "multiples of 16 in fp32 and of 64 in bf16 (there is deliberately no library fallback)")
            off += N
        return (dx,) + tuple(grads)


class _GeluTokens(Function):
    """GELU between the two Linears of Mlp (M1:460-461) on dhz_gelu_fwd / dhz_gelu_bwd."""

    @staticmethod
    def forward(ctx, u):
        _require_gpu(u)
        u = u.contiguous()
        y = torch.empty_like(u)
        _lib.call("dhz_gelu_fwd_dt", _p(u), _p(y), u.numel(), _dt(u), _stream())
        ctx.save_for_backward(u)
        return y

    @staticmethod
    def backward(ctx, dy):
        (u,) = ctx.saved_tensors
        dy = dy.contiguous()
        du = torch.empty_like(u)
        _lib.call("dhz_gelu_bwd_dt", _p(dy), _p(u), _p(du), u.numel(), None, 0, _dt(u), _stream())
        return du


def gelu_tokens(u):
    return _GeluTokens.apply(u)


def linear_tokens(x, *wb):
    """x [T,K] (contiguous); wb = W1, b1, W2, b2, ...  ->  [T, sum N_i]."""
    if not torch.is_grad_enabled() or not any(t is not None and t.requires_grad for t in (x,) + wb):
        params = [(wb[i], wb[i + 1]) for i in range(0, len(wb), 2)]
        W = params[0][0] if len(params) == 1 else torch.cat([w for w, _ in params], 0)
        b = params[0][1] if len(params) == 1 else torch.cat([b_ for _, b_ in params], 0)
        return gemm_fwd(x.contiguous(), W, b)
    return _LinearTokens.apply(x.contiguous(), *wb)


# ----------------------------------------------------------------------------- K1
class _LNPartition(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, Hres, Wres, shift, partition):
        _require_gpu(x, gamma, beta)
        x = x.contiguous()
        B, L, C = x.shape
        assert L == Hres * Wres
        y = torch.empty((B * L, C), device=x.device, dtype=x.dtype)
        stats = torch.empty((B * L, 2), device=x.device, dtype=torch.float32)
        _lib.call("dhz_ln_partition_fwd_dt", _p(x), _p(gamma), _p(beta), _p(y), _p(stats), B, Hres, Wres, C, shift,
                  int(partition), _dt(x), _stream())
        ctx.save_for_backward(x, gamma, stats)
        ctx.geom = (B, Hres, Wres, C, shift, int(partition))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, stats = ctx.saved_tensors
        B, Hres, Wres, C, shift, partition = ctx.geom
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        dgb = torch.zeros((2, C), device=x.device, dtype=torch.float32)
        _lib.call("dhz_ln_partition_bwd_dt", _p(dy), _p(x), _p(gamma), _p(stats), None, _p(dx), dgb[0].data_ptr(),
                  dgb[1].data_ptr(), B, Hres, Wres, C, shift, partition, _dt(x), _stream())
        return dx, dgb[0], dgb[1], None, None, None, None


def ln_partition(x, gamma, beta, Hres, Wres, shift):
    """LayerNorm -> roll(-shift) -> window_partition  (M1:839-852).  [B,L,C] -> [B*nW*64, C]."""
    return _LNPartition.apply(x, gamma, beta, Hres, Wres, shift, True)


def layer_norm_tokens(x, gamma, beta):
    """Plain LayerNorm over the last dim, tokens stay in place (norm2, M1:873). [B,L,C] -> [B*L, C]."""
    return _LNPartition.apply(x, gamma, beta, x.shape[1], 1, 0, False)


# ----------------------------------------------------------------------------- K4 tail
class _ReverseResidual(Function):
    @staticmethod
    def forward(ctx, yw, shortcut, scale, Hres, Wres, shift, partition):
        _require_gpu(yw, shortcut, scale)
        shortcut = shortcut.contiguous()
        yw = yw.contiguous()
        B, L, C = shortcut.shape
        out = torch.empty_like(shortcut)
        _lib.call("dhz_reverse_residual_fwd_dt", _p(yw), _p(shortcut), _p(scale), _p(out), B, Hres, Wres, C, shift,
                  int(partition), _dt(shortcut), _stream())
        ctx.save_for_backward(scale)
        ctx.geom = (B, Hres, Wres, C, shift, int(partition), tuple(yw.shape))
        return out

    @staticmethod
    def backward(ctx, dout):
        (scale,) = ctx.saved_tensors
        B, Hres, Wres, C, shift, partition, yshape = ctx.geom
        dout = dout.contiguous()
        dyw = torch.empty(yshape, device=dout.device, dtype=dout.dtype)
        _lib.call("dhz_reverse_residual_bwd_dt", _p(dout), _p(scale), _p(dyw), B, Hres, Wres, C, shift, partition,
                  _dt(dout), _stream())
        return dyw, dout, None, None, None, None, None


def reverse_residual(yw, shortcut, scale, Hres, Wres, shift):
    """window_reverse -> roll(+shift) -> shortcut + drop_path(.)  (M1:859-872)."""
    return _ReverseResidual.apply(yw, shortcut, scale, Hres, Wres, shift, True)


def residual_scale(y, shortcut, scale):
    """shortcut + scale[b] * y with y already in token order (M1:873)."""
    return _ReverseResidual.apply(y, shortcut, scale, shortcut.shape[1], 1, 0, False)


# ----------------------------------------------------------------------------- K5 middle
class _LeffDwconv(Function):
    @staticmethod
    def forward(ctx, u, w, b, Hres, Wres):
        _require_gpu(u, w, b)
        u = u.contiguous()
        B, L, Ch = u.shape
        assert L == Hres * Wres
        z = torch.empty_like(u)
        keep = any(ctx.needs_input_grad)
        t = torch.empty_like(u) if keep else None
        w = w.contiguous()
        _lib.call("dhz_leff_dwconv_fwd_dt", _p(u), _p(w), _p(b), _p(t), _p(z), B, Hres, Wres, Ch, _dt(u), _stream())
        if keep:
            ctx.save_for_backward(u, t, w)
        ctx.geom = (B, Hres, Wres, Ch)
        return z

    @staticmethod
    def backward(ctx, dz):
        u, t, w = ctx.saved_tensors
        B, Hres, Wres, Ch = ctx.geom
        dz = dz.contiguous()
        du = torch.empty_like(u)
        dwb = torch.zeros((Ch * 10,), device=u.device, dtype=torch.float32)
        _lib.call("dhz_leff_dwconv_bwd_dt", _p(dz), _p(u), _p(t), _p(w), _p(du), dwb.data_ptr(),
                  dwb.data_ptr() + 4 * Ch * 9, B, Hres, Wres, Ch, _dt(u), _stream())
        return du, dwb[:Ch * 9].view(Ch, 1, 3, 3), dwb[Ch * 9:], None, None


def leff_dwconv(u, w, b, Hres, Wres):
    """gelu(dwconv3x3(gelu(u)) + b) in token layout (M1:488, 514-520)."""
    return _LeffDwconv.apply(u, w, b, Hres, Wres)


# ----------------------------------------------------------------------------- K10
class _CharbonnierClamped(Function):
    @staticmethod
    def forward(ctx, x, y, eps, clamp01):
        _require_gpu(x, y)
        x = x.contiguous()
        y = y.contiguous()
        n = x.numel()
        acc = zeros_f32((), x.device)
        clamped = torch.empty_like(x) if clamp01 else None
        _lib.call("dhz_charbonnier_fwd", _p(x), _p(y), _p(clamped), _p(acc), n, float(eps), int(clamp01), _stream())
        ctx.save_for_backward(x, y)
        ctx.eps, ctx.clamp01 = float(eps), int(clamp01)
        if not clamp01:
            clamped = x.new_empty(0)
            ctx.mark_non_differentiable(clamped)
        return acc / n, clamped

    @staticmethod
    def backward(ctx, gloss, gclamp):
        x, y = ctx.saved_tensors
        dx = torch.empty_like(x)
        if gloss is None:
            gloss = torch.zeros((), device=x.device, dtype=torch.float32)
        gloss = gloss.contiguous().float()
        gclamp = gclamp.contiguous() if (gclamp is not None and ctx.clamp01) else None
        _lib.call("dhz_charbonnier_bwd", _p(x), _p(y), _p(gloss), _p(gclamp), _p(dx), x.numel(), ctx.eps,
                  1.0 / x.numel(), ctx.clamp01, _stream())
        return dx, None, None, None


def charbonnier_clamped(x, y, eps=1e-3):
    """(mean(sqrt((clamp(x,0,1)-y)^2 + eps^2)), clamp(x,0,1))  - TR:230 + losses.py:48-52 in one pass."""
    return _CharbonnierClamped.apply(x, y, eps, True)


def charbonnier(x, y, eps=1e-3):
    """CharbonnierLoss.forward(x, y) - losses.py:48-52 (no clamp)."""
    return _CharbonnierClamped.apply(x, y, eps, False)[0]


# ----------------------------------------------------------------------------- K12
def adamw_step_(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0, p16=None):
    """In-place AdamW over flat fp32 buffers (torch.optim.AdamW semantics, TR:90-92); p16: optional bf16 buffer that receives
    the updated parameters in the same pass (the weight copy of the bf16 GEMMs)."""
    _require_gpu(p, g, m, v)
    assert p.is_contiguous() and g.is_contiguous() and m.is_contiguous() and v.is_contiguous()
    assert p16 is None or (p16.dtype == BF16 and p16.is_contiguous() and p16.numel() == p.numel())
    _lib.call("dhz_adamw_step_shadow", _p(p), _p(g), _p(m), _p(v), _p(p16), p.numel(), float(lr), float(beta1), float(beta2),
              float(eps), float(weight_decay), int(step), float(grad_scale), _stream())


# ----------------------------------------------------------------------------- K9b
class _ThinConv(Function):
    """Output projection: tokens [B, H*W, C] -> Conv2d(C, 3, 3x3, pad 1) -> [B, 3, H, W]; weight and bias gradients are
    accumulated in place when the parameters are leaves."""

    @staticmethod
    def forward(ctx, x, w, b, H, W):
        _require_gpu(x, w)
        x = x.contiguous()
        B, L, C = x.shape
        assert L == H * W and w.shape == (3, C, 3, 3)
        wc = w.contiguous()
        y = torch.empty((B, 3, H, W), device=x.device, dtype=torch.float32)
        _lib.call("dhz_thin_conv3x3_fwd_dt", _p(x), _p(wc), _p(b), _p(y), B, H, W, C, _dt(x), _stream())
        ctx.save_for_backward(x, wc)
        ctx.params, ctx.geom = (w, b), (B, H, W, C)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wc = ctx.saved_tensors
        w, b = ctx.params
        B, H, W, C = ctx.geom
        dy = dy.contiguous().float()
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            _lib.call("dhz_thin_conv3x3_dgrad_dt", _p(dy), _p(wc), _p(dx), B, H, W, C, _dt(x), _stream())
        inplace = w.is_leaf and w.requires_grad and (b is None or (b.is_leaf and b.requires_grad))
        if inplace:
            for p_ in (w, b):
                if p_ is not None and p_.grad is None:
                    p_.grad = torch.zeros_like(p_, memory_format=torch.contiguous_format)
            inplace = w.grad.is_contiguous()
        if inplace:
            _lib.call("dhz_thin_conv3x3_wgrad_dt", _p(dy), _p(x), _p(w.grad), _p(b.grad) if b is not None else None, B, H, W, C,
                      _dt(x), _stream())
            if GRAD_READY is not None:
                GRAD_READY(w)
                if b is not None:
                    GRAD_READY(b)
            return dx, None, None, None, None
        dw = torch.zeros((3, C, 3, 3), device=x.device, dtype=torch.float32)
        db = torch.zeros(3, device=x.device, dtype=torch.float32) if b is not None else None
        _lib.call("dhz_thin_conv3x3_wgrad_dt", _p(dy), _p(x), _p(dw), _p(db), B, H, W, C, _dt(x), _stream())
        return dx, dw, db, None, None


def thin_conv3x3(x, w, b, H, W):
    return _ThinConv.apply(x, w, b, H, W)


# ----------------------------------------------------------------------------- K8: 4x4 / stride-2 down-sampling convolution
def conv4s2_supported(x, H, W, need_grad):
    """Shapes the implicit-GEMM kernels take (csrc/conv_gemm.hip): fp32 tokens, even map, channels multiples of 32; the weight
    gradient additionally wants power-of-two output maps (the training patch sizes 128 / 256)."""
    Cin = x.shape[-1]
    ok = x.is_cuda and x.dtype == torch.float32 and H % 2 == 0 and W % 2 == 0 and Cin % 32 == 0
    if ok and need_grad:
        Ho, Wo = H // 2, W // 2
        ok = (Ho & (Ho - 1)) == 0 and (Wo & (Wo - 1)) == 0 and (x.shape[0] * Ho * Wo) % 32 == 0
    return ok


class _Conv4s2(Function):
    """Downsample.conv on the token layout (M1:606-622): x [B, H*W, Cin] -> [B, (H/2)*(W/2), Cout]."""

    @staticmethod
    def forward(ctx, x, w, b, H, W):
        _require_gpu(x, w, b)
        x = x.contiguous()
        B, L, Cin = x.shape
        Cout = w.shape[0]
        assert L == H * W and tuple(w.shape) == (Cout, Cin, 4, 4)
        wp = w.detach().permute(0, 2, 3, 1).reshape(Cout, 16 * Cin).contiguous()      # [co][(ky, kx, ci)]
        y = torch.empty((B, L // 4, Cout), device=x.device, dtype=torch.float32)
        _lib.call("dhz_conv4s2_fwd", _p(x), _p(wp), _p(b), _p(y), B, H, W, Cin, Cout, _stream())
        ctx.save_for_backward(x, w)
        ctx.params, ctx.geom = (w, b), (B, H, W, Cin, Cout)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w_ = ctx.saved_tensors
        w, b = ctx.params
        B, H, W, Cin, Cout = ctx.geom
        dy = dy.contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            wq = w_.detach().permute(2, 3, 0, 1).contiguous()                         # [(ky, kx, co)][ci]
            dx = torch.empty_like(x)
            _lib.call("dhz_conv4s2_dgrad", _p(dy), _p(wq), _p(dx), B, H, W, Cin, Cout, _stream())
        gw = gb = None
        if w.requires_grad or (b is not None and b.requires_grad):
            dwp = zeros_f32((Cout, 16 * Cin), x.device)
            inplace_b = b is not None and b.is_leaf and b.requires_grad
            if inplace_b and b.grad is None:
                b.grad = torch.zeros_like(b)
            dbv = b.grad if inplace_b else (torch.zeros_like(b) if b is not None else None)
            _lib.call("dhz_conv4s2_wgrad", _p(dy), _p(x), _p(dwp), _p(dbv), B, H, W, Cin, Cout, _stream())
            dw = dwp.view(Cout, 4, 4, Cin).permute(0, 3, 1, 2)                        # back to [co][ci][ky][kx]
            if w.is_leaf and w.requires_grad:
                if w.grad is None:
                    w.grad = torch.zeros_like(w, memory_format=torch.contiguous_format)
                w.grad.add_(dw)
                if GRAD_READY is not None:
                    GRAD_READY(w)
            else:
                gw = dw.contiguous()
            if inplace_b:
                if GRAD_READY is not None:
                    GRAD_READY(b)
            else:
                gb = dbv
        return dx, gw, gb, None, None


class _Conv4s2BF16(Function):
    """Downsample.conv for bf16 tokens (BASELINE config 4): an explicit tap-major patch matrix (streaming copy in this layout)
    and the three bf16-MFMA token-Linear GEMMs over it (csrc/conv_bf16.hip); fp32 master weights / gradients."""

    @staticmethod
    def forward(ctx, x, w, b, H, W):
        _require_gpu(x, w, b)
        x = x.contiguous()
        B, L, Cin = x.shape
        Cout = w.shape[0]
        assert L == H * W and tuple(w.shape) == (Cout, Cin, 4, 4) and x.dtype == BF16
        wp = w.detach().permute(0, 2, 3, 1).reshape(Cout, 16 * Cin).to(BF16)          # [co][(ky, kx, ci)], one cast per pass
        col = torch.empty((B * (L // 4), 16 * Cin), device=x.device, dtype=BF16)
        _lib.call("dhz_im2col_k4s2_bf16", _p(x), _p(col), B, H, W, Cin, _stream())
        y = gemm_fwd(col, wp, b.detach() if b is not None else None)
        ctx.save_for_backward(col, wp)
        ctx.params, ctx.geom = (w, b), (B, H, W, Cin, Cout)
        return y.view(B, L // 4, Cout)

    @staticmethod
    def backward(ctx, dy):
        col, wp = ctx.saved_tensors
        w, b = ctx.params
        B, H, W, Cin, Cout = ctx.geom
        dy = dy.contiguous().view(-1, Cout)
        dx = None
        if ctx.needs_input_grad[0]:
            dcol = gemm_dgrad(dy, wp)
            dx = torch.empty((B, H * W, Cin), device=dy.device, dtype=BF16)
            _lib.call("dhz_col2im_k4s2_bf16", _p(dcol), _p(dx), B, H, W, Cin, _stream())
        gw = gb = None
        if w.requires_grad or (b is not None and b.requires_grad):
            dwp = torch.zeros((Cout, 16 * Cin), device=dy.device, dtype=torch.float32)
            inplace_b = b is not None and b.is_leaf and b.requires_grad
            if inplace_b and b.grad is None:
                b.grad = torch.zeros_like(b)
            dbv = b.grad if inplace_b else (torch.zeros_like(b) if b is not None else None)
            wgrad_into(dy, 0, col, Cout, dwp, dbv)
            dw = dwp.view(Cout, 4, 4, Cin).permute(0, 3, 1, 2)                        # back to [co][ci][ky][kx]
            if w.is_leaf and w.requires_grad:
                if w.grad is None:
                    w.grad = torch.zeros_like(w, memory_format=torch.contiguous_format)
                w.grad.add_(dw)
                if GRAD_READY is not None:
                    GRAD_READY(w)
            else:
                gw = dw.contiguous()
            if inplace_b:
                if GRAD_READY is not None:
                    GRAD_READY(b)
            else:
                gb = dbv
        return dx, gw, gb, None, None


def conv4s2_bf16_supported(x, H, W):
    """bf16 tokens, even map, channels in 64s (the bf16 GEMM kernels' tiles), output tokens in 64s."""
    Cin = x.shape[-1]
    return x.is_cuda and x.dtype == BF16 and H % 2 == 0 and W % 2 == 0 and Cin % 64 == 0 and (x.shape[0] * (H // 2) * (W // 2)) % 64 == 0


def conv4s2_tokens(x, w, b, H, W):
    if x.dtype == BF16:
        return _Conv4s2BF16.apply(x, w, b, H, W)
    return _Conv4s2.apply(x, w, b, H, W)


# ----------------------------------------------------------------------------- K9: input projection
class _InputProj(Function):
    """Conv2d(3, E, 3x3, pad 1) + LeakyReLU from the NCHW image into tokens [B, H*W, E] (M1:659-682)."""

    @staticmethod
    def forward(ctx, img, w, b, slope, out_dtype=torch.float32):
        _require_gpu(img, w, b)
        img = img.contiguous()
        B, _, H, W = img.shape
        E = w.shape[0]
        y = torch.empty((B, H * W, E), device=img.device, dtype=out_dtype)
        _lib.call("dhz_input_proj_fwd_dt", _p(img), _p(w.contiguous()), _p(b), _p(y), B, H, W, E, float(slope), _dt(y), _stream())
        ctx.save_for_backward(img, y)
        ctx.params, ctx.slope = (w, b), float(slope)
        return y

    @staticmethod
    def backward(ctx, dy):
        img, y = ctx.saved_tensors
        w, b = ctx.params
        if ctx.needs_input_grad[0]:
            raise RuntimeError("dehaze_hip: InputProj has no backward-data kernel (the image never needs a gradient on this path)")
        B, _, H, W = img.shape
        E = w.shape[0]
        inplace = w.is_leaf and w.requires_grad and b.is_leaf and b.requires_grad
        if inplace:
            for p_ in (w, b):
                if p_.grad is None:
                    p_.grad = torch.zeros_like(p_, memory_format=torch.contiguous_format)
            inplace = w.grad.is_contiguous()
        gw = w.grad if inplace else torch.zeros_like(w, memory_format=torch.contiguous_format)
        gb = b.grad if inplace else torch.zeros_like(b)
        _lib.call("dhz_input_proj_bwd_dt", _p(dy.contiguous().to(y.dtype)), _p(y), _p(img), _p(gw), _p(gb), B, H, W, E, ctx.slope, _dt(y),
                  _stream())
        if inplace:
            if GRAD_READY is not None:
                GRAD_READY(w)
                GRAD_READY(b)
            return None, None, None, None, None
        return None, gw, gb, None, None


def input_proj(img, w, b, slope=0.01, out_dtype=torch.float32):
    return _InputProj.apply(img, w, b, slope, out_dtype)
