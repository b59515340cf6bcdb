"""The two branches of a LeWin block (M1:839-873) as hand-sequenced kernel chains behind autograd nodes.

attention branch : out = x + drop_scale * OutProj(ProbAttn(QKV(partition(roll(LN(x))))))           (M1:839-872)
    forward  : ONE kernel (dhz_fused_window_attn_fwd) where it wins, else dhz_ln_partition_fwd -> QKV GEMM -> dhz_ps_attn_fwd ->
               out-projection GEMM whose EPILOGUE is window reverse + un-roll + DropPath factor + residual (ops.gemm_fwd_res)
    backward : out-proj dgrad + wgrad -> dhz_ps_attn_bwd (+ bias table gradient) -> QKV dgrad + wgrad -> dhz_ln_partition_bwd with
               the shortcut gradient folded in (dx = dout + dLN) - no autograd-side accumulation kernels at all
LeFF / Mlp branch: out = x + drop_scale * linear2(gelu(dwconv(gelu(linear1(LN(x))))))                (M1:873)
whole block      : block() runs both branches as ONE node, so that the gradient between them can live in the layout its consumer
                   reads: the LeFF branch's LayerNorm backward writes d(x1) in the attention branch's WINDOW order
                   (dhz_ln_partition_bwd_lay), the out-projection's backward products read it as it lies (the DropPath factor as a
                   row factor inside them), the attention LayerNorm backward reads its residual gradient in the same order -
                   dhz_reverse_residual_bwd never runs.
Each branch's forward / backward is a plain function over a record (`_Rec`); the autograd nodes only store and replay records.
"""
import torch
from torch.autograd import Function

from . import _lib, ops
from .ops import NTOK, _p, _require_gpu, _stream

SUPPORTED_C = (32, 64, 128)
ENABLED = True      # set False to force the unfused chain (tests compare the two)
ATTN_FUSED_C128_MAX_HW = int(__import__("os").environ.get("DHZ_FUSED_C128_MAX_HW", "1024"))   # C = 128 takes the fused attention forward up to this map size (32 x 32), the chain above
# Fused LeFF kernels (csrc/leff_fused.hip).  Measured on MI355X (tools/bench_leff.py, bs 32): the fused forward wins at C = 32 / 64
# (inference 0.76-0.85x of the chain, training 0.87-0.95x) and loses at C = 128; a fused backward-data kernel (round 2, removed
# in round 3) was 1.5-2x slower than the kernel chain in fp32 - fp32-input MFMA and fp32 VALU instructions share the SIMD's
# issue (profiles/r02_coexec_micro.txt), so fusing the GELU / depthwise VALU work into the GEMM kernels ADDS its time to the
# matrix time instead of hiding it behind HBM traffic as the stand-alone streaming kernels do.
# Fused attention BACKWARD (csrc/fused_attn_bwd.hip, C = 32): the forward then saves only the selection ranks.  A/B against the
# backward kernel chain: tools/bench_fused.py, profiles/r03_fused_attn_bwd_ab.txt.
ATTN_FUSED_BWD_C = (32,)
# Six-term QKV projections inside the fused forward (csrc/fused_attn.hip, P6): weight planes brought once per workgroup by LDS-DMA into
# the dead Q / K / V / S tiles.  DHZ_FUSED_P6=0 selects the fp32-pipe projections (A/B: tools/bench_fused.py).
ATTN_FUSED_P6 = __import__("os").environ.get("DHZ_FUSED_P6", "1") != "0"
# C = 32 (planes in registers of the persistent workgroups) exists and is tested, but is not dispatched: its extra 50 registers cost the third
# workgroup per CU (tools/bench_fused_p6.py: 101.5 -> 100.1 us training, 100.0 -> 96.0 inference at 128 x 128)
ATTN_FUSED_P6_C = (64, 128)
LEFF_FUSED = True           # False forces the kernel chain everywhere
LEFF_FUSED_C = (32, 64)     # widths that take the fused forward
# ... C = 64 only below this many tokens when the chain's GEMMs run in the six-term form (tools/bench_leff.py, bs 32, forward with the
# training saves: 64 x 64 maps 204 us fused / 218 chain, 128 x 128 maps 794 / 765; forward + backward 2058 / 2010)
LEFF_FUSED_C64_MAX_T = 262144
# the fused LeFF forward with its two weight products six-term on the bf16 matrix pipe (csrc/leff_fused.hip, L6); DHZ_LEFF_P6=0: fp32 pipe
LEFF_FUSED_P6 = __import__("os").environ.get("DHZ_LEFF_P6", "1") != "0"
# measured (tools/bench_leff_p6.py, bs 32, forward with the training saves / inference, us): C = 64 at 64 x 64: 218.7 / 183.5 -> 196.6 / 168.1;
# C = 64 at 128 x 128: 798.8 / 641.9 -> 741.5 / 625.3 (the kernel chain: 745.4 / 632.0 - the dispatch there does not change); C = 32 at
# 128 x 128: 367.6 / 264.0 -> 393.4 / 308.8, SLOWER: at C = 32 the matrix share of this kernel is a fifth of its time and the piece images
# cost more vector / LDS work than the shorter MFMAs return - the fp32-pipe instance stays there
LEFF_FUSED_P6_C = (64,)


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
        fp = w.dtype == torch.float32
        dw = ops.zeros_f32(tuple(w.shape), w.device) if fp else torch.zeros_like(w, memory_format=torch.contiguous_format)
        db = (ops.zeros_f32(tuple(b.shape), b.device) if fp else torch.zeros_like(b)) if b is not None else None
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


def _ln_backward(dxn, x, gamma_p, beta_p, gamma, stats, dres, B, Hres, Wres, C, shift, partition, dres_windowed=False, dx_window=None,
                 dx2=None):
    """LayerNorm backward with the shortcut gradient folded in; d(gamma), d(beta) accumulated in place when the
    parameters are leaves (returns (dx, dgamma, dbeta) with None for gradients already accumulated).
    dres_windowed: dres lies in this call's window order; dx_window = shift: dx is WRITTEN in the window order of that shift."""
    dx = torch.empty_like(x)
    gg, gb = _grad_buf(gamma_p), _grad_buf(beta_p)
    lay = (1 if dres_windowed else 0, 0 if dx_window is None else 1, 0 if dx_window is None else int(dx_window))
    # dx2 = (shift, scale or None): a second copy of dx in the window order of that shift, times the per-image factor; returned as dx[1]
    d2 = None
    if dx2 is not None:
        d2 = torch.empty_like(x)
        lay = (lay[0], lay[1], int(dx2[0]))
        dx = (dx, d2)
    tail = (_p(d2), _p(dx2[1]) if dx2 is not None else None)
    dx0 = dx[0] if d2 is not None else dx
    if gg is not None and gb is not None:
        _lib.call("dhz_ln_partition_bwd_lay2", _p(dxn), _p(x), _p(gamma), _p(stats), _p(dres), _p(dx0), _p(gg), _p(gb),
                  B, Hres, Wres, C, shift, partition, *lay, *tail, ops._dt(x), _stream())
        _ready(gamma_p, beta_p)
        return dx, None, None
    dgb = torch.zeros((2, C), device=x.device, dtype=torch.float32)
    _lib.call("dhz_ln_partition_bwd_lay2", _p(dxn), _p(x), _p(gamma), _p(stats), _p(dres), _p(dx0), dgb[0].data_ptr(),
              dgb[1].data_ptr(), B, Hres, Wres, C, shift, partition, *lay, *tail, ops._dt(x), _stream())
    return dx, dgb[0], dgb[1]


class _Rec:
    """what one branch's forward leaves for its backward: tensors (through ctx.save_for_backward), parameters, geometry"""
    __slots__ = ("saved", "params", "geom", "kind")

    def __init__(self, kind=None, saved=(), params=None, geom=None):
        self.kind, self.saved, self.params, self.geom = kind, tuple(saved), params, geom


# Single-process runs: the 18 table gradients of a backward pass are reduced in ONE launch at its end (an autograd engine callback) instead
# of 18 launches of ~11 us.  With a gradient reducer attached (ops.GRAD_READY set: N > 1 ranks) every table is reduced where it is produced,
# so that its bucket's all-reduce is not held back to the end of the pass.
DEFER_TABLE_GRADS = True
_PENDING_TABLES = []


def _flush_table_grads():
    import ctypes
    pend = list(_PENDING_TABLES)
    _PENDING_TABLES.clear()
    arr = lambda ptrs: ctypes.cast((ctypes.c_void_p * len(ptrs))(*ptrs), ctypes.c_void_p)
    ints = lambda v: ctypes.cast((ctypes.c_int * len(v))(*v), ctypes.c_void_p)
    for i0 in range(0, len(pend), 32):
        part = pend[i0: i0 + 32]
        _lib.call("dhz_bias_table_grad_multi", arr([_p(d) for d, _, _, _ in part]), ints([n for _, n, _, _ in part]),
                  arr([_p(g) for _, _, g, _ in part]), ints([H for _, _, _, H in part]), len(part), _stream())


def _table_backward(dpart, parts, table_p, H, dev):
    gt = _grad_buf(table_p)
    if gt is not None and DEFER_TABLE_GRADS and ops.GRAD_READY is None:
        queued = bool(_PENDING_TABLES)
        if not queued:
            try:                                        # (only valid inside a running backward pass)
                torch.autograd.Variable._execution_engine.queue_callback(_flush_table_grads)
                queued = True
            except RuntimeError:
                queued = False
        if queued:
            _PENDING_TABLES.append((dpart, parts, gt, H))
            return None
    if gt is not None:
        _lib.call("dhz_bias_table_grad", _p(dpart), parts, _p(gt), H, 1, _stream())
        _ready(table_p)
        return None
    dtable = torch.empty((225, H), device=dev, dtype=torch.float32)
    _lib.call("dhz_bias_table_grad", _p(dpart), parts, _p(dtable), H, 0, _stream())
    return dtable


# ----------------------------------------------------------------------------- per-forward staging of parameter-derived operands
# Uformer.forward derives, in ONE launch each for the whole model, what every block would otherwise derive for itself in a launch
# of a few microseconds: the relative-position bias tiles (dhz_bias_gather_multi) and the fragment-ordered weight copies of the
# fused attention kernel (dhz_fused_attn_prepack_multi).  The branch forwards take a staged operand when there is one for their
# parameter (consumed on use: a second forward without a new staging derives its own again) and derive it themselves otherwise.
STAGED_BIAS = {}        # id(table parameter) -> [H, 64, 64]
STAGED_PREPACK = {}     # id(query weight)    -> (wqkv_p, wo_p)
STAGED_PACK6 = {}       # id(query weight)    -> six-term planes of Q / K / V / out-projection for the fused attention kernel
STAGED_LEFF6 = {}       # id(linear1 weight)  -> six-term planes of linear1 / linear2 for the fused LeFF kernel


def _n6(C):
    """bf16 elements of dhz_fused_attn_prepack6's planes: C = 32: 8 tiles x 3 pieces of 512; else per head 36 KiB-runs per 64 channels of Q / K / V
    and 3 per 16 output features of the out-projection"""
    return 8 * 3 * 512 if C == 32 else (C // 32) * ((C // 64) * 36 + (C // 16) * 3) * 512


def stage_block_operands(entries, device):
    """entries: [(table or None, H, (wq, wk, wv, wo) or None, C)] in execution order (Uformer.forward builds it)."""
    import ctypes
    STAGED_BIAS.clear()
    STAGED_PREPACK.clear()
    STAGED_PACK6.clear()
    STAGED_LEFF6.clear()
    _PENDING_TABLES.clear()          # (a backward pass that died before its end-of-pass callback)
    arr = lambda ptrs: ctypes.cast((ctypes.c_void_p * len(ptrs))(*ptrs), ctypes.c_void_p)
    tabs = [(t, H) for t, H, _, _ in entries if t is not None]
    if tabs:
        flat = torch.empty((sum(H for _, H in tabs) * NTOK * NTOK,), device=device, dtype=torch.float32)
        outs, off = [], 0
        for t, H in tabs:
            outs.append(flat[off: off + H * NTOK * NTOK].view(H, NTOK, NTOK))
            off += H * NTOK * NTOK
        for i0 in range(0, len(tabs), 32):
            part = list(zip(tabs, outs))[i0: i0 + 32]
            tcs = [t.contiguous() for (t, _), _ in part]
            _lib.call("dhz_bias_gather_multi", arr([_p(t) for t in tcs]), arr([_p(o) for _, o in part]),
                      ctypes.cast((ctypes.c_int * len(part))(*[H for (_, H), _ in part]), ctypes.c_void_p), len(part), _stream())
        for (t, _), o in zip(tabs, outs):
            STAGED_BIAS[id(t)] = (t, o)          # (the parameter itself rides along: its id cannot be reused while the entry lives)
    p6 = lambda C: ATTN_FUSED_P6 and C in ATTN_FUSED_P6_C
    packs6 = [(w, C) for _, _, w, C in entries if w is not None and p6(C)]
    if packs6:
        n6 = _n6
        flat = torch.empty((sum(n6(C) for _, C in packs6),), device=device, dtype=torch.bfloat16)
        outs, off = [], 0
        for _, C in packs6:
            outs.append(flat[off: off + n6(C)])
            off += n6(C)
        for i0 in range(0, len(packs6), 16):
            part = list(zip(packs6, outs))[i0: i0 + 16]
            ws = [[_p(w[k].contiguous()) for (w, _), _ in part] for k in range(4)]
            _lib.call("dhz_fused_attn_prepack6_multi", arr(ws[0]), arr(ws[1]), arr(ws[2]), arr(ws[3]), arr([_p(o) for _, o in part]),
                      ctypes.cast((ctypes.c_int * len(part))(*[C for (_, C), _ in part]), ctypes.c_void_p), len(part), _stream())
        for (w, _), o in zip(packs6, outs):
            STAGED_PACK6[id(w[0])] = (w[0], o)
    packs = [(w, C) for _, _, w, C in entries if w is not None and (not p6(C) or C in (32, 128))]     # (C = 128: the out-projection's fp32 pack too; C = 32: the fused backward's)
    if packs:
        flat = torch.empty((sum(4 * C * C for _, C in packs),), device=device, dtype=torch.float32)
        outs, off = [], 0
        for _, C in packs:
            outs.append((flat[off: off + 3 * C * C], flat[off + 3 * C * C: off + 4 * C * C]))
            off += 4 * C * C
        for i0 in range(0, len(packs), 16):
            part = list(zip(packs, outs))[i0: i0 + 16]
            ws = [[_p(w[k].contiguous()) for (w, _), _ in part] for k in range(4)]
            _lib.call("dhz_fused_attn_prepack_multi", arr(ws[0]), arr(ws[1]), arr(ws[2]), arr(ws[3]), arr([_p(o[0]) for _, o in part]),
                      arr([_p(o[1]) for _, o in part]), ctypes.cast((ctypes.c_int * len(part))(*[C for (_, C), _ in part]), ctypes.c_void_p),
                      len(part), _stream())
        for (w, _), o in zip(packs, outs):
            STAGED_PREPACK[id(w[0])] = (w[0], o)


def _bias_tile(table, H, dev):
    if table is None:
        return None
    hit = STAGED_BIAS.pop(id(table), None)
    if hit is not None and hit[0] is table and hit[1].shape[0] == H and hit[1].device == dev:
        return hit[1]
    bias = torch.empty((H, NTOK, NTOK), device=dev, dtype=torch.float32)
    _lib.call("dhz_bias_gather", _p(table.contiguous()), _p(bias), H, _stream())
    return bias


# ----------------------------------------------------------------------------- attention branch: forward / backward as functions
def _attn_fused_fwd(train, x, gamma, beta, wq, bq, wk, bk, wv, bv, wo, bo, table, idx, mask, dscale, Hres, Wres, shift, H):
    """one kernel: LN, roll, partition, QKV, ProbSparse core, out-projection, reverse, residual (csrc/fused_attn.hip)"""
    _require_gpu(x, gamma, beta, wq, wo, table, idx, mask, dscale)
    x = x.contiguous()
    B, L, C = x.shape
    assert L == Hres * Wres and C == 32 * H and C in SUPPORTED_C
    dev = x.device
    T = B * L
    f32 = dict(device=dev, dtype=torch.float32)
    use6 = ATTN_FUSED_P6 and C in ATTN_FUSED_P6_C
    n6 = _n6(C)
    wqkv_p = wo_p = None
    will_fuse_bwd = train and C in ATTN_FUSED_BWD_C and x.dtype == torch.float32      # (the fused backward reads the fp32 fragment pack)
    if not use6 or C == 128 or will_fuse_bwd:                  # fp32 fragment packs (C = 128 with six-term Q / K / V: the out-projection's only)
        hit = STAGED_PREPACK.pop(id(wq), None)
        if hit is not None and hit[0] is wq and hit[1][0].numel() == 3 * C * C and hit[1][0].device == dev:
            wqkv_p, wo_p = hit[1]
        else:
            wqkv_p = torch.empty(3 * C * C, **f32)
            wo_p = torch.empty(C * C, **f32)
            _lib.call("dhz_fused_attn_prepack", _p(wq), _p(wk), _p(wv), _p(wo), _p(wqkv_p), _p(wo_p), C, _stream())
    wqkv_f32 = wqkv_p
    if use6:
        # the weight products of every head on the bf16 matrix pipe (six-term, fp32-class): planes in the kernel's fragment order
        hit = STAGED_PACK6.pop(id(wq), None)
        if hit is not None and hit[0] is wq and hit[1].numel() == n6 and hit[1].device == dev:
            wqkv_p = hit[1]
        else:
            wqkv_p = torch.empty(n6, device=dev, dtype=torch.bfloat16)
            _lib.call("dhz_fused_attn_prepack6", _p(wq), _p(wk), _p(wv), _p(wo), _p(wqkv_p), C, _stream())
        if wo_p is None:
            wo_p = wqkv_p                                       # (C = 64: not read by the six-term kernel; a valid pointer for the argument check)
    bqkv = ops.cat_rows([bq.detach(), bk.detach(), bv.detach()])
    bias = _bias_tile(table, H, dev)
    out = torch.empty_like(x)
    fused_bwd = train and C in ATTN_FUSED_BWD_C and x.dtype == torch.float32
    xn = qkv = cx = stats = rank = None
    if train:
        if not fused_bwd:
            xn = torch.empty((T, C), **f32)
            qkv = torch.empty((T, 3 * C), **f32)
            cx = torch.empty((T, C), **f32)
            stats = torch.empty((T, 2), **f32)
        rank = torch.empty(((T // NTOK) * H * NTOK,), device=dev, dtype=torch.uint8)
    entry = "dhz_fused_window_attn_fwd6" if use6 else "dhz_fused_window_attn_fwd"
    timing = ops.KERNEL_TIMING.get("dhz_fused_window_attn_fwd") if ops.KERNEL_TIMING is not None else None
    if timing is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _lib.call(entry, _p(x), _p(gamma), _p(beta), _p(wqkv_p), _p(bqkv), _p(wo_p), _p(bo), _p(idx),
              _p(bias), _p(mask), _p(dscale), _p(out), _p(xn), _p(qkv), _p(cx), _p(stats), _p(rank), B, Hres, Wres, C,
              shift, _stream())
    if timing is not None:
        e1.record()
        timing.append((e0, e1, T // NTOK, C))
    rec = None
    params = (wq, bq, wk, bk, wv, bv, wo, bo, gamma, beta, table)
    geom = (B, Hres, Wres, C, shift, H)
    if fused_bwd:
        wt = torch.empty(4096, **f32)
        _lib.call("dhz_fused_attn_bwd_prepack", _p(wq), _p(wk), _p(wv), _p(wo), _p(wt), C, _stream())
        rec = _Rec("attn_fused_bwd", (x, gamma, beta, rank, bias, mask, dscale, wqkv_f32, bqkv, wt), params, geom)
    elif train:
        rec = _Rec("attn_chain_bwd", (x, gamma, stats, xn, qkv, cx, rank, bias, mask, dscale, wq, wk, wv, wo), params, geom)
    return out, rec


def _attn_chain_fwd(train, x, gamma, beta, wq, bq, wk, bk, wv, bv, wo, bo, table, idx, mask, dscale, Hres, Wres, shift, H):
    """dhz_ln_partition_fwd -> QKV GEMM -> dhz_ps_attn_fwd -> out-projection GEMM with the residual epilogue (K4 inside the GEMM)"""
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
    bias = _bias_tile(table, H, dev)
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
    out = ops.gemm_fwd_res(cx, wo, bo, x.view(T, C), dscale, B, Hres, Wres, shift, True).view(B, L, C)
    rec = None
    if train:
        rec = _Rec("attn_chain_bwd", (x, gamma, stats, xn, qkv, cx, rank, bias, mask, dscale, wq, wk, wv, wo),
                   (wq, bq, wk, bk, wv, bv, wo, bo, gamma, beta, table), (B, Hres, Wres, C, shift, H))
    return out, rec


def _attn_bwd_fused(rec, dout):
    """One kernel from d(out) to dx and every parameter gradient (csrc/fused_attn_bwd.hip)."""
    x, gamma, beta, rank, bias, mask, dscale, wqkv_p, bqkv, wt = rec.saved
    wq, bq, wk, bk, wv, bv, wo, bo, gamma_p, beta_p, table_p = rec.params
    B, Hres, Wres, C, shift, H = rec.geom
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
        return (dx, None, None, None, None, None, None, None, None, None, None, dtable)
    return (dx, gg, gb, gwq, gbq, gwk, gbk, gwv, gbv, gwo, gbo, dtable)


def _attn_bwd(rec, dout, windowed=False, daw_pre=None):
    """backward of the attention branch from d(out): (dx, dgamma, dbeta, dWq, dbq, dWk, dbk, dWv, dbv, dWo, dbo, dtable), None for
    gradients accumulated in place.  windowed: dout arrives in this branch's window order, WITHOUT the DropPath factor (written by
    the LeFF branch's LayerNorm backward, block()): no dhz_reverse_residual_bwd pass - the factor rides in the out-projection's
    backward-data epilogue and as the row factor of its weight gradient."""
    if rec.kind == "attn_fused_bwd":
        assert not windowed
        return _attn_bwd_fused(rec, dout)
    x, gamma, stats, xn, qkv, cx, rank, bias, mask, dscale, wq_, wk_, wv_, wo_ = rec.saved
    wq, bq, wk, bk, wv, bv, wo, bo, gamma_p, beta_p, table_p = rec.params
    B, Hres, Wres, C, shift, H = rec.geom
    d = C // H
    dout = dout.contiguous()
    dev = x.device
    T = B * Hres * Wres
    B_ = T // NTOK
    f32 = dict(device=dev, dtype=torch.float32)
    if daw_pre is not None:
        # (bf16 storage) the LeFF branch's LayerNorm backward wrote the scaled, window-ordered copy beside the token-order gradient
        daw = daw_pre.view(T, C)
        dctx = ops.gemm_dgrad(daw, wo_)
        g_wo, g_bo = _wgrad(daw, 0, cx, wo, bo)
    elif windowed:
        daw = dout.view(T, C)
        rs = (dscale, Hres * Wres) if dscale is not None else None
        dctx = ops.gemm_dgrad(daw, wo_, rs)
        g_wo, g_bo = _wgrad(daw, 0, cx, wo, bo, rs)
    else:
        # (1) gradient of the window-ordered out-projection output
        daw = torch.empty((T, C), device=dev, dtype=dout.dtype)
        _lib.call("dhz_reverse_residual_bwd_dt", _p(dout), _p(dscale), _p(daw), B, Hres, Wres, C, shift, 1, ops._dt(dout), _stream())
        # (2) out-projection
        dctx = ops.gemm_dgrad(daw, wo_)
        g_wo, g_bo = _wgrad(daw, 0, cx, wo, bo)
    # (3) attention core
    dqkv = torch.empty_like(qkv)
    parts = _lib.load().dhz_ps_attn_bwd_parts_d(B_, H, d)
    dpart = torch.empty((parts, NTOK, NTOK), **f32) if bias is not None else None
    nW = mask.shape[0] if mask is not None else 1
    base, gb = qkv.data_ptr(), dqkv.data_ptr()
    es = qkv.element_size()
    _lib.call("dhz_ps_attn_bwd_dt", base, base + es * C, base + 2 * es * C, 3 * C, _p(bias), _p(mask), _p(rank), _p(dctx), C,
              gb, gb + es * C, gb + 2 * es * C, 3 * C, _p(dpart), B_, H, nW, d, ops._dt(qkv), _stream())
    dtable = _table_backward(dpart, parts, table_p, H, dev) if bias is not None else None
    # (4) QKV projection
    dxn = ops.gemm_dgrad(dqkv, ops.cat_rows([wq_.detach(), wk_.detach(), wv_.detach()]))
    g_wq, g_bq, g_wk, g_bk, g_wv, g_bv = _wgrad_qkv(dqkv, xn, C, [(wq, bq), (wk, bk), (wv, bv)])
    # (5) LayerNorm backward + shortcut gradient in one pass
    dx, dgamma, dbeta = _ln_backward(dxn, x, gamma_p, beta_p, gamma, stats, dout, B, Hres, Wres, C, shift, 1, dres_windowed=windowed)
    return (dx, dgamma, dbeta, g_wq, g_bq, g_wk, g_bk, g_wv, g_bv, g_wo, g_bo, dtable)


def fused_c128_ok(HW):
    """C = 128: in a no-gradient forward the fused kernel writes no training saves and wins on every map size (whole-image eval, 416 x 416
    maps: 42 -> 39 ms per image); with the saves it ties with the kernel chain above 32 x 32 maps (31.25 / 31.26 ms per step) and stays below"""
    return HW <= ATTN_FUSED_C128_MAX_HW or not torch.is_grad_enabled()


def _use_fused_attn(x, heads, Hres, Wres):
    C = x.shape[-1]
    return ENABLED and x.dtype == torch.float32 and C == 32 * heads and (C in (32, 64) or (C == 128 and fused_c128_ok(Hres * Wres)))


class _AttnNode(Function):
    """the attention branch as one autograd node; FUSED selects the forward (the backward follows the record)"""

    @staticmethod
    def forward(ctx, fused, x, gamma, beta, wq, bq, wk, bk, wv, bv, wo, bo, table, idx, mask, dscale, Hres, Wres, shift, H, grad_mode):
        train = grad_mode and any(ctx.needs_input_grad)      # grad mode reads False inside Function.forward: passed in
        out, rec = (_attn_fused_fwd if fused else _attn_chain_fwd)(train, x, gamma, beta, wq, bq, wk, bk, wv, bv, wo, bo, table, idx, mask,
                                                                   dscale, Hres, Wres, shift, H)
        if rec is not None:
            ctx.save_for_backward(*rec.saved)
            rec.saved = ()
        ctx.rec = rec
        return out

    @staticmethod
    def backward(ctx, dout):
        rec = ctx.rec
        rec.saved = ctx.saved_tensors
        g = _attn_bwd(rec, dout)
        rec.saved = ()
        return (None,) + g + (None,) * 8


def fused_attn_branch(x, norm, layer, table, idx, mask, dscale, Hres, Wres, shift, heads):
    """x: [B,L,C]; norm: nn.LayerNorm; layer: AttentionLayer (query/key/value/out projections)."""
    q, k, v, o = layer.query_projection, layer.key_projection, layer.value_projection, layer.out_projection
    return _AttnNode.apply(True, x, norm.weight, norm.bias, q.weight, q.bias, k.weight, k.bias, v.weight, v.bias,
                           o.weight, o.bias, table, idx, mask, dscale, Hres, Wres, shift, heads, torch.is_grad_enabled())


def attn_branch(x, norm, layer, table, idx, mask, dscale, Hres, Wres, shift, heads):
    """Dispatch: fused kernel where it wins (measured, tools/bench_fused.py), kernel chain elsewhere."""
    q, k, v, o = layer.query_projection, layer.key_projection, layer.value_projection, layer.out_projection
    return _AttnNode.apply(_use_fused_attn(x, heads, Hres, Wres), x, norm.weight, norm.bias, q.weight, q.bias, k.weight, k.bias, v.weight,
                           v.bias, o.weight, o.bias, table, idx, mask, dscale, Hres, Wres, shift, heads, torch.is_grad_enabled())


# ----------------------------------------------------------------------------- LeFF branch
def _leff_fwd(train, x, gamma, beta, w1, b1, wd, bd, w2, b2, dscale, Hres, Wres):
    """out = x + drop_scale * linear2(gelu(dwconv3x3(gelu(linear1(LayerNorm(x))))))       (M1:873 + M1:496-534)
    one kernel (csrc/leff_fused.hip) at C = 32 / 64, else dhz_ln_partition_fwd (plain LN) -> GEMM -> dhz_leff_dwconv_fwd -> GEMM with
    the residual epilogue (token order)"""
    _require_gpu(x, gamma, beta, w1, wd, w2, dscale)
    x = x.contiguous()
    B, L, C = x.shape
    Ch = w1.shape[0]
    T = B * L
    dev = x.device
    f32 = dict(device=dev, dtype=torch.float32)
    wdc = wd.contiguous()
    tiled = Hres % 8 == 0 and Wres % 16 == 0
    fp32 = x.dtype == torch.float32                      # the fused LeFF kernels are fp32-only
    if LEFF_FUSED and C in LEFF_FUSED_C and tiled and fp32 and not (C == 64 and T > LEFF_FUSED_C64_MAX_T and ops.SPLIT_BF16 == 6):
        # one kernel: norm2, linear1, GELU, depthwise 3x3, GELU, linear2, DropPath scale, residual (csrc/leff_fused.hip)
        out = torch.empty_like(x)
        xn = stats = u = tg = z = None
        if train:
            xn = torch.empty((T, C), **f32)
            stats = torch.empty((T, 2), **f32)
            u = torch.empty((T, Ch), **f32)
            tg = torch.empty((T, Ch), **f32)
            z = torch.empty((T, Ch), **f32)
        if LEFF_FUSED_P6 and C in LEFF_FUSED_P6_C:
            # both weight products six-term on the bf16 matrix pipe: planes in the kernel's fragment order (staged per forward, or packed here)
            hit = STAGED_LEFF6.pop(id(w1), None)
            if hit is not None and hit[0] is w1 and hit[1].numel() == 24 * C * C and hit[1].device == dev:
                w6 = hit[1]
            else:
                w6 = torch.empty(24 * C * C, device=dev, dtype=torch.bfloat16)
                _lib.call("dhz_leff_prepack6", _p(w1), _p(w2), _p(w6), C, _stream())
            _lib.call("dhz_leff_fused_fwd6", _p(x), _p(gamma), _p(beta), _p(w6), _p(b1), _p(wdc), _p(bd), _p(b2),
                      _p(dscale), _p(out), _p(xn), _p(stats), _p(u), _p(tg), _p(z), B, Hres, Wres, C, _stream())
        else:
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
        out = ops.gemm_fwd_res(z, w2, b2, x.view(T, C), dscale, B, L, 1, 0, False).view(B, L, C)
    rec = None
    if train:
        rec = _Rec("leff", (x, gamma, stats, xn, u, tg, z, dscale, w1, wdc, w2), (w1, b1, wd, bd, w2, b2, gamma, beta), (B, L, C, Ch, Hres, Wres))
    return out, rec


def _leff_bwd(rec, dout, dx_window=None, dx2=None):
    """the mirror sequence with in-place weight gradients and the shortcut gradient folded into the LayerNorm backward (no autograd
    accumulation kernels): (dx, dgamma, dbeta, dW1, db1, dWd, dbd, dW2, db2).  dx_window = shift: dx is written in the window order
    of that shift on the Hres x Wres map (block(): what the attention branch's backward reads)."""
    x, gamma, stats, xn, u, tg, z, dscale, w1_, wdc, w2_ = rec.saved
    w1, b1, wd, bd, w2, b2, gamma_p, beta_p = rec.params
    B, L, C, Ch, Hres, Wres = rec.geom
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
    if dx2 is not None:       # dx in token order AND a scaled window-ordered copy (block(), bf16 storage): dx = (dx, copy)
        dx, dgamma, dbeta = _ln_backward(dxn, x, gamma_p, beta_p, gamma, stats, dout, B, Hres, Wres, C, 0, 0, dx2=dx2)
    elif dx_window is None:
        dx, dgamma, dbeta = _ln_backward(dxn, x, gamma_p, beta_p, gamma, stats, dout, B, L, 1, C, 0, 0)
    else:
        dx, dgamma, dbeta = _ln_backward(dxn, x, gamma_p, beta_p, gamma, stats, dout, B, Hres, Wres, C, 0, 0, dx_window=dx_window)
    return (dx, dgamma, dbeta, g_w1, g_b1, g_wd, g_bd, g_w2, g_b2)


class _LeffBranch(Function):
    """the LeFF branch as one autograd node"""

    @staticmethod
    def forward(ctx, x, gamma, beta, w1, b1, wd, bd, w2, b2, dscale, Hres, Wres, grad_mode):
        train = grad_mode and any(ctx.needs_input_grad)      # grad mode reads False inside Function.forward: passed in
        out, rec = _leff_fwd(train, x, gamma, beta, w1, b1, wd, bd, w2, b2, dscale, Hres, Wres)
        if rec is not None:
            ctx.save_for_backward(*rec.saved)
            rec.saved = ()
        ctx.rec = rec
        return out

    @staticmethod
    def backward(ctx, dout):
        rec = ctx.rec
        rec.saved = ctx.saved_tensors
        g = _leff_bwd(rec, dout)
        rec.saved = ()
        return g + (None, None, None, None)


class _FfnBranch(Function):
    """out = x + drop_scale * fc2(gelu(fc1(LayerNorm(x))))        (M1:873 with token_mlp = 'ffn': M1:442-468)
    forward : dhz_ln_partition_fwd (plain LN) -> GEMM -> dhz_gelu_fwd -> GEMM with the residual epilogue (token order)
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
        out = ops.gemm_fwd_res(z, w2, b2, x.view(T, C), dscale, B, L, 1, 0, False).view(B, L, C)
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


# ----------------------------------------------------------------------------- the whole block as one node
DUAL_BF16 = True        # bf16 storage: window-ordered scaled gradient copy from the LeFF LayerNorm backward (False: dhz_reverse_residual_bwd)
BLOCK_NODE = True       # False: two nodes per block (attn_branch, leff_branch) with a token-order gradient between them


class _BlockNode(Function):
    """attention branch + LeFF branch of one LeWin block (M1:839-873).  Forward = the two branch forwards; backward = the two branch
    backwards with the gradient between them in the attention branch's window order whenever that branch runs the kernel chain
    backward (module docstring).  Argument layout: (fused, x, 18 attention arguments, 11 LeFF arguments, grad_mode)."""
    NA, NL = 18, 11

    @staticmethod
    def forward(ctx, fused, x, *args):
        a, l, grad_mode = args[:_BlockNode.NA], args[_BlockNode.NA:_BlockNode.NA + _BlockNode.NL], args[-1]
        train = grad_mode and any(ctx.needs_input_grad)
        x1, ra = (_attn_fused_fwd if fused else _attn_chain_fwd)(train, x, *a)
        out, rl = _leff_fwd(train, x1, *l)
        if train:
            ctx.na = len(ra.saved)
            ctx.save_for_backward(*ra.saved, *rl.saved)
            ra.saved = rl.saved = ()
        ctx.recs = (ra, rl)
        return out

    @staticmethod
    def backward(ctx, dout):
        ra, rl = ctx.recs
        sv = ctx.saved_tensors
        ra.saved, rl.saved = sv[:ctx.na], sv[ctx.na:]
        B, Hres, Wres, C, shift, H = ra.geom
        # window-order hand-over: chain backward, fp32 storage (the row-factor forms of the products), whole 64-token images
        windowed = ra.kind == "attn_chain_bwd" and dout.dtype == torch.float32 and rl.geom[4:] == (Hres, Wres) and (Hres * Wres) % 64 == 0
        # bf16 storage (the row-factor forms do not exist there): the same LayerNorm backward writes the scaled window-ordered copy
        # beside the token-order gradient - one more store instead of the dhz_reverse_residual_bwd pass
        dual = (not windowed) and ra.kind == "attn_chain_bwd" and dout.dtype == ops.BF16 and rl.geom[4:] == (Hres, Wres) \
            and Hres % 8 == 0 and Wres % 8 == 0 and DUAL_BF16
        if dual:
            gl = _leff_bwd(rl, dout, dx2=(shift, ra.saved[9]))          # ra.saved[9]: the attention branch's DropPath factor (or None)
            ga = _attn_bwd(ra, gl[0][0], daw_pre=gl[0][1])
        else:
            gl = _leff_bwd(rl, dout, dx_window=shift if windowed else None)
            ga = _attn_bwd(ra, gl[0], windowed=windowed)
        ra.saved = rl.saved = ()
        #      fused   x       attention parameters (11)   idx, mask, dscale, Hres, Wres, shift, H      LeFF parameters (8)   dscale, Hres, Wres, grad_mode
        return (None, ga[0]) + ga[1:] + (None,) * 7 + gl[1:] + (None,) * 4


def block(x, norm1, layer, table, idx, mask, dscale_attn, Hres, Wres, shift, heads, norm2, mlp, dscale_mlp):
    """both branches of a LeWin block with a LeFF token mixer as one autograd node"""
    q, k, v, o = layer.query_projection, layer.key_projection, layer.value_projection, layer.out_projection
    return _BlockNode.apply(_use_fused_attn(x, heads, Hres, Wres), x,
                            norm1.weight, norm1.bias, q.weight, q.bias, k.weight, k.bias, v.weight, v.bias, o.weight, o.bias, table, idx,
                            mask, dscale_attn, Hres, Wres, shift, heads,
                            norm2.weight, norm2.bias, mlp.linear1[0].weight, mlp.linear1[0].bias, mlp.dwconv[0].weight, mlp.dwconv[0].bias,
                            mlp.linear2[0].weight, mlp.linear2[0].bias, dscale_mlp, Hres, Wres,
                            torch.is_grad_enabled())
