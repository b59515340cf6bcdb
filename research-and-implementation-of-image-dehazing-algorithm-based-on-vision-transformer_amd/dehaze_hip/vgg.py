"""VGG19[:30] feature stack of the contrastive loss (My_CR.py:56-86) on the Winograd-MFMA convolution kernel.

conv 0 (3 -> 64) runs thin-input kernels (dhz_conv3x3_in3_blocked forward with bias + ReLU straight into the blocked layout,
dhz_thin_conv3x3_dgrad_blocked backward-data straight from it); convs 1..12 run dhz_winograd_conv3x3 with bias + ReLU fused,
in the channel-blocked NCHW8c layout end to end (conv 12 works on 8x8 maps for 128x128 patches: the kernel then packs four
images into one of its 16x16 blocks; for patch sizes whose conv-12 maps it does not tile, e.g. 24x24 at 384x384, that one layer
falls back to the library).  The filters are frozen
(My_CR.py:75-77), so their transform-domain forms (forward and backward-data) are prepacked once per device.

Forward-only passes (target / hazy input, My_CR.py:102) save nothing; the pass on the restored image is one autograd
node whose backward walks the stack with the same kernel (rotated/transposed filters, ReLU mask fused into the patch
load) - weight gradients are never formed (the reference freezes the VGG, My_CR.py:75-77).
"""
import os

import torch
import torch.nn.functional as F
from torch.autograd import Function

from . import _lib
from . import ops
from .ops import _p, _stream

CONVS = ((3, 64), (64, 64), (64, 128), (128, 128), (128, 256), (256, 256), (256, 256), (256, 256), (256, 512),
         (512, 512), (512, 512), (512, 512), (512, 512))
POOL_AFTER = (1, 3, 7, 11)
TAPS = (0, 2, 4, 8, 12)          # relu1_1, relu2_1, relu3_1, relu4_1, relu5_1


def wino_supported(H, W):
    """map sizes dhz_winograd_conv3x3 tiles: multiples of 16, or 8x8 (four images per block)"""
    return (H % 16 == 0 and W % 16 == 0) or (H == 8 and W == 8)


def to_blocked(x, bias=None, relu=False):
    """NCHW -> NCHW8c; optionally max(x + bias[c], 0) on the way (the bias + ReLU behind a library convolution)."""
    B, C, H, W = x.shape
    out = torch.empty((B, C // 8, H, W, 8), device=x.device, dtype=torch.float32)
    _lib.call("dhz_layout_blocked8", _p(x.contiguous()), _p(out), B, C, H * W, 1, _p(bias), int(relu), _stream())
    return out


def to_plain(xb):
    B, CG, H, W, _ = xb.shape
    out = torch.empty((B, CG * 8, H, W), device=xb.device, dtype=torch.float32)
    _lib.call("dhz_layout_blocked8", _p(xb), _p(out), B, CG * 8, H * W, 0, None, 0, _stream())
    return out


def pool_fwd(xb):
    B, CG, H, W, _ = xb.shape
    yb = torch.empty((B, CG, H // 2, W // 2, 8), device=xb.device, dtype=torch.float32)
    _lib.call("dhz_maxpool2x2_blocked_fwd", _p(xb), _p(yb), B * CG, H, W, _stream())
    return yb


def pool_bwd_relu(gb, actb):
    """gradient w.r.t. the pre-activation below the post-ReLU map `actb` whose 2x2 max pooling received gradient gb."""
    B, CG, H, W, _ = actb.shape
    gx = torch.empty_like(actb)
    _lib.call("dhz_maxpool2x2_blocked_bwd", _p(gb.contiguous()), _p(actb), _p(gx), B * CG, H, W, _stream())
    return gx


def _timing_begin():
    """bench.py's per-kernel HIP events (ops.KERNEL_TIMING), on the launch stream, bracketing exactly this kernel."""
    timing = ops.KERNEL_TIMING.get("dhz_winograd_conv3x3") if ops.KERNEL_TIMING is not None else None
    if timing is None:
        return None
    e0 = torch.cuda.Event(enable_timing=True)
    e0.record()
    return timing, e0


def _timing_end(ev, B, H, W, Cin, Kout, f43=False):
    if ev is not None:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        direct = 2.0 * 9 * B * H * W * Cin * Kout                             # direct-convolution FLOPs of this launch
        # ISSUED matrix FLOPs: F(2x2,3x3) multiplies 16 / 4 per output and tap-sum instead of 9 (direct / 2.25), F(4x4,3x3) 36 / 16 (/ 4)
        ev[0].append((ev[1], e1, direct, direct / (4.0 if f43 else 2.25)))


# Which Winograd form a layer takes: F(4x4,3x3) (csrc/winograd43_conv.hip, 1.78 x fewer matrix products) or F(2x2,3x3)
# (csrc/winograd_conv.hip).  Measured per layer (tools/bench_wino.py): F(4x4) is 1.14 - 1.22 x faster whenever its grid - one workgroup
# per four 16 x 16 blocks and 32 output channels - fills the CUs in whole rounds: always on the 128 / 64 / 32-pixel maps (thousands of
# workgroups), on the 16-pixel maps only for some batches (64 images x 512 channels = 256 workgroups = one round: faster; 96 images = 1.5
# rounds: 0.91 x; 32 images = half a round: slower still) - hence the occupancy rule below.  DHZ_WINO_F43=0 disables the F(4x4) form.
F43_ON = os.environ.get("DHZ_WINO_F43", "1") != "0"


# F(4x4) also in the DIFFERENTIATED forward pass (the pass whose roundings decide the ReLU masks of the backward pass).  Rounds 4 - 5 kept that
# pass on F(2x2) by an argument (three times the near-zero units flip); round 6 measured it (tools/wino_f43_diff.py, profiles/
# r06_wino_f43_diff.txt): over 50 training steps the loss differs from the all-F(2x2)-forward run by <= 1.6e-4 - inside the run-to-run spread of
# the F(2x2) step itself (atomics) -, d(loss)/d(restored) at equal weights by 9e-4 of its mean, every kernel- and model-level tolerance of
# tests/ holds unchanged with it, and the step gains 0.13 - 0.28 ms.  DHZ_WINO_F43_DIFF=0 restores the F(2x2) forward.
F43_DIFF = os.environ.get("DHZ_WINO_F43_DIFF", "1") != "0"
POOL_FUSED = os.environ.get("DHZ_WINO_POOL", "1") != "0"    # A/B switch: the pooling of the no-gradient pass inside the F(4x4) launch


def use_f43(B, H, W, Cin, Kout):
    if not F43_ON or H % 16 or W % 16 or H < 16 or W < 16 or Cin % 16 or Kout % 32:
        return False
    wgs = (B * (H // 16) * (W // 16) + 3) // 4 * (Kout // 32)
    cus = _lib.load().dhz_grid_cus() + _lib.load().dhz_get_reserved_cus()      # PHYSICAL CUs: this grid is not persistent, a reservation does not shrink it
    rounds = (wgs + cus - 1) // cus
    return wgs >= 0.92 * rounds * cus          # the last round of workgroups nearly full


class VggEngine:
    def __init__(self, convs):
        """convs: the 13 nn.Conv2d modules of vgg19.features[0:30]."""
        self.convs = convs
        self._packed = {}
        self._packed43 = {}

    def packed(self, i, device):
        key = (i, str(device), self.convs[i].weight.data_ptr(), self.convs[i].weight._version)
        hit = self._packed.get(i)
        if hit is None or hit[0] != key:
            w = self.convs[i].weight.detach().contiguous()
            K, C = w.shape[0], w.shape[1]
            uf = torch.empty(16 * K * C, device=device, dtype=torch.float32)
            ub = torch.empty(16 * K * C, device=device, dtype=torch.float32)
            _lib.call("dhz_winograd_prepack", _p(w), _p(uf), K, C, 0, _stream())
            _lib.call("dhz_winograd_prepack", _p(w), _p(ub), C, K, 1, _stream())
            hit = (key, uf, ub)
            self._packed[i] = hit
        return hit[1], hit[2]

    def packed43(self, i, device):
        """F(4x4,3x3) filters of layer i (forward and backward-data forms, 36 K C floats each), built at first use"""
        key = (i, str(device), self.convs[i].weight.data_ptr(), self.convs[i].weight._version)
        hit = self._packed43.get(i)
        if hit is None or hit[0] != key:
            w = self.convs[i].weight.detach().contiguous()
            K, C = w.shape[0], w.shape[1]
            uf = torch.empty(36 * K * C, device=device, dtype=torch.float32)
            ub = torch.empty(36 * K * C, device=device, dtype=torch.float32)
            _lib.call("dhz_winograd43_prepack", _p(w), _p(uf), K, C, 0, _stream())
            _lib.call("dhz_winograd43_prepack", _p(w), _p(ub), C, K, 1, _stream())
            hit = (key, uf, ub)
            self._packed43[i] = hit
        return hit[1], hit[2]

    # ---- one Winograd layer
    def conv(self, i, xb, allow43=True, pool=False):
        """conv i + bias + ReLU; pool=True: + the 2 x 2 max pooling behind it, in the same launch when the layer runs the F(4x4) kernel (the
        un-pooled map is then never written: only for layers whose output is no tap and is not saved)."""
        B, CG, H, W, _ = xb.shape
        C, K = CONVS[i]
        f43 = allow43 and use_f43(B, H, W, C, K)
        uf, _ = self.packed43(i, xb.device) if f43 else self.packed(i, xb.device)
        if pool and f43:
            yp = torch.empty((B, K // 8, H // 2, W // 2, 8), device=xb.device, dtype=torch.float32)
            scratch = torch.empty((B, K // 8, H, W, 8), device=xb.device, dtype=torch.float32) if C > 256 else None
            ev = _timing_begin()
            _lib.call("dhz_winograd43_conv3x3_pool", _p(xb), _p(uf), _p(self.convs[i].bias), _p(yp), _p(scratch) if scratch is not None else None,
                      B, H, W, C, K, _stream())
            _timing_end(ev, B, H, W, C, K, f43)
            return yp
        yb = torch.empty((B, K // 8, H, W, 8), device=xb.device, dtype=torch.float32)
        ev = _timing_begin()
        _lib.call("dhz_winograd43_conv3x3" if f43 else "dhz_winograd_conv3x3", _p(xb), _p(uf), _p(self.convs[i].bias), 1, None, None,
                  _p(yb), B, H, W, C, K, _stream())
        _timing_end(ev, B, H, W, C, K, f43)
        return pool_fwd(yb) if pool else yb

    def conv_dgrad(self, i, gb, below_act=None, addend=None):
        """gb: gradient w.r.t. the pre-activation of conv i.  Returns the gradient w.r.t. conv i's input; with
        `below_act` (the saved post-ReLU map that IS that input) the tap gradient `addend` is added and the ReLU below
        applied in the kernel's store, i.e. the result is the gradient w.r.t. the pre-activation of conv i-1."""
        B, KG, H, W, _ = gb.shape
        C, K = CONVS[i]
        f43 = use_f43(B, H, W, K, C)                       # the backward-data form: K input channels, C outputs
        _, ub = self.packed43(i, gb.device) if f43 else self.packed(i, gb.device)
        dxb = torch.empty((B, C // 8, H, W, 8), device=gb.device, dtype=torch.float32)
        ev = _timing_begin()
        _lib.call("dhz_winograd43_conv3x3" if f43 else "dhz_winograd_conv3x3", _p(gb), _p(ub), None, 0,
                  _p(below_act) if below_act is not None else None,
                  _p(addend.contiguous()) if addend is not None else None, _p(dxb), B, H, W, K, C, _stream())
        _timing_end(ev, B, H, W, K, C, f43)
        return dxb

    # ---- full stack, forward only
    def forward_taps(self, x, save=None):
        """x: [B,3,H,W] NCHW.  Returns the 5 tap features, blocked [B,C/8,H,W,8] (tap 5 NCHW when conv 12 ran on the library).
        `save` (dict) receives what the backward needs."""
        c0 = self.convs[0]
        # first layer (3 -> 64): thin on the input side - convolution + bias + ReLU straight into the blocked layout
        xc = x.contiguous()
        cur = torch.empty((xc.shape[0], 8, xc.shape[2], xc.shape[3], 8), device=x.device, dtype=torch.float32)
        _lib.call("dhz_conv3x3_in3_blocked", _p(xc), _p(c0.weight.contiguous()), _p(c0.bias), _p(cur), xc.shape[0], xc.shape[2],
                  xc.shape[3], 64, 1, _stream())
        acts = {0: cur}
        taps = [cur]
        x12 = None
        for i in range(1, 13):
            if i == 12 and not wino_supported(cur.shape[2], cur.shape[3]):
                # conv 12 on maps the kernel does not tile (e.g. 24x24 for 384x384 patches): library convolution, NCHW tap
                c12 = self.convs[12]
                ops.warn_library_fallback("VGG19 conv5_1", tuple(cur.shape))
                x12 = to_plain(cur)
                cur = F.relu(F.conv2d(x12, c12.weight, c12.bias, padding=1))
            else:
                # F(4x4,3x3) carries ~3 x the rounding error of F(2x2,3x3) (1.1 - 2.1e-6 rms on O(1) features against 3.7 - 6.7e-7): harmless for
                # feature VALUES and backward-data products; for the differentiated pass see F43_DIFF above (measured, on by default)
                if POOL_FUSED and save is None and i in POOL_AFTER and i not in TAPS:
                    cur = self.conv(i, cur, pool=True)          # no-gradient pass, the un-pooled map has no other reader
                    continue
                cur = self.conv(i, cur, allow43=(save is None or F43_DIFF))
            acts[i] = cur
            if i in TAPS:
                taps.append(cur)
            if i in POOL_AFTER:
                cur = pool_fwd(cur)
        if save is not None:
            save.update(x=x, acts=acts, x12=x12)
        return taps


class _VggTaps(Function):
    """taps(x) with gradient w.r.t. x only (frozen weights)."""

    @staticmethod
    def forward(ctx, engine, x):
        saved = {}
        with torch.no_grad():
            taps = engine.forward_taps(x, save=saved)
        ctx.engine, ctx.saved = engine, saved
        return tuple(taps)

    @staticmethod
    def backward(ctx, g1, g2, g3, g4, g5):
        eng, sv = ctx.engine, ctx.saved
        acts = sv["acts"]
        tap_grad = {0: g1, 2: g2, 4: g3, 8: g4}
        c0 = eng.convs[0]
        with torch.no_grad():
            # G = gradient w.r.t. the PRE-activation of conv i (None while nothing has arrived from above)
            G = g5 * (acts[12] > 0) if g5 is not None else None
            top = 12
            if sv["x12"] is not None:                                        # conv 12 ran on the library (see forward_taps)
                top = 11
                if G is not None:
                    gx12 = torch.ops.aten.convolution_backward(G, sv["x12"], eng.convs[12].weight, None, [1, 1], [1, 1], [1, 1],
                                                               False, [0, 0], 1, [True, False, False])[0]
                    G = pool_bwd_relu(to_blocked(gx12), acts[11])
            for i in range(top, 0, -1):
                below = i - 1                                                # conv i's input is a_{i-1} (pooled if i-1 in POOL_AFTER)
                tg = tap_grad.get(below)
                if G is None:
                    if tg is not None:                                       # the first gradient enters at this tap
                        G = tg * (acts[below] > 0)
                    continue
                if below in POOL_AFTER:
                    G = pool_bwd_relu(eng.conv_dgrad(i, G), acts[below])
                else:
                    G = eng.conv_dgrad(i, G, below_act=acts[below], addend=tg)
            if G is None:
                return None, None
            # first layer (3 -> 64): its backward-data is a thin 64 -> 3 convolution - straight from the blocked gradient
            x0 = sv["x"]
            gx = torch.empty_like(x0, memory_format=torch.contiguous_format)
            _lib.call("dhz_thin_conv3x3_dgrad_blocked", _p(G.contiguous()), _p(c0.weight.contiguous()), _p(gx), x0.shape[0],
                      x0.shape[2], x0.shape[3], 64, _stream())
        ctx.saved = None
        return None, gx


# ------------------------------------------------------------------------------------------------ bf16 feature maps
class VggEngineBF16:
    """The same stack for BASELINE config 4: feature maps in bf16, NHWC ([N, H, W, C]), every 64+-channel convolution an implicit
    GEMM on the bf16 matrix pipe (csrc/vgg_bf16.hip), conv 0 on the input-projection kernel (3 -> 64, bias + ReLU, bf16 tokens out)
    and its backward-data on the output-projection kernel (64 -> 3 from bf16 tokens) with the rotated filter.  The tap features
    leave in NHWC - the L1 means do not care about element order as long as a / p / n agree."""

    dtype = torch.bfloat16

    def __init__(self, convs):
        self.convs = convs
        self._packed = {}

    def packed(self, i, device):
        wt = self.convs[i].weight
        key = (i, str(device), wt.data_ptr(), wt._version)
        hit = self._packed.get(i)
        if hit is None or hit[0] != key:
            w = wt.detach().float().contiguous()
            K, C = w.shape[0], w.shape[1]
            if i == 0:
                # backward-data of conv 0 as a thin 64 -> 3 convolution: filter [3][64][3][3], taps rotated
                hit = (key, w, w.flip(2, 3).transpose(0, 1).contiguous())
            else:
                wf = torch.empty(K * 9 * C, device=device, dtype=torch.bfloat16)
                wb = torch.empty(K * 9 * C, device=device, dtype=torch.bfloat16)
                _lib.call("dhz_vgg_prepack_bf16", _p(w), _p(wf), K, C, 0, _stream())
                _lib.call("dhz_vgg_prepack_bf16", _p(w), _p(wb), K, C, 1, _stream())
                hit = (key, wf, wb)
            self._packed[i] = hit
        return hit[1], hit[2]

    @staticmethod
    def _timing_begin():
        timing = ops.KERNEL_TIMING.get("dhz_vgg_conv3x3_bf16") if ops.KERNEL_TIMING is not None else None
        if timing is None:
            return None
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
        return timing, e0

    def conv(self, i, x):
        N, H, W, C = x.shape
        K = CONVS[i][1]
        wf, _ = self.packed(i, x.device)
        y = torch.empty((N, H, W, K), device=x.device, dtype=torch.bfloat16)
        ev = self._timing_begin()
        _lib.call("dhz_vgg_conv3x3_bf16", _p(x), _p(wf), _p(self.convs[i].bias), 1, None, None, _p(y), N, H, W, C, K, _stream())
        _timing_end(ev, N, H, W, C, K)
        return y

    def conv_dgrad(self, i, g, below_act=None, addend=None):
        """as VggEngine.conv_dgrad, on NHWC bf16 maps"""
        g = g.contiguous()                     # (a tap gradient may arrive as a permuted view of an NCHW tensor)
        N, H, W, K = g.shape
        C = CONVS[i][0]
        _, wb = self.packed(i, g.device)
        dx = torch.empty((N, H, W, C), device=g.device, dtype=torch.bfloat16)
        ev = self._timing_begin()
        _lib.call("dhz_vgg_conv3x3_bf16", _p(g), _p(wb), None, 0, _p(below_act) if below_act is not None else None,
                  _p(addend.contiguous()) if addend is not None else None, _p(dx), N, H, W, K, C, _stream())
        _timing_end(ev, N, H, W, K, C)
        return dx

    @staticmethod
    def pool_fwd(x):
        N, H, W, C = x.shape
        y = torch.empty((N, H // 2, W // 2, C), device=x.device, dtype=torch.bfloat16)
        _lib.call("dhz_maxpool2x2_nhwc_bf16_fwd", _p(x), _p(y), N, H, W, C, _stream())
        return y

    @staticmethod
    def pool_bwd_relu(g, act):
        N, H, W, C = act.shape
        gx = torch.empty_like(act)
        _lib.call("dhz_maxpool2x2_nhwc_bf16_bwd", _p(g.contiguous()), _p(act), _p(gx), N, H, W, C, _stream())
        return gx

    def forward_taps(self, x, save=None):
        """x: [N,3,H,W] NCHW fp32.  Returns the 5 tap features [N,H,W,C] bf16."""
        c0 = self.convs[0]
        xc = x.contiguous().float()
        N, _, H, W = xc.shape
        w0, _ = self.packed(0, x.device)
        cur = torch.empty((N, H, W, 64), device=x.device, dtype=torch.bfloat16)
        # LeakyReLU with slope 0 = the ReLU behind conv 0
        _lib.call("dhz_input_proj_fwd_dt", _p(xc), _p(w0), _p(c0.bias), _p(cur), N, H, W, 64, 0.0, 1, _stream())
        acts = {0: cur}
        taps = [cur]
        for i in range(1, 13):
            cur = self.conv(i, cur)
            acts[i] = cur
            if i in TAPS:
                taps.append(cur)
            if i in POOL_AFTER:
                cur = self.pool_fwd(cur)
        if save is not None:
            save.update(x=xc, acts=acts)
        return taps

    def backward_taps(self, saved, grads):
        """gradient w.r.t. the image from the gradients of the five taps (None where a tap received none)"""
        acts = saved["acts"]
        g1, g2, g3, g4, g5 = grads
        tap_grad = {0: g1, 2: g2, 4: g3, 8: g4}
        G = g5 * (acts[12] > 0) if g5 is not None else None
        for i in range(12, 0, -1):
            below = i - 1
            tg = tap_grad.get(below)
            if G is None:
                if tg is not None:
                    G = tg * (acts[below] > 0)
                continue
            if below in POOL_AFTER:
                G = self.pool_bwd_relu(self.conv_dgrad(i, G), acts[below])
            else:
                G = self.conv_dgrad(i, G, below_act=acts[below], addend=tg)
        if G is None:
            return None
        x0 = saved["x"]
        N, _, H, W = x0.shape
        _, w0r = self.packed(0, x0.device)
        gx = torch.empty_like(x0)
        _lib.call("dhz_thin_conv3x3_fwd_dt", _p(G.contiguous()), _p(w0r), None, _p(gx), N, H, W, 64, 1, _stream())
        return gx


class _VggTapsBF16(Function):
    @staticmethod
    def forward(ctx, engine, x):
        saved = {}
        with torch.no_grad():
            taps = engine.forward_taps(x, save=saved)
        ctx.engine, ctx.saved, ctx.xdtype = engine, saved, x.dtype
        return tuple(taps)

    @staticmethod
    def backward(ctx, *grads):
        with torch.no_grad():
            gx = ctx.engine.backward_taps(ctx.saved, grads)
        ctx.saved = None
        return None, (gx.to(ctx.xdtype) if gx is not None else None)


def vgg_taps(engine, x):
    if torch.is_grad_enabled() and x.requires_grad:
        return list((_VggTapsBF16 if isinstance(engine, VggEngineBF16) else _VggTaps).apply(engine, x))
    with torch.no_grad():
        return engine.forward_taps(x)


class _ToPlain(Function):
    @staticmethod
    def forward(ctx, xb):
        return to_plain(xb)

    @staticmethod
    def backward(ctx, g):
        return to_blocked(g)


def to_plain_tap(t):
    """tap feature -> NCHW (differentiable); tap 5 already is."""
    return _ToPlain.apply(t) if t.dim() == 5 else t


def _sfx(t):
    """entry-point suffix for the storage type of a feature map (the fp32 and the bf16 L1 kernels)"""
    return "_bf16" if t.dtype == torch.bfloat16 else ""


class _L1Pair(Function):
    """(mean|a - p|, mean|a - n|) of one feature tap in one pass; gradient w.r.t. a only (p, n carry none, My_CR.py:102)."""

    @staticmethod
    def forward(ctx, a, p, n):
        a, p = a.contiguous(), p.contiguous()
        n = n.contiguous() if n is not None else None
        sums = torch.zeros(2, device=a.device, dtype=torch.float32)
        _lib.call("dhz_l1_pair_fwd" + _sfx(a), _p(a), _p(p), _p(n) if n is not None else None, _p(sums), a.numel(), _stream())
        ctx.save_for_backward(a, p, n) if n is not None else ctx.save_for_backward(a, p)
        ctx.has_n = n is not None
        return sums / a.numel()

    @staticmethod
    def backward(ctx, g):
        saved = ctx.saved_tensors
        a, p = saved[0], saved[1]
        n = saved[2] if ctx.has_n else None
        da = torch.empty_like(a)
        _lib.call("dhz_l1_pair_bwd" + _sfx(a), _p(a), _p(p), _p(n) if n is not None else None, _p(g.contiguous()), _p(da),
                  a.numel(), _stream())
        return da, None, None


def l1_pair(a, p, n=None):
    return _L1Pair.apply(a, p, n)


_TAP_CONST = {}


class _ContrastTaps(Function):
    """The scalar side of ContrastLoss.forward (My_CR.py:104-123) for ALL taps as one autograd node:
        loss = sum_i w_i * ap_i / (an_i + 1e-7)   (ablation: sum_i w_i * ap_i),  all_ap = sum_i ap_i,  all_an = sum_i an_i
    with ap_i / an_i the L1 means of tap i (dhz_l1_pair_fwd, one pass over a, p, n per tap).  As separate autograd scalars the
    five taps cost ~100 launches of 4-8 us per step (0-dim add / div / mul / neg, select_backward, zero fills); here the
    combination and the ten backward coefficients are one small kernel each way (dhz_contrast_combine_fwd / _bwd)."""

    @staticmethod
    def forward(ctx, weights, ablation, B, *feats):
        k = len(feats) // 2
        a = [f.contiguous() for f in feats[:k]]
        dev = a[0].device
        key = (str(dev), tuple(f.numel() for f in a), tuple(weights))
        if key not in _TAP_CONST:
            _TAP_CONST[key] = (torch.tensor([1.0 / f.numel() for f in a], dtype=torch.float32).view(k, 1).to(dev),
                               torch.tensor(list(weights), dtype=torch.float32).to(dev))
        inv_cnt, w = _TAP_CONST[key]
        sums = ops.zeros_f32((k, 2), dev)          # (accumulation target of this pass: the optimizer's zeroed scratch when there is one)
        ps, ns = [], []
        for i in range(k):
            p = feats[k + i][:B].contiguous()
            n = None if ablation else feats[k + i][B:].contiguous()
            _lib.call("dhz_l1_pair_fwd" + _sfx(a[i]), _p(a[i]), _p(p), _p(n) if n is not None else None, sums.data_ptr() + 8 * i,
                      a[i].numel(), _stream())
            ps.append(p)
            ns.append(n)
        d = torch.empty((k, 2), device=dev, dtype=torch.float32)       # [k, 2] means
        out = torch.empty(3, device=dev, dtype=torch.float32)
        _lib.call("dhz_contrast_combine_fwd", _p(sums), _p(inv_cnt), _p(w), k, int(ablation), _p(d), _p(out), _stream())
        ctx.set_materialize_grads(False)
        ctx.k, ctx.ablation = k, ablation
        ctx.save_for_backward(d, w, *a, *ps, *([] if ablation else ns))
        return out[0], out[1], out[2]

    @staticmethod
    def backward(ctx, g_loss, g_ap, g_an):
        k, ablation = ctx.k, ctx.ablation
        saved = ctx.saved_tensors
        d, w = saved[0], saved[1]
        a, ps = saved[2:2 + k], saved[2 + k:2 + 2 * k]
        ns = [None] * k if ablation else saved[2 + 2 * k:2 + 3 * k]
        g = torch.empty((k, 2), device=d.device, dtype=torch.float32)          # d loss / d (ap_i, an_i)
        # converted copies stay referenced until the launch is enqueued (a temporary's pointer would dangle when the incoming
        # gradient is not already a contiguous fp32 scalar)
        gl, gp, gn = (None if t is None else t.to(torch.float32).contiguous() for t in (g_loss, g_ap, g_an))
        _lib.call("dhz_contrast_combine_bwd", _p(d), _p(w), k, int(ablation), None if gl is None else _p(gl),
                  None if gp is None else _p(gp), None if gn is None else _p(gn), _p(g), _stream())
        del gl, gp, gn
        das = []
        for i in range(k):
            da = torch.empty_like(a[i])
            _lib.call("dhz_l1_pair_bwd" + _sfx(a[i]), _p(a[i]), _p(ps[i]), _p(ns[i]) if ns[i] is not None else None, g.data_ptr() + 8 * i,
                      _p(da), a[i].numel(), _stream())
            das.append(da)
        return (None, None, None) + tuple(das) + (None,) * k


def contrast_taps(a_taps, pn_taps, B, weights, ablation):
    """(loss, all_ap, all_an) of ContrastLoss from the feature taps of a (grad) and of cat([p, n]) / p (no grad)."""
    return _ContrastTaps.apply(tuple(weights), bool(ablation), int(B), *a_taps, *pn_taps)
