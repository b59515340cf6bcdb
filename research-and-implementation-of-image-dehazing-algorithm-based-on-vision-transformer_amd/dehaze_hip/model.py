"""Uformer (ProbSparse and dense twins) on the MI355X kernels.

The module TREE, parameter registration order, initialisation stream and `state_dict` keys are those
of the reference (Uformer_ProbSparse/My_model_1.py and My_model.py) so checkpoints and optimizer
states are interchangeable; the COMPUTE inside every block runs through dehaze_hip.ops (hand-written HIP
kernels - token GEMMs, projections and resampling convolutions included; the library convolution is reached only
for shapes the kernels do not tile, with a one-time warning naming the shape).

Reference map (M1 = My_model_1.py, M0 = My_model.py, ATT = ProbSparse/attn.py):
  Uformer M1:955-1207 | BasicUformerLayer M1:894-946 | LeWinTransformerBlock M1:738-875 |
  WindowAttention M1:336-415 / M0:428-518 | AttentionLayer ATT:345-461 | ProbAttention ATT:43-342 |
  LeFF M1:477-534 | Downsample M1:606-622 | Upsample M1:633-648 | InputProj M1:659-682 |
  OutputProj M1:696-723
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import fused, ops

# ----------------------------------------------------------------------------- small reference-surface helpers


def to_2tuple(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x, x)


def trunc_normal_(t, mean=0., std=1., a=-2., b=2.):
    return nn.init.trunc_normal_(t, mean=mean, std=std, a=a, b=b)


class DropPath(nn.Module):
    """Per-sample stochastic depth (timm semantics).  `sample_scale` returns the [B] keep/keep_prob
    vector that the fused residual kernels consume; forward() keeps the stand-alone behaviour."""

    def __init__(self, drop_prob=0., scale_by_keep=True):
        super().__init__()
        self.drop_prob = drop_prob
        self.scale_by_keep = scale_by_keep

    def sample_scale(self, x):
        if self.drop_prob == 0. or not self.training:
            return None
        keep = 1 - self.drop_prob
        # always fp32 (the kernels read `const float* scale`; with bf16 token storage x.new_empty would hand them a bf16 vector)
        r = torch.empty((x.shape[0],), device=x.device, dtype=torch.float32).bernoulli_(keep)
        if keep > 0 and self.scale_by_keep:
            r.div_(keep)
        return r

    def forward(self, x):
        s = self.sample_scale(x)
        return x if s is None else x * s.to(x.dtype).view((-1,) + (1,) * (x.ndim - 1))


def n_top(L, factor=5):
    """u = U_part = min(L, factor*ceil(ln L))  (ATT:310-315)."""
    return min(L, factor * int(math.ceil(math.log(L))))


def draw_sample_index(n_blocks=1, L=ops.NTOK):
    """Sampled-key table(s) from the GLOBAL CPU generator, exactly like ATT:91 (`torch.randint(L_K,
    (L_Q, sample_k))`); a batched draw consumes the stream identically to consecutive draws."""
    return torch.randint(L, (n_blocks, L, n_top(L)))


def window_partition(x, win_size, dilation_rate=1):
    """[B,H,W,C] -> [B*nW, win, win, C]  (reference helper M1:550-574; kept for API parity - the model
    itself never materialises this, the partition is folded into the LN kernel's store addresses)."""
    assert dilation_rate == 1, "dilated windows are unused by the reference path"
    B, H, W, C = x.shape
    x = x.reshape(B, H // win_size, win_size, W // win_size, win_size, C)
    return x.permute(0, 1, 3, 2, 4, 5).reshape(-1, win_size, win_size, C)


def window_reverse(windows, win_size, H, W, dilation_rate=1):
    """inverse of window_partition (M1:577-601)."""
    assert dilation_rate == 1
    C = windows.shape[-1]
    x = windows.reshape(-1, H // win_size, W // win_size, win_size, win_size, C)
    return x.permute(0, 1, 3, 2, 4, 5).reshape(-1, H, W, C)


# ----------------------------------------------------------------------------- attention

class ProbAttention(nn.Module):
    """Parameter-free holder mirroring ATT:43-69 (the dropout it owns is never applied, ATT:68)."""

    def __init__(self, mask_flag=False, factor=5, scale=None, attention_dropout=0.1, output_attention=False):
        super().__init__()
        self.factor, self.scale, self.mask_flag, self.output_attention = factor, scale, mask_flag, output_attention
        self.dropout = nn.Dropout(attention_dropout)
        self.softmax = nn.Softmax(dim=-1)


class AttentionLayer(nn.Module):
    """Q/K/V/out projections around the ProbSparse core (ATT:345-461)."""

    def __init__(self, d_model, n_heads, d_keys=None, d_values=None, mix=False):
        super().__init__()
        d_keys = d_keys or (d_model // n_heads)
        d_values = d_values or (d_model // n_heads)
        self.inner_attention = ProbAttention(False, 5, None, 0.1, False)
        self.query_projection = nn.Linear(d_model, d_keys * n_heads)
        self.key_projection = nn.Linear(d_model, d_keys * n_heads)
        self.value_projection = nn.Linear(d_model, d_values * n_heads)
        self.out_projection = nn.Linear(d_values * n_heads, d_model)
        self.n_heads = n_heads
        self.mix = mix

    def forward(self, queries, keys, values, table, SW_mask, attn_mask=None, idx=None):
        """queries (= keys = values): [B_, 64, C] window tokens.  `table` is the [225,H] bias table (or
        None when options.is_relative_position_bias is False); returns ([B_,64,C], None)."""
        assert keys is queries and values is queries, "self-attention only (M1:413 passes x, x, x)"
        B_, N, C = queries.shape
        H = self.n_heads
        qp, kp, vp = self.query_projection, self.key_projection, self.value_projection
        qkv = ops.linear_tokens(queries.reshape(B_ * N, C), qp.weight, qp.bias, kp.weight, kp.bias,
                                vp.weight, vp.bias)                              # one library GEMM [T,3C]
        if idx is None:
            idx = draw_sample_index(1, N)[0]
        if idx.device != qkv.device or idx.dtype != torch.uint8:
            idx = idx.to(device=qkv.device, dtype=torch.uint8)
        ctx = ops.ps_window_attention(qkv, table, idx.contiguous(), SW_mask, H, C // H)
        out = ops.linear_tokens(ctx, self.out_projection.weight, self.out_projection.bias)
        return out.view(B_, N, C), None


class LinearProjection(nn.Module):
    """to_q / to_kv of the dense twin (M0:264-300); present-but-dead in the ProbSparse model (M1:389)."""

    def __init__(self, dim, heads=8, dim_head=64, dropout=0., bias=True):
        super().__init__()
        inner = dim_head * heads
        self.heads = heads
        self.to_q = nn.Linear(dim, inner, bias=bias)
        self.to_kv = nn.Linear(dim, inner * 2, bias=bias)
        self.dim, self.inner_dim = dim, inner

    def forward(self, x, attn_kv=None):
        B_, N, C = x.shape
        attn_kv = x if attn_kv is None else attn_kv
        q = self.to_q(x).reshape(B_, N, 1, self.heads, C // self.heads).permute(2, 0, 3, 1, 4)[0]
        kv = self.to_kv(attn_kv).reshape(B_, N, 2, self.heads, C // self.heads).permute(2, 0, 3, 1, 4)
        return q, kv[0], kv[1]


# ---- the off-default token projections / squeeze-excite of M1:167-335.  In the ProbSparse model WindowAttention.forward (M1:400-415) calls
#      ONLY self.ProbSpare: self.qkv, self.proj and self.se_layer are registered, initialised and checkpointed but never run ("dead"
#      parameters).  So `--token_projection conv | linear_concat` and `se_layer=True` change the state_dict and the init
#      RNG stream of that model, not its function: the modules below exist for exactly that (same attribute names, shapes, creation
#      order); their forward() is plain torch and off the accelerated path.  The dense twin (My_model.Uformer) RUNS its qkv / proj
#      projection: there only token_projection = 'linear' without se_layer is implemented.
class SELayer(nn.Module):
    """M1:167-185: channel gate (mean over tokens -> C/16 -> C -> sigmoid)."""

    def __init__(self, channel, reduction=16):
        super().__init__()
        self.avg_pool = nn.AdaptiveAvgPool1d(1)
        self.fc = nn.Sequential(nn.Linear(channel, channel // reduction, bias=False), nn.ReLU(inplace=True),
                                nn.Linear(channel // reduction, channel, bias=False), nn.Sigmoid())

    def forward(self, x):                                   # [B, N, C]
        return x * self.fc(x.mean(dim=1)).unsqueeze(1)


class SepConv2d(nn.Module):
    """M1:188-221: depthwise k x k convolution, activation, 1 x 1 convolution."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, act_layer=nn.ReLU):
        super().__init__()
        self.depthwise = nn.Conv2d(in_channels, in_channels, kernel_size=kernel_size, stride=stride, padding=padding,
                                   dilation=dilation, groups=in_channels)
        self.pointwise = nn.Conv2d(in_channels, out_channels, kernel_size=1)
        self.act_layer = act_layer() if act_layer is not None else nn.Identity()
        self.in_channels, self.out_channels, self.kernel_size, self.stride = in_channels, out_channels, kernel_size, stride

    def forward(self, x):
        return self.pointwise(self.act_layer(self.depthwise(x)))


class ConvProjection(nn.Module):
    """M1:226-254: q / k / v from three separable convolutions over the window's 8 x 8 token map.  (The reference hands `bias` to
    SepConv2d's sixth positional parameter, which is `dilation`: True -> 1.  Kept.)"""

    def __init__(self, dim, heads=8, dim_head=64, kernel_size=3, q_stride=1, k_stride=1, v_stride=1, dropout=0.,
                 last_stage=False, bias=True):
        super().__init__()
        inner = dim_head * heads
        self.heads = heads
        pad = (kernel_size - q_stride) // 2
        self.to_q = SepConv2d(dim, inner, kernel_size, q_stride, pad, int(bias))
        self.to_k = SepConv2d(dim, inner, kernel_size, k_stride, pad, int(bias))
        self.to_v = SepConv2d(dim, inner, kernel_size, v_stride, pad, int(bias))

    def forward(self, x, attn_kv=None):
        b, n, c = x.shape
        side = int(math.sqrt(n))
        kv = x if attn_kv is None else attn_kv
        to_map = lambda t: t.transpose(1, 2).reshape(b, c, side, side)
        heads = lambda t: t.flatten(2).reshape(b, self.heads, -1, n).transpose(2, 3)
        return heads(self.to_q(to_map(x))), heads(self.to_k(to_map(kv))), heads(self.to_v(to_map(kv)))


class LinearProjection_Concat_kv(nn.Module):
    """M1:309-331: q, k, v from one Linear, a second k, v from another, keys / values concatenated along the tokens."""

    def __init__(self, dim, heads=8, dim_head=64, dropout=0., bias=True):
        super().__init__()
        inner = dim_head * heads
        self.heads, self.dim, self.inner_dim = heads, dim, inner
        self.to_qkv = nn.Linear(dim, inner * 3, bias=bias)
        self.to_kv = nn.Linear(dim, inner * 2, bias=bias)

    def forward(self, x, attn_kv=None):
        B_, N, C = x.shape
        kv_in = x if attn_kv is None else attn_kv
        q, kd, vd = self.to_qkv(x).reshape(B_, N, 3, self.heads, C // self.heads).permute(2, 0, 3, 1, 4)
        ke, ve = self.to_kv(kv_in).reshape(B_, N, 2, self.heads, C // self.heads).permute(2, 0, 3, 1, 4)
        return q, torch.cat((kd, ke), dim=2), torch.cat((vd, ve), dim=2)


def relative_position_index(win):
    c = torch.arange(win)
    hh, ww = torch.meshgrid(c, c, indexing="ij")
    coords = torch.stack([hh.reshape(-1), ww.reshape(-1)])
    rel = coords[:, :, None] - coords[:, None, :] + (win - 1)
    return rel[0] * (2 * win - 1) + rel[1]


class WindowAttention(nn.Module):
    """M1:336-415 (variant='probsparse') / M0:428-518 (variant='dense')."""

    def __init__(self, dim, win_size, num_heads, token_projection='linear', qkv_bias=True, qk_scale=None,
                 attn_drop=0., proj_drop=0., se_layer=False, variant="probsparse"):
        super().__init__()
        if variant != "probsparse" and (token_projection != 'linear' or se_layer):
            # the dense twin RUNS qkv / proj / se_layer (M0:428-518): only the options.py defaults are implemented there
            raise NotImplementedError("My_model (dense attention): only token_projection='linear', se_layer=False are implemented")
        self.dim, self.win_size, self.num_heads, self.variant = dim, win_size, num_heads, variant
        head_dim = dim // num_heads
        self.scale = qk_scale or head_dim ** -0.5
        if variant == "probsparse":
            self.ProbSpare = AttentionLayer(dim, num_heads)
        self.relative_position_bias_table = nn.Parameter(
            torch.zeros((2 * win_size[0] - 1) * (2 * win_size[1] - 1), num_heads))
        self.register_buffer("relative_position_index", relative_position_index(win_size[0]))
        if token_projection == 'conv':                                       # M1:384-390 (dead in the ProbSparse forward: see above)
            self.qkv = ConvProjection(dim, num_heads, dim // num_heads, bias=qkv_bias)
        elif token_projection == 'linear_concat':
            self.qkv = LinearProjection_Concat_kv(dim, num_heads, dim // num_heads, bias=qkv_bias)
        else:
            self.qkv = LinearProjection(dim, num_heads, dim // num_heads, bias=qkv_bias)
        self.token_projection = token_projection
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.se_layer = SELayer(dim) if se_layer else nn.Identity()          # M1:394
        self.proj_drop = nn.Dropout(proj_drop)
        trunc_normal_(self.relative_position_bias_table, std=.02)
        self.softmax = nn.Softmax(dim=-1)

    def forward(self, x, attn_kv=None, mask=None, idx=None):
        if self.win_size[0] * self.win_size[1] != ops.NTOK:
            raise NotImplementedError("the HIP kernels are specialised to 8x8 windows")
        if self.variant == "probsparse":
            import options                                         # read at call time, like ATT:227
            table = self.relative_position_bias_table if options.is_relative_position_bias else None
            out, _ = self.ProbSpare(x, x, x, table, mask, idx=idx)
            return out
        return self._dense(x, mask)

    def _dense(self, x, mask):
        """M0:428-518: q = to_q(x), [k|v] = to_kv(x) as ONE packed GEMM, fused dense window attention, proj."""
        B_, N, C = x.shape
        H = self.num_heads
        qkv = ops.linear_tokens(x.reshape(B_ * N, C), self.qkv.to_q.weight, self.qkv.to_q.bias,
                                self.qkv.to_kv.weight, self.qkv.to_kv.bias)
        ctx = ops.dense_window_attention(qkv, self.relative_position_bias_table, mask, H, C // H, self.scale)
        return ops.linear_tokens(ctx, self.proj.weight, self.proj.bias).view(B_, N, C)

    def extra_repr(self):
        return f'dim={self.dim}, win_size={self.win_size}, num_heads={self.num_heads}'


# ----------------------------------------------------------------------------- feed-forward

class Mlp(nn.Module):
    """M1:442-468 (token_mlp = 'ffn', the constructor default of Uformer; options.py selects 'leff'): fc1 -> GELU -> fc2; the two
    Dropouts are identities at the model's drop_rate = 0 (kept for the attribute names)."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)
        self.in_features, self.hidden_features, self.out_features = in_features, hidden_features, out_features
        # the HIP path of this branch (fused._FfnBranch, Mlp.forward) evaluates exact-erf GELU and no dropout: what the reference runs at
        # the model's defaults.  Anything else would silently diverge from M1:442-468 - refuse it
        if drop > 0. or act_layer is not nn.GELU:
            raise NotImplementedError(f"Mlp(act_layer={getattr(act_layer, '__name__', act_layer)}, drop={drop}): the HIP Mlp branch implements "
                                      "nn.GELU with drop = 0 (the reference's defaults) only")

    def forward(self, x):
        B, L, C = x.shape
        u = ops.linear_tokens(x.reshape(B * L, C), self.fc1.weight, self.fc1.bias)
        z = ops.gelu_tokens(u)
        return ops.linear_tokens(z, self.fc2.weight, self.fc2.bias).view(B, L, self.out_features)


class LeFF(nn.Module):
    """M1:477-534.  Linear+GELU -> depthwise3x3+GELU -> Linear; the middle stage runs in token (NHWC)
    layout in one HIP kernel, so neither GELU nor the NCHW rearranges touch HBM."""

    def __init__(self, dim=32, hidden_dim=128, act_layer=nn.GELU, drop=0.):
        super().__init__()
        self.linear1 = nn.Sequential(nn.Linear(dim, hidden_dim), act_layer())
        self.dwconv = nn.Sequential(nn.Conv2d(hidden_dim, hidden_dim, groups=hidden_dim, kernel_size=3, stride=1, padding=1),
                                    act_layer())
        self.linear2 = nn.Sequential(nn.Linear(hidden_dim, dim))
        self.dim, self.hidden_dim = dim, hidden_dim

    def forward(self, x):
        B, L, C = x.shape
        hh = int(math.sqrt(L))
        u = ops.linear_tokens(x.reshape(B * L, C), self.linear1[0].weight, self.linear1[0].bias).view(B, L, self.hidden_dim)
        z = ops.leff_dwconv(u, self.dwconv[0].weight, self.dwconv[0].bias, hh, hh)
        y = ops.linear_tokens(z.view(B * L, self.hidden_dim), self.linear2[0].weight, self.linear2[0].bias)
        return y.view(B, L, C)


# ----------------------------------------------------------------------------- block

class LeWinTransformerBlock(nn.Module):
    """M1:738-875."""

    def __init__(self, dim, input_resolution, num_heads, win_size=8, shift_size=0, mlp_ratio=4., qkv_bias=True,
                 qk_scale=None, drop=0., attn_drop=0., drop_path=0., act_layer=nn.GELU, norm_layer=nn.LayerNorm,
                 token_projection='linear', token_mlp='leff', se_layer=False, variant="probsparse"):
        super().__init__()
        if token_mlp not in ('leff', 'ffn'):
            raise Exception("FFN error!")                                   # M1:780-781
        self.dim, self.input_resolution, self.num_heads = dim, input_resolution, num_heads
        self.win_size, self.shift_size, self.mlp_ratio, self.token_mlp = win_size, shift_size, mlp_ratio, token_mlp
        if min(self.input_resolution) <= self.win_size:                     # M1:764-766
            self.shift_size = 0
            self.win_size = min(self.input_resolution)
        assert 0 <= self.shift_size < self.win_size, "shift_size must in 0-win_size"
        self.norm1 = norm_layer(dim)
        self.attn = WindowAttention(dim, win_size=to_2tuple(self.win_size), num_heads=num_heads, qkv_bias=qkv_bias,
                                    qk_scale=qk_scale, attn_drop=attn_drop, proj_drop=drop,
                                    token_projection=token_projection, se_layer=se_layer, variant=variant)
        self.drop_path = DropPath(drop_path) if drop_path > 0. else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop) if token_mlp == 'ffn' \
            else LeFF(dim, int(dim * mlp_ratio), act_layer=act_layer, drop=drop)                              # M1:778-779
        self._mask_cache = {}
        self._staged_idx = None      # set by Uformer.forward (one batched host draw per model forward)
        self._staged_scales = None   # set by Uformer.forward on the GPU: the two DropPath vectors of this block

    def extra_repr(self):
        return (f"dim={self.dim}, input_resolution={self.input_resolution}, num_heads={self.num_heads}, "
                f"win_size={self.win_size}, shift_size={self.shift_size}, mlp_ratio={self.mlp_ratio}")

    def _shift_mask(self, H, W, device):
        key = (H, W, str(device))
        m = self._mask_cache.get(key)
        if m is None:
            m = ops.shift_mask(H, W, self.shift_size, device)
            self._mask_cache[key] = m
        return m

    def _scale(self, x):
        if self._staged_scales:                      # drawn for the whole model in one call (Uformer._stage_drop_path)
            return self._staged_scales.pop(0)
        return self.drop_path.sample_scale(x) if isinstance(self.drop_path, DropPath) else None

    def forward(self, x, mask=None):
        B, L, C = x.shape
        H = W = int(math.sqrt(L))
        attn_mask = None
        if mask is not None:                                                 # input-mask path, M1:791-800
            im = F.interpolate(mask, size=(H, W)).permute(0, 2, 3, 1)
            am = window_partition(im, self.win_size).view(-1, self.win_size * self.win_size)
            am = am.unsqueeze(2) * am.unsqueeze(1)
            attn_mask = am.masked_fill(am != 0, float(-100.0)).masked_fill(am == 0, float(0.0))
        if self.shift_size > 0:
            sm = self._shift_mask(H, W, x.device)
            attn_mask = attn_mask + sm if attn_mask is not None else sm
        idx, self._staged_idx = self._staged_idx, None

        if self.attn.variant == "probsparse" and self.win_size == 8 and C in (16 * self.num_heads, 32 * self.num_heads, 64 * self.num_heads) and mask is None:
            # attention branch as ONE autograd node: the fused kernel (LN, roll, partition, QKV, ProbSparse core,
            # out-proj, residual) where it wins, the kernel chain elsewhere; hand-sequenced backward in both cases
            import options
            table = self.attn.relative_position_bias_table if options.is_relative_position_bias else None
            if idx is None:
                idx = draw_sample_index(1, ops.NTOK)[0]
            if idx.device != x.device or idx.dtype != torch.uint8:
                idx = idx.to(device=x.device, dtype=torch.uint8)
            if fused.BLOCK_NODE and self.token_mlp == 'leff':
                # both branches as one node: the gradient between them never takes the token-order detour (fused.block)
                return fused.block(x, self.norm1, self.attn.ProbSpare, table, idx.contiguous(), attn_mask, self._scale(x), H, W,
                                   self.shift_size, self.num_heads, self.norm2, self.mlp, self._scale(x))
            x = fused.attn_branch(x, self.norm1, self.attn.ProbSpare, table, idx.contiguous(), attn_mask,
                                  self._scale(x), H, W, self.shift_size, self.num_heads)
        else:
            xw = ops.ln_partition(x, self.norm1.weight, self.norm1.bias, H, W, self.shift_size)   # LN+roll+partition
            aw = self.attn(xw.view(-1, self.win_size * self.win_size, C), mask=attn_mask, idx=idx)
            x = ops.reverse_residual(aw.reshape(-1, C), x, self._scale(x), H, W, self.shift_size)  # reverse+unroll+res
        if self.token_mlp == 'ffn':
            # Mlp branch (norm2 -> fc1 -> GELU -> fc2 -> residual) as one autograd node
            return fused.ffn_branch(x, self.norm2, self.mlp, self._scale(x))
        # LeFF branch (norm2 -> linear1 -> dwconv -> linear2 -> residual) as one autograd node
        return fused.leff_branch(x, self.norm2, self.mlp, self._scale(x), H, W)


class BasicUformerLayer(nn.Module):
    """M1:894-946."""

    def __init__(self, dim, output_dim, input_resolution, depth, num_heads, win_size, mlp_ratio=4., qkv_bias=True,
                 qk_scale=None, drop=0., attn_drop=0., drop_path=0., norm_layer=nn.LayerNorm, use_checkpoint=False,
                 token_projection='linear', token_mlp='ffn', se_layer=False, variant="probsparse"):
        super().__init__()
        self.dim, self.input_resolution, self.depth, self.use_checkpoint = dim, input_resolution, depth, use_checkpoint
        self.blocks = nn.ModuleList([
            LeWinTransformerBlock(dim=dim, input_resolution=input_resolution, num_heads=num_heads, win_size=win_size,
                                  shift_size=0 if (i % 2 == 0) else win_size // 2, mlp_ratio=mlp_ratio,
                                  qkv_bias=qkv_bias, qk_scale=qk_scale, drop=drop, attn_drop=attn_drop,
                                  drop_path=drop_path[i] if isinstance(drop_path, list) else drop_path,
                                  norm_layer=norm_layer, token_projection=token_projection, token_mlp=token_mlp,
                                  se_layer=se_layer, variant=variant)
            for i in range(depth)])

    def extra_repr(self):
        return f"dim={self.dim}, input_resolution={self.input_resolution}, depth={self.depth}"

    def forward(self, x, mask=None):
        for blk in self.blocks:
            if self.use_checkpoint:
                x = torch.utils.checkpoint.checkpoint(blk, x)
            else:
                x = blk(x, mask)
        return x


# ----------------------------------------------------------------------------- resampling / projections

def _tokens_to_map(x):
    """[B, H*W, C] tokens -> logical NCHW map that ALIASES the token buffer (channels_last strides): the
    token layout is NHWC already, so no transpose kernel runs and MIOpen picks its NHWC kernels directly."""
    B, L, C = x.shape
    s = int(math.sqrt(L))
    return x.contiguous().view(B, s, s, C).permute(0, 3, 1, 2)


def _map_to_tokens(y):
    """NCHW-logical conv output (channels_last in memory) -> [B, H*W, C] tokens, copy-free when possible."""
    B, C, H, W = y.shape
    return y.permute(0, 2, 3, 1).reshape(B, H * W, C)


class Downsample(nn.Module):
    """Conv k4 s2 p1 on the token map (M1:606-622)."""

    def __init__(self, in_channel, out_channel):
        super().__init__()
        self.conv = nn.Sequential(nn.Conv2d(in_channel, out_channel, kernel_size=4, stride=2, padding=1))
        self.in_channel, self.out_channel = in_channel, out_channel

    def forward(self, x):
        conv = self.conv[0]
        s_ = int(math.sqrt(x.shape[1]))
        need_grad = torch.is_grad_enabled() and (x.requires_grad or conv.weight.requires_grad)
        if s_ * s_ == x.shape[1] and ops.conv4s2_supported(x, s_, s_, need_grad) and self.out_channel % 32 == 0:
            # implicit GEMM on the fp32 matrix pipe straight from / to the token layout (csrc/conv_gemm.hip)
            return ops.conv4s2_tokens(x, conv.weight, conv.bias, s_, s_)
        if (s_ * s_ == x.shape[1] and x.dtype == torch.bfloat16 and ops.conv4s2_bf16_supported(x, s_, s_)
                and self.out_channel % 64 == 0):
            # config 4: patch matrix + the bf16-MFMA token-Linear GEMMs (csrc/conv_bf16.hip), fp32 master weights
            return ops.conv4s2_tokens(x, conv.weight, conv.bias, s_, s_)
        ho = s_ // 2
        if (s_ * s_ == x.shape[1] and x.is_cuda and x.dtype == torch.float32 and self.in_channel % 32 == 16 and self.out_channel % 32 == 0
                and s_ % 2 == 0 and (not need_grad or ((ho & (ho - 1)) == 0 and (x.shape[0] * ho * ho) % 32 == 0))):
            # 16 (mod 32) input channels (the embed_dim = 16 model's first down-sampling): the implicit-GEMM kernels tile the contraction in
            # 32-channel stages, so the tokens and the filters get 16 zero channels (two small copies; autograd slices the gradients back)
            return ops.conv4s2_tokens(F.pad(x, (0, 16)), F.pad(conv.weight, (0, 0, 0, 0, 0, 16)), conv.bias, s_, s_)
        if x.is_cuda:
            ops.warn_library_fallback("Downsample", (self.in_channel, self.out_channel, tuple(x.shape[1:])))
        if x.dtype == torch.bfloat16:       # shapes the kernels do not tile: library convolution in bf16
            with torch.autocast("cuda", dtype=torch.bfloat16):
                return _map_to_tokens(self.conv(_tokens_to_map(x)))
        return _map_to_tokens(self.conv(_tokens_to_map(x)))


class _ShuffleConcat(torch.autograd.Function):
    """cat([pixel_shuffle(y), skip], -1) for the decoder (M1:1192-1204): y [B*s*s, (a, b, o)] is the token-Linear form of
    the 2x2/stride-2 transposed convolution, skip [B, 4*s*s, Co].  Two strided copies straight into the concatenated
    buffer instead of a shuffle copy followed by a cat (and one strided copy instead of two on the way back)."""

    @staticmethod
    def forward(ctx, y, skip, B, s, Co):
        Cs = skip.shape[-1]
        out = torch.empty((B, 4 * s * s, Co + Cs), device=y.device, dtype=y.dtype)
        out.view(B, s, 2, s, 2, Co + Cs)[..., :Co].copy_(y.view(B, s, s, 2, 2, Co).permute(0, 1, 3, 2, 4, 5))
        out[..., Co:].copy_(skip)
        ctx.dims = (B, s, Co, Cs)
        return out

    @staticmethod
    def backward(ctx, g):
        B, s, Co, Cs = ctx.dims
        gy = torch.empty((B * s * s, 4 * Co), device=g.device, dtype=g.dtype)
        gy.view(B, s, s, 2, 2, Co).copy_(g.view(B, s, 2, s, 2, Co + Cs)[..., :Co].permute(0, 1, 3, 2, 4, 5))
        return gy, g[..., Co:], None, None, None


class Upsample(nn.Module):
    """ConvTranspose k2 s2 (M1:633-648)."""

    def __init__(self, in_channel, out_channel):
        super().__init__()
        self.deconv = nn.Sequential(nn.ConvTranspose2d(in_channel, out_channel, kernel_size=2, stride=2))
        self.in_channel, self.out_channel = in_channel, out_channel

    def forward(self, x, skip=None):
        """skip: optional [B, 4L, Cs] tensor to concatenate behind the up-sampled tokens (the decoder's skip connection)."""
        if not x.is_cuda:
            y = _map_to_tokens(self.deconv(_tokens_to_map(x)))
            return y if skip is None else torch.cat([y, skip], -1)
        # kernel 2 / stride 2: every input token produces its own 2x2 output pixels and nothing overlaps, so the layer IS
        # a token Linear K = Cin -> N = 4*Cout followed by a pixel shuffle: one library GEMM (forward / dgrad) + the
        # split-T weight-gradient kernel instead of MIOpen's implicit-GEMM transposed convolution (~2.3x slower here).
        B, L, Cin = x.shape
        s = int(math.sqrt(L))
        Co = self.out_channel
        dc = self.deconv[0]
        w4 = dc.weight.permute(2, 3, 1, 0).reshape(4 * Co, Cin)            # rows (a, b, o): y[(2i+a, 2j+b), o]
        b4 = dc.bias.repeat(4)
        y = ops.linear_tokens(x.reshape(B * L, Cin), w4, b4)                # [B*s*s, (a, b, o)]
        if skip is not None and (y.requires_grad or skip.requires_grad or not torch.is_grad_enabled()):
            return _ShuffleConcat.apply(y, skip.contiguous(), B, s, Co)
        y = y.view(B, s, s, 2, 2, Co).permute(0, 1, 3, 2, 4, 5)             # [B, i, a, j, b, o]
        y = y.reshape(B, 4 * L, Co)
        return y if skip is None else torch.cat([y, skip], -1)


class InputProj(nn.Module):
    """Conv3x3 + LeakyReLU -> tokens (M1:659-682)."""

    def __init__(self, in_channel=3, out_channel=64, kernel_size=3, stride=1, norm_layer=None, act_layer=nn.LeakyReLU):
        super().__init__()
        self.proj = nn.Sequential(nn.Conv2d(in_channel, out_channel, kernel_size=3, stride=stride, padding=kernel_size // 2),
                                  act_layer(inplace=True))
        self.norm = norm_layer(out_channel) if norm_layer is not None else None
        self.in_channel, self.out_channel = in_channel, out_channel

    def forward(self, x):
        conv, act = self.proj[0], self.proj[1]
        if (x.is_cuda and x.dtype == torch.float32 and self.in_channel == 3 and self.out_channel in (16, 32, 64)
                and conv.stride == (1, 1) and isinstance(act, nn.LeakyReLU) and not x.requires_grad):
            # convolution + LeakyReLU straight into the token layout (csrc/input_proj.hip)
            # (out_dtype: bf16 tokens straight from the kernel when the model runs BASELINE config 4's storage type)
            x = ops.input_proj(x, conv.weight, conv.bias, act.negative_slope, getattr(self, "out_dtype", torch.float32))
        else:
            x = _map_to_tokens(self.proj(x.contiguous(memory_format=torch.channels_last)))
        return self.norm(x) if self.norm is not None else x


class OutputProj(nn.Module):
    """tokens -> Conv3x3 -> image (M1:696-723)."""

    def __init__(self, in_channel=64, out_channel=3, kernel_size=3, stride=1, norm_layer=None, act_layer=None):
        super().__init__()
        self.proj = nn.Sequential(nn.Conv2d(in_channel, out_channel, kernel_size=3, stride=stride, padding=kernel_size // 2))
        if act_layer is not None:
            self.proj.add_module("act", act_layer(inplace=True))
        self.norm = norm_layer(out_channel) if norm_layer is not None else None
        self.in_channel, self.out_channel = in_channel, out_channel

    def forward(self, x):
        conv = self.proj[0]
        if (x.is_cuda and len(self.proj) == 1 and self.out_channel == 3 and self.in_channel == 32 and conv.stride == (1, 1)
                and x.dtype == torch.float32):
            # 32 input channels (the embed_dim = 16 model): the thin-convolution kernels work in 64-channel chunks - 32 zero channels on the
            # tokens and the filters (two small copies; autograd slices the gradients back)
            s_ = int(math.sqrt(x.shape[1]))
            x = ops.thin_conv3x3(F.pad(x, (0, 32)), F.pad(conv.weight, (0, 0, 0, 0, 0, 32)), conv.bias, s_, s_)
            return self.norm(x) if self.norm is not None else x
        if (x.is_cuda and len(self.proj) == 1 and self.out_channel == 3 and self.in_channel in (64, 128)
                and conv.stride == (1, 1) and x.dtype in (torch.float32, torch.bfloat16)):
            # 3 output channels: nothing for the matrix pipe, the library's implicit GEMM runs at 0.6 TB/s - hand-written
            # forward / backward-data / weight-gradient kernels on the token layout (csrc/thin_conv.hip); fp32 or bf16 tokens in,
            # fp32 image out
            s_ = int(math.sqrt(x.shape[1]))
            x = ops.thin_conv3x3(x, conv.weight, conv.bias, s_, s_)
        elif x.dtype == torch.bfloat16:
            ops.warn_library_fallback("OutputProj (bf16)", (self.in_channel, self.out_channel, tuple(conv.stride)))
            with torch.autocast("cuda", dtype=torch.bfloat16):
                x = self.proj(_tokens_to_map(x)).float()
        elif x.is_cuda and len(self.proj) == 1 and conv.bias is not None:
            ops.warn_library_fallback("OutputProj", (self.in_channel, self.out_channel, tuple(conv.stride)))
            # the convolution without its bias + an explicit bias add whose backward sums in two stages: the library's
            # bias gradient of a 3-channel map is ONE 4-block reduction over the whole gradient image (190 us per step)
            x = _BiasAddMap.apply(F.conv2d(_tokens_to_map(x), conv.weight, None, conv.stride, conv.padding), conv.bias)
        else:
            x = self.proj(_tokens_to_map(x))
        return self.norm(x) if self.norm is not None else x


class _BiasAddMap(torch.autograd.Function):
    """y[b,c,h,w] = x[b,c,h,w] + bias[c]; d(bias) = sum over (h,w) per (b,c), then over b."""

    @staticmethod
    def forward(ctx, x, bias):
        return x + bias.view(1, -1, 1, 1)

    @staticmethod
    def backward(ctx, g):
        return g, g.flatten(2).sum(2).sum(0)


# ----------------------------------------------------------------------------- the model

class Uformer(nn.Module):
    """U-shaped encoder/decoder of LeWin blocks (M1:955-1207); constructor signature of M1:961-967."""

    variant = "probsparse"
    # Storage type of the token tensors between kernels: torch.float32 (BASELINE configs 1-3, 5) or torch.bfloat16 (config 4:
    # bf16 activations and bf16 weight copies in HBM, fp32 accumulation, fp32 LayerNorm statistics / sparsity measure / softmax,
    # fp32 master parameters and parameter gradients - the role fp16 autocast + GradScaler play in the reference, TR:224,249;
    # bf16 keeps fp32's exponent range, so no loss scaling).  Set on the instance: model.act_dtype = torch.bfloat16.
    act_dtype = torch.float32

    def __init__(self, img_size=128, in_chans=3, embed_dim=32, depths=[2, 2, 2, 2, 2, 2, 2, 2, 2],
                 num_heads=[1, 2, 4, 8, 16, 16, 8, 4, 2], win_size=8, mlp_ratio=4., qkv_bias=True, qk_scale=None,
                 drop_rate=0., attn_drop_rate=0., drop_path_rate=0.1, norm_layer=nn.LayerNorm, patch_norm=True,
                 use_checkpoint=False, token_projection='linear', token_mlp='ffn', se_layer=False,
                 dowsample=Downsample, upsample=Upsample, **kwargs):
        super().__init__()
        self.num_enc_layers = self.num_dec_layers = len(depths) // 2
        self.embed_dim, self.patch_norm, self.mlp_ratio = embed_dim, patch_norm, mlp_ratio
        self.token_projection, self.mlp, self.win_size, self.reso = token_projection, token_mlp, win_size, img_size
        self.pos_drop = nn.Dropout(p=drop_rate)
        enc_dpr = [x.item() for x in torch.linspace(0, drop_path_rate, sum(depths[:self.num_enc_layers]))]
        conv_dpr = [drop_path_rate] * depths[4]
        dec_dpr = enc_dpr[::-1]

        def layer(i, dim, res, dpr):
            return BasicUformerLayer(dim=dim, output_dim=dim, input_resolution=(res, res), depth=depths[i],
                                     num_heads=num_heads[i], win_size=win_size, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias,
                                     qk_scale=qk_scale, drop=drop_rate, attn_drop=attn_drop_rate, drop_path=dpr,
                                     norm_layer=norm_layer, use_checkpoint=use_checkpoint,
                                     token_projection=token_projection, token_mlp=token_mlp, se_layer=se_layer,
                                     variant=self.variant)

        self.input_proj = InputProj(in_channel=in_chans, out_channel=embed_dim, kernel_size=3, stride=1, act_layer=nn.LeakyReLU)
        self.output_proj = OutputProj(in_channel=2 * embed_dim, out_channel=in_chans, kernel_size=3, stride=1)
        for s in range(4):                                                   # encoder: dim E*2^s at res/2^s
            lo, hi = sum(depths[:s]), sum(depths[:s + 1])
            setattr(self, f"encoderlayer_{s}", layer(s, embed_dim * 2 ** s, img_size // 2 ** s, enc_dpr[lo:hi]))
            setattr(self, f"dowsample_{s}", dowsample(embed_dim * 2 ** s, embed_dim * 2 ** (s + 1)))
        self.conv = layer(4, embed_dim * 16, img_size // 16, conv_dpr)
        up_io = [(16, 8), (16, 4), (8, 2), (4, 1)]                           # M1:1082,1098,1114,1130
        dec_dim = [16, 8, 4, 2]
        for s in range(4):
            lo, hi = sum(depths[5:5 + s]), sum(depths[5:6 + s])
            setattr(self, f"upsample_{s}", upsample(embed_dim * up_io[s][0], embed_dim * up_io[s][1]))
            setattr(self, f"decoderlayer_{s}", layer(5 + s, embed_dim * dec_dim[s], img_size // 2 ** (3 - s), dec_dpr[lo:hi]))
        self.apply(self._init_weights)

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    @torch.jit.ignore
    def no_weight_decay(self):
        return {'absolute_pos_embed'}

    @torch.jit.ignore
    def no_weight_decay_keywords(self):
        return {'relative_position_bias_table'}

    def extra_repr(self):
        return (f"embed_dim={self.embed_dim}, token_projection={self.token_projection}, token_mlp={self.mlp},"
                f"win_size={self.win_size}")

    def stages(self):
        names = [f"encoderlayer_{s}" for s in range(4)] + ["conv"] + [f"decoderlayer_{s}" for s in range(4)]
        return [getattr(self, n) for n in names]

    def live_parameters(self):
        """Parameters that receive gradients (the reference keeps 108 dead tensors - attn.qkv.*, attn.proj.* -
        in parameters() and in the optimizer; they are excluded from gradient buckets)."""
        dead = ("attn.qkv.", "attn.proj.", "attn.se_layer.") if self.variant == "probsparse" else ()
        return [(n, p) for n, p in self.named_parameters() if not any(d in n for d in dead)]

    def _stage_sample_indices(self, device):
        """One host draw for all ProbSparse blocks of this forward (same CPU-generator stream as the
        reference's 18 per-block draws), one H2D copy, handed to the blocks in execution order."""
        blocks = [b for st in self.stages() for b in st.blocks]
        if self.variant != "probsparse":
            return
        idx = draw_sample_index(len(blocks)).to(torch.uint8)
        idx = idx.pin_memory().to(device, non_blocking=True) if device.type == "cuda" else idx
        for i, b in enumerate(blocks):
            b._staged_idx = idx[i]

    def _stage_drop_path(self, x):
        """All DropPath keep/keep_prob vectors of one training forward (two per block with drop_prob > 0) from ONE
        bernoulli launch + one division instead of two tiny launches per residual (~70 launches of ~4 us per step).  GPU
        only: on the CPU every residual keeps its own draw, in the order the reference consumes the CPU generator (the
        golden training trajectory pins that order); the GPU's Philox stream has no reference counterpart to match."""
        blocks = [b for st in self.stages() for b in st.blocks]
        for b in blocks:
            b._staged_scales = None
        if not (self.training and x.is_cuda):
            return
        live = [b for b in blocks if isinstance(b.drop_path, DropPath) and b.drop_path.drop_prob > 0.]
        if not live:
            return
        B = x.shape[0]
        key = (B, str(x.device), tuple(b.drop_path.drop_prob for b in live))
        if getattr(self, "_keep_key", None) != key:
            keep = torch.tensor([1.0 - b.drop_path.drop_prob for b in live for _ in range(2)], dtype=torch.float32)
            self._keep_mat = keep.view(-1, 1).expand(-1, B).contiguous().to(x.device)
            self._keep_inv = (1.0 / keep).view(-1, 1).to(x.device)
            self._keep_key = key
        r = torch.bernoulli(self._keep_mat)
        r.mul_(self._keep_inv)                       # scale_by_keep (timm default)
        for i, b in enumerate(live):
            b._staged_scales = [r[2 * i], r[2 * i + 1]]

    def _stage_block_operands(self, x, mask):
        """The relative-position bias tiles of all blocks and the fragment-ordered weights of the fused attention blocks of this forward in
        ONE launch each (fused.stage_block_operands) - 26 launches of ~4.5 us per training step otherwise."""
        if not x.is_cuda or self.variant != "probsparse":
            return
        import options
        rel = options.is_relative_position_bias
        Himg, Wimg = x.shape[-2], x.shape[-1]
        scales = [1, 2, 4, 8, 16, 8, 4, 2]
        entries = []
        for st, sc in zip(self.stages(), scales + [1]):
            for b in st.blocks:
                if b.attn.variant != "probsparse" or b.win_size != 8 or mask is not None \
                        or b.dim not in (16 * b.num_heads, 32 * b.num_heads, 64 * b.num_heads):
                    continue
                lay = b.attn.ProbSpare
                fused_fwd = self.act_dtype != torch.bfloat16 and fused.ENABLED and b.dim == 32 * b.num_heads and \
                    (b.dim in (32, 64) or (b.dim == 128 and fused.fused_c128_ok((Himg // sc) * (Wimg // sc))))
                w = (lay.query_projection.weight, lay.key_projection.weight, lay.value_projection.weight, lay.out_projection.weight) \
                    if fused_fwd else None
                entries.append((b.attn.relative_position_bias_table if rel else None, b.num_heads, w, b.dim))
        if entries:
            fused.stage_block_operands(entries, x.device)

    def forward(self, x, mask=None):
        ops.sync_shadows()          # derived weight copies (bf16 / split planes) follow parameters written outside the optimizer
        self._stage_sample_indices(x.device)
        self._stage_drop_path(x)
        self._stage_block_operands(x, mask)
        self.input_proj.out_dtype = self.act_dtype if x.is_cuda else torch.float32
        y = self.pos_drop(self.input_proj(x))
        if self.act_dtype == torch.bfloat16 and y.is_cuda and y.dtype != torch.bfloat16:
            y = y.to(torch.bfloat16)
        skips = []
        for s in range(4):
            y = getattr(self, f"encoderlayer_{s}")(y, mask=mask)
            skips.append(y)
            y = getattr(self, f"dowsample_{s}")(y)
        y = self.conv(y, mask=mask)
        for s in range(4):
            y = getattr(self, f"upsample_{s}")(y, skips[3 - s])
            y = getattr(self, f"decoderlayer_{s}")(y, mask=mask)
        return x + self.output_proj(y).float()


class UformerDense(Uformer):
    variant = "dense"
