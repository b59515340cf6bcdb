"""CPU ORACLE (test infrastructure - NOT product code).

A functional, plain-PyTorch (CPU, fp32/fp64) restatement of the reference's Uformer_ProbSparse
training forward path.  Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline`
leg may import this module; the product package never does (it fails loudly without its HIP
library instead of falling back to anything here).

Parity status: PINNED.  Every function below is checked in tests/test_oracle_golden.py against
golden vectors produced by importing the reference itself (tests/golden/gen_golden.py, run in the
build container where /root/reference is mounted).  Exception: the ImageNet VGG19 weights used by
My_CR are a third-party artefact that is not in the reference tree -> the ContrastLoss *formula and
slicing* are pinned (through a seeded-random VGG19), the pretrained weights are "parity unpinned".

Reference citations use the SURVEY abbreviations:
  M1  = Uformer_ProbSparse/My_model_1.py      M0 = Uformer_ProbSparse/My_model.py
  ATT = Uformer_ProbSparse/ProbSparse/attn.py TR = Uformer_ProbSparse/My_train.py

All functions take a flat mapping `P` (name -> tensor) that uses the reference's state_dict keys,
so a product model's `state_dict()` (or `dict(named_parameters())` for autograd) can be fed directly.
"""
import math

import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------- configuration

DEPTHS = (2, 2, 2, 2, 2, 2, 2, 2, 2)
HEADS = (1, 2, 4, 8, 16, 16, 8, 4, 2)          # M1:962
STAGE_NAMES = ("encoderlayer_0", "encoderlayer_1", "encoderlayer_2", "encoderlayer_3", "conv",
               "decoderlayer_0", "decoderlayer_1", "decoderlayer_2", "decoderlayer_3")


def drop_path_schedule(rate=0.1, depths=DEPTHS):
    """Per-block DropPath probabilities, M1:984-986 (enc linspace, bottleneck constant, dec reversed)."""
    n_enc = sum(depths[:4])
    enc = [x.item() for x in torch.linspace(0, rate, n_enc)]
    conv = [rate] * depths[4]
    dec = enc[::-1]
    out, e, d = [], 0, 0
    for s in range(9):
        if s < 4:
            out.append(enc[e:e + depths[s]]); e += depths[s]
        elif s == 4:
            out.append(conv)
        else:
            out.append(dec[d:d + depths[s]]); d += depths[s]
    return out


def n_top(L, factor=5):
    """u = U_part = min(L, factor*ceil(ln L)) - ATT:310-315 (25 for L=64)."""
    return min(L, factor * int(math.ceil(math.log(L))))


# ----------------------------------------------------------------------------- window helpers

def relative_position_index(win):
    """[N,N] int64 index into the (2w-1)^2 bias table - M1:366-381."""
    c = torch.arange(win)
    hh, ww = torch.meshgrid(c, c, indexing="ij")
    coords = torch.stack([hh.reshape(-1), ww.reshape(-1)])            # 2,N
    rel = coords[:, :, None] - coords[:, None, :] + (win - 1)         # 2,N,N  in [0, 2w-2]
    return rel[0] * (2 * win - 1) + rel[1]


def window_partition(x, win):
    """[B,H,W,C] -> [B*nW, win*win, C] (row-major windows) - M1:550-574, dilation branch unused."""
    B, H, W, C = x.shape
    x = x.reshape(B, H // win, win, W // win, win, C).permute(0, 1, 3, 2, 4, 5)
    return x.reshape(-1, win * win, C)


def window_reverse(w, win, H, W):
    """inverse of window_partition - M1:577-601."""
    C = w.shape[-1]
    x = w.reshape(-1, H // win, W // win, win, win, C).permute(0, 1, 3, 2, 4, 5)
    return x.reshape(-1, H, W, C)


def shift_attn_mask(H, W, win, shift, dtype=torch.float32):
    """[nW,N,N] with 0 where two tokens of a (shifted) window come from the same image region and
    -100 otherwise - M1:803-836 (9-region labelling, outer difference, masked_fill)."""
    lab = torch.zeros(H, W, dtype=dtype)
    bounds_h = (slice(0, -win), slice(-win, -shift), slice(-shift, None))
    cnt = 0
    for hs in bounds_h:
        for ws in bounds_h:
            lab[hs, ws] = cnt
            cnt += 1
    lw = window_partition(lab.reshape(1, H, W, 1), win).squeeze(-1)   # nW,N
    diff = lw[:, None, :] - lw[:, :, None]
    return torch.where(diff != 0, torch.full_like(diff, -100.0), torch.zeros_like(diff))


# ----------------------------------------------------------------------------- attention cores

def prob_attention(q, k, v, idx, bias=None, mask=None, return_aux=False):
    """ProbSparse attention core, ATT:287-342 with _prob_QK (ATT:71-152), _get_initial_context
    (ATT:154-176) and _update_context (ATT:178-281).

    q,k,v : [B_,H,N,d]      idx : [N,u] int64 sampled key ids (shared by all B_,H - ATT:91)
    bias  : [H,N,N] or None (None <=> options.is_relative_position_bias False, ATT:227-232)
    mask  : [nW,N,N] (0/-100) or None; window id of row b is b mod nW (ATT:246-261)
    returns ctx [B_,H,N,d]
    """
    B_, H, N, d = q.shape
    u = idx.shape[1]
    # sampled scores S[b,h,i,s] = q_i . k_{idx[i,s]}   (unscaled, ATT:104-110)
    ks = k[:, :, idx, :]                                              # B_,H,N,u,d
    S = torch.matmul(q.unsqueeze(-2), ks.transpose(-2, -1)).squeeze(-2)
    Mq = S.max(-1)[0] - S.sum(-1) / N                                 # divides by L_K=N, ATT:117
    top = Mq.topk(u, sorted=False)[1]                                 # B_,H,u   (non-differentiable)
    bi = torch.arange(B_)[:, None, None]
    hi = torch.arange(H)[None, :, None]
    q_red = q[bi, hi, top]                                            # B_,H,u,d
    scores = torch.matmul(q_red, k.transpose(-2, -1)) * (1.0 / math.sqrt(d))   # ATT:150,327-329
    ctx = v.mean(dim=-2, keepdim=True).expand(B_, H, N, d).clone()    # ATT:168-172
    a = torch.softmax(scores, dim=-1)                                 # first softmax ATT:195
    if bias is not None:
        a = a + bias[hi, top]                                         # bias added to PROBABILITIES ATT:229
    if mask is not None:
        nW = mask.shape[0]
        wi = (torch.arange(B_) % nW)[:, None, None]
        a = a + mask[wi, top]                                         # ATT:251-258
    a = torch.softmax(a, dim=-1)                                      # second softmax ATT:262/264
    ctx[bi, hi, top] = torch.matmul(a, v)                             # ATT:271-272
    if return_aux:
        return ctx, top, Mq, scores, a
    return ctx


def dense_attention(q, k, v, bias, mask, scale):
    """Dense window attention of the My_model twin - M0:428-492."""
    B_, H, N, d = q.shape
    attn = (q * scale) @ k.transpose(-2, -1)
    attn = attn + bias.unsqueeze(0)
    if mask is not None:
        nW = mask.shape[0]
        attn = attn.view(B_ // nW, nW, H, N, N) + mask[None, :, None]
        attn = attn.view(B_, H, N, N)
    attn = torch.softmax(attn, dim=-1)
    return attn @ v


def gather_bias(P, pre, win, heads):
    """table[(2w-1)^2,H][index] -> [H,N,N] - M1:408-410."""
    N = win * win
    ridx = relative_position_index(win).reshape(-1)
    return P[pre + "relative_position_bias_table"][ridx].reshape(N, N, heads).permute(2, 0, 1).contiguous()


def window_attention(xw, P, pre, heads, win, variant, idx, mask, use_bias=True):
    """WindowAttention.forward: M1:400-415 (-> AttentionLayer ATT:385-461) or dense M0:428-518.
    xw: [B_,N,C].  `pre` is the key prefix ending in 'attn.'."""
    B_, N, C = xw.shape
    d = C // heads
    bias = gather_bias(P, pre, win, heads)
    if variant == "probsparse":
        L = pre + "ProbSpare."
        def proj(name):
            return F.linear(xw, P[L + name + ".weight"], P[L + name + ".bias"]).view(B_, N, heads, d).transpose(1, 2)
        q, k, v = proj("query_projection"), proj("key_projection"), proj("value_projection")
        ctx = prob_attention(q, k, v, idx, bias if use_bias else None, mask)
        ctx = ctx.transpose(1, 2).reshape(B_, N, C)
        return F.linear(ctx, P[L + "out_projection.weight"], P[L + "out_projection.bias"])
    # dense twin: LinearProjection to_q / to_kv (M0:264-300) then proj
    q = F.linear(xw, P[pre + "qkv.to_q.weight"], P[pre + "qkv.to_q.bias"]).view(B_, N, heads, d).transpose(1, 2)
    kv = F.linear(xw, P[pre + "qkv.to_kv.weight"], P[pre + "qkv.to_kv.bias"]).view(B_, N, 2, heads, d)
    k, v = kv[:, :, 0].transpose(1, 2), kv[:, :, 1].transpose(1, 2)
    out = dense_attention(q, k, v, bias, mask, d ** -0.5).transpose(1, 2).reshape(B_, N, C)
    return F.linear(out, P[pre + "proj.weight"], P[pre + "proj.bias"])


# ----------------------------------------------------------------------------- block pieces

def leff(x, P, pre):
    """LeFF.forward - M1:496-534: Linear+GELU -> depthwise 3x3 + GELU -> Linear.  x: [B,HW,C]."""
    B, L, C = x.shape
    hh = int(math.sqrt(L))
    h = F.gelu(F.linear(x, P[pre + "linear1.0.weight"], P[pre + "linear1.0.bias"]))
    hid = h.shape[-1]
    h = h.transpose(1, 2).reshape(B, hid, hh, hh)
    h = F.gelu(F.conv2d(h, P[pre + "dwconv.0.weight"], P[pre + "dwconv.0.bias"], padding=1, groups=hid))
    h = h.flatten(2).transpose(1, 2)
    return F.linear(h, P[pre + "linear2.0.weight"], P[pre + "linear2.0.bias"])


def mlp_ffn(x, P, pre):
    """Mlp.forward - M1:456-468 (token_mlp = 'ffn'): fc1 -> GELU -> fc2 (the Dropouts are identities at drop_rate 0)."""
    return F.linear(F.gelu(F.linear(x, P[pre + "fc1.weight"], P[pre + "fc1.bias"])), P[pre + "fc2.weight"], P[pre + "fc2.bias"])


def _drop_path(x, p, training):
    """timm DropPath semantics: per-sample keep mask, scaled by 1/keep."""
    if p == 0.0 or not training:
        return x
    keep = 1.0 - p
    r = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
    if keep > 0:
        r.div_(keep)
    return x * r


def lewin_block(x, P, pre, heads, win=8, shift=0, variant="probsparse", idx=None, drop_path=0.0,
                training=False, use_bias=True, input_resolution=None, input_mask=None):
    """LeWinTransformerBlock.forward - M1:785-875.  x: [B,HW,C].  `pre` ends with '.' (e.g.
    'encoderlayer_0.blocks.1.').  win/shift clamp of M1:764-766 uses `input_resolution` (the
    *constructor* resolution, which differs from the runtime one in whole-image eval)."""
    B, L, C = x.shape
    H = W = int(math.sqrt(L))
    res = input_resolution if input_resolution is not None else H
    if res <= win:
        shift, win = 0, res
    mask = shift_attn_mask(H, W, win, shift, x.dtype) if shift > 0 else None
    if input_mask is not None:                                          # input-mask path, M1:791-800 (+ :833-836)
        im = F.interpolate(input_mask, size=(H, W)).permute(0, 2, 3, 1)
        am = window_partition(im, win).view(-1, win * win)
        am = am.unsqueeze(2) * am.unsqueeze(1)
        am = am.masked_fill(am != 0, float(-100.0)).masked_fill(am == 0, float(0.0))
        mask = am + mask if mask is not None else am
    shortcut = x
    y = F.layer_norm(x, (C,), P[pre + "norm1.weight"], P[pre + "norm1.bias"], 1e-5).view(B, H, W, C)
    if shift > 0:
        y = torch.roll(y, shifts=(-shift, -shift), dims=(1, 2))
    xw = window_partition(y, win)
    if variant == "probsparse" and idx is None:
        N = win * win
        idx = torch.randint(N, (N, n_top(N)))                          # global CPU generator, ATT:91
    aw = window_attention(xw, P, pre + "attn.", heads, win, variant, idx, mask, use_bias)
    y = window_reverse(aw, win, H, W)
    if shift > 0:
        y = torch.roll(y, shifts=(shift, shift), dims=(1, 2))
    x = shortcut + _drop_path(y.reshape(B, L, C), drop_path, training)
    xn2 = F.layer_norm(x, (C,), P[pre + "norm2.weight"], P[pre + "norm2.bias"], 1e-5)
    z = mlp_ffn(xn2, P, pre + "mlp.") if (pre + "mlp.fc1.weight") in P else leff(xn2, P, pre + "mlp.")      # M1:778-779
    return x + _drop_path(z, drop_path, training)


def tokens_to_map(x):
    B, L, C = x.shape
    s = int(math.sqrt(L))
    return x.transpose(1, 2).reshape(B, C, s, s)


def input_proj(img, P):
    """Conv3x3 + LeakyReLU(0.01) -> tokens  - M1:677-682."""
    y = F.leaky_relu(F.conv2d(img, P["input_proj.proj.0.weight"], P["input_proj.proj.0.bias"], padding=1), 0.01)
    return y.flatten(2).transpose(1, 2)


def output_proj(x, P):
    """tokens -> Conv3x3 -> image - M1:715-723."""
    return F.conv2d(tokens_to_map(x), P["output_proj.proj.0.weight"], P["output_proj.proj.0.bias"], padding=1)


def downsample(x, P, name):
    """Conv k4 s2 p1 - M1:615-622."""
    y = F.conv2d(tokens_to_map(x), P[name + ".conv.0.weight"], P[name + ".conv.0.bias"], stride=2, padding=1)
    return y.flatten(2).transpose(1, 2)


def upsample(x, P, name):
    """ConvTranspose k2 s2 - M1:642-648."""
    y = F.conv_transpose2d(tokens_to_map(x), P[name + ".deconv.0.weight"], P[name + ".deconv.0.bias"], stride=2)
    return y.flatten(2).transpose(1, 2)


# ----------------------------------------------------------------------------- whole model

def uformer_forward(P, img, variant="probsparse", img_size=128, win=8, drop_path_rate=0.1, training=False,
                    idx_seq=None, use_bias=True, depths=DEPTHS, heads=HEADS, mask=None):
    """Uformer.forward - M1:1169-1207.  `idx_seq` (optional): [18,N,u] sampled-key indices in block
    order; None => each block draws from the global CPU generator exactly where the reference does."""
    dpr = drop_path_schedule(drop_path_rate, depths)
    blk_counter = [0]

    def stage(x, s, res):
        for i in range(depths[s]):
            idx = None
            if idx_seq is not None:
                idx = idx_seq[blk_counter[0]]
            blk_counter[0] += 1
            x = lewin_block(x, P, f"{STAGE_NAMES[s]}.blocks.{i}.", heads[s], win, 0 if i % 2 == 0 else win // 2,
                            variant, idx, dpr[s][i], training, use_bias, input_resolution=res, input_mask=mask)
        return x

    y = input_proj(img, P)
    skips = []
    for s in range(4):
        y = stage(y, s, img_size // (2 ** s))
        skips.append(y)
        y = downsample(y, P, f"dowsample_{s}")
    y = stage(y, 4, img_size // 16)
    for s in range(4):
        y = upsample(y, P, f"upsample_{s}")
        y = torch.cat([y, skips[3 - s]], -1)
        y = stage(y, 5 + s, img_size // (2 ** (3 - s)))
    return img + output_proj(y, P)


# ----------------------------------------------------------------------------- losses

def charbonnier(x, y, eps=1e-3):
    """losses.py:48-52."""
    d = x - y
    return torch.mean(torch.sqrt(d * d + eps * eps))


VGG19_CFG = (64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M', 512)
# feature taps: relu1_1, relu2_1, relu3_1, relu4_1, relu5_1  = outputs of features[1],[6],[11],[20],[29]
VGG_TAPS = (0, 2, 4, 8, 12)      # conv ordinal (0-based) after whose ReLU a feature map is emitted
CR_WEIGHTS = (1.0 / 32, 1.0 / 16, 1.0 / 8, 1.0 / 4, 1.0)


def vgg19_features(x, W):
    """Vgg19.forward - My_CR.py:79-86 with the slice boundaries of :65-74 (features[0:30]).
    W: list of 13 (weight,bias) tuples.  NO ImageNet mean/std normalisation (My_CR.py:99-102)."""
    feats, ci = [], 0
    for v in VGG19_CFG:
        if v == 'M':
            x = F.max_pool2d(x, 2, 2)
        else:
            x = F.relu(F.conv2d(x, W[ci][0], W[ci][1], padding=1))
            if ci in VGG_TAPS:
                feats.append(x)
            ci += 1
    return feats


def contrast_loss(a, p, n, W, ablation=False):
    """ContrastLoss.forward - My_CR.py:99-123.  returns (loss, all_ap, all_an)."""
    fa, fp, fn = vgg19_features(a, W), vgg19_features(p, W), vgg19_features(n, W)
    loss, all_ap, all_an = 0, 0, 0
    for i in range(5):
        d_ap = F.l1_loss(fa[i], fp[i].detach())
        all_ap = all_ap + d_ap
        if not ablation:
            d_an = F.l1_loss(fa[i], fn[i].detach())
            all_an = all_an + d_an
            c = d_ap / (d_an + 1e-7)
        else:
            c = d_ap
        loss = loss + CR_WEIGHTS[i] * c
    return loss, all_ap, all_an


def seeded_vgg_weights(seed=1905, dtype=torch.float32):
    """Seeded random VGG19[:30] conv weights (He-normal weights, 0.05*N(0,1) biases) - same recipe as
    the golden generator's torchvision shim (the real ImageNet checkpoint is unavailable offline)."""
    g = torch.Generator().manual_seed(seed)
    W, cin = [], 3
    full = (64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M', 512, 512, 512, 512, 'M')
    for v in full:
        if v == 'M':
            continue
        w = torch.randn(v, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
        b = torch.randn(v, generator=g) * 0.05
        W.append((w.to(dtype), b.to(dtype)))
        cin = v
    return W[:13]


# ----------------------------------------------------------------------------- train step (TR:212-250)

def train_step_loss(P, hazy, gt, variant="probsparse", w_char=1.0, w_cr=0.0, vggW=None, training=True,
                    idx_seq=None, drop_path_rate=0.1):
    """Forward + loss of one training step (TR:227-238): clamp(restored,0,1), Charbonnier and
    (optionally) the contrastive term, weighted sum."""
    restored = torch.clamp(uformer_forward(P, hazy, variant, training=training, idx_seq=idx_seq,
                                           drop_path_rate=drop_path_rate), 0, 1)
    loss = 0
    if w_char > 0:
        loss = loss + w_char * charbonnier(restored, gt)
    if w_cr > 0:
        loss = loss + w_cr * contrast_loss(restored, gt, hazy, vggW)[0]
    return loss, restored


def psnr(a, b):
    """PSNR with data_range=1 (skimage semantics for float images in [0,1]) - TR:189,282."""
    mse = torch.mean((a.double() - b.double()) ** 2)
    return float(10.0 * torch.log10(1.0 / mse))
