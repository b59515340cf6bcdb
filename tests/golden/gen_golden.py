#!/usr/bin/env python3
"""Golden-vector generator: imports the REFERENCE (read-only, /root/reference) in the build container
and dumps small input/output fixtures next to this script.

This script only runs where /root/reference exists (the build container).  Nothing here ships to the
GPU box except the .npz files it writes.  The fixtures are data (inputs, expected outputs) - no reference
source text is stored.

Harness shims (none of them is product code):
  * timm.models.layers.{DropPath,to_2tuple,trunc_normal_}  - SURVEY Appendix A semantics
  * torchvision.models.vgg19  - a cfg-'E' feature stack with SEEDED RANDOM weights (the ImageNet
    checkpoint vgg19-dcbb9e9d.pth is a third-party artefact that is not in the reference tree and
    there is no network => ContrastLoss parity is pinned on the loss formula / slicing, NOT on the
    pretrained weights: "parity unpinned" for the weights themselves)
  * nn.Module.cuda -> identity (My_CR.py:94 hard-codes .cuda())

usage:  python tests/golden/gen_golden.py
"""
import hashlib
import os
import random
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/Uformer_ProbSparse"


# ----------------------------------------------------------------------------- shims
def install_shims():
    sys.dont_write_bytecode = True
    L = types.ModuleType("timm.models.layers")
    L.to_2tuple = lambda x: tuple(x) if isinstance(x, (tuple, list)) else (x, x)
    L.trunc_normal_ = lambda t, mean=0., std=1., a=-2., b=2.: nn.init.trunc_normal_(t, mean=mean, std=std, a=a, b=b)

    class DropPath(nn.Module):
        def __init__(self, p=0., scale_by_keep=True):
            super().__init__()
            self.drop_prob, self.scale_by_keep = p, scale_by_keep

        def forward(self, x):
            if self.drop_prob == 0. or not self.training:
                return x
            keep = 1 - self.drop_prob
            r = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
            if keep > 0 and self.scale_by_keep:
                r.div_(keep)
            return x * r

    L.DropPath = DropPath
    sys.modules.update({"timm": types.ModuleType("timm"), "timm.models": types.ModuleType("timm.models"),
                        "timm.models.layers": L})

    # fake torchvision: vgg19 cfg 'E' features, seeded random weights
    tv = types.ModuleType("torchvision")
    tvm = types.ModuleType("torchvision.models")

    def vgg19(pretrained=False):
        cfg = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M', 512, 512, 512, 512, 'M']
        layers, cin = [], 3
        for v in cfg:
            if v == 'M':
                layers.append(nn.MaxPool2d(2, 2))
            else:
                layers += [nn.Conv2d(cin, v, 3, padding=1), nn.ReLU(inplace=True)]
                cin = v
        net = types.SimpleNamespace()
        g = torch.Generator().manual_seed(1905)
        feats = nn.Sequential(*layers)
        for m in feats:
            if isinstance(m, nn.Conv2d):
                fan_in = m.weight.shape[1] * 9
                m.weight.data = torch.randn(m.weight.shape, generator=g) * (2.0 / fan_in) ** 0.5
                m.bias.data = torch.randn(m.bias.shape, generator=g) * 0.05
        net.features = feats
        return net

    tvm.vgg19 = vgg19
    tv.models = tvm
    sys.modules.update({"torchvision": tv, "torchvision.models": tvm})
    nn.Module.cuda = lambda self, *a, **k: self
    sys.path.insert(0, REF)


def npz(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = v
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"  wrote {name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


def seed_all(s):
    random.seed(s)
    np.random.seed(s)
    torch.manual_seed(s)


def ref_shift_mask(M1, res, win=8, shift=4):
    """Run the reference block once at (res,res) just to capture the mask it hands to attention."""
    blk = M1.LeWinTransformerBlock(dim=32, input_resolution=(res, res), num_heads=1, win_size=win, shift_size=shift,
                                   token_mlp='leff')
    cap = {}
    orig = blk.attn.forward
    blk.attn.forward = lambda x, mask=None: (cap.__setitem__('m', mask), orig(x, mask=mask))[1]
    with torch.no_grad():
        blk(torch.zeros(1, res * res, 32))
    return cap['m']


# ----------------------------------------------------------------------------- generators
def gen_rng():
    torch.manual_seed(0)
    a = [torch.randint(64, (64, 25)) for _ in range(3)]
    npz("rng_stream", idx=torch.stack(a).to(torch.int8))


def gen_probattn(M1, ATT, only=None):
    import options
    mask16 = ref_shift_mask(M1, 16)  # [4,64,64] 0/-100
    cases = [("h1_nomask_bias", 8, 1, False, True), ("h2_mask_bias", 4, 2, True, True),
             ("h16_nomask_nobias", 1, 16, False, False), ("h2_mask_nobias", 4, 2, True, False),
             ("h2_mask_bias_d64", 4, 2, True, True), ("h2_mask_bias_d16", 4, 2, True, True)]      # d16: round 5, embed_dim 16
    for name, B_, H, use_mask, use_bias in cases:
        if only is not None and name not in only:
            continue
        d = 64 if name.endswith("d64") else 16 if name.endswith("d16") else 32
        g = torch.Generator().manual_seed(100 + H + 7 * use_mask + 13 * use_bias + d)
        q = torch.randn(B_, 64, H, d, generator=g).requires_grad_()
        k = torch.randn(B_, 64, H, d, generator=g).requires_grad_()
        v = torch.randn(B_, 64, H, d, generator=g).requires_grad_()
        bias = (0.5 * torch.randn(H, 64, 64, generator=g)).requires_grad_()
        gout = torch.randn(B_, 64, H, d, generator=g)
        seed = 4242 + H
        torch.manual_seed(seed)
        idx = torch.randint(64, (64, 25))
        att = ATT.ProbAttention(mask_flag=False, factor=5, scale=None, attention_dropout=0.1, output_attention=False)
        options.is_relative_position_bias = use_bias
        torch.manual_seed(seed)   # the reference draws idx itself from the global CPU generator
        ctx, _ = att(q, k, v, bias, mask16 if use_mask else None, None)
        # intermediate (scores, top index) by calling the reference's own helper with the same seed
        torch.manual_seed(seed)
        with torch.no_grad():
            sc, top = att._prob_QK(q.transpose(2, 1), k.transpose(2, 1), sample_k=25, n_top=25)
        (ctx * gout).sum().backward()
        options.is_relative_position_bias = True
        npz("probattn_" + name, q=q, k=k, v=v, bias=bias, gout=gout, idx=idx.to(torch.int8),
            mask=(mask16 if use_mask else torch.zeros(0)), use_bias=np.int32(use_bias),
            ctx=ctx, top=top.to(torch.int8), scores_top=sc,
            dq=q.grad, dk=k.grad, dv=v.grad, dbias=(bias.grad if bias.grad is not None else torch.zeros(0)))


def sd_arrays(mod, prefix="sd/"):
    return {prefix + k: v for k, v in mod.state_dict().items()}


def grad_arrays(mod, prefix="g/"):
    return {prefix + k: (p.grad if p.grad is not None else torch.zeros(0)) for k, p in mod.named_parameters()}


def gen_blocks(M1, M0):
    for modname, MM in (("m1", M1), ("m0", M0)):
        for shift in (0, 4):
            seed_all(11 + shift)
            blk = MM.LeWinTransformerBlock(dim=32, input_resolution=(16, 16), num_heads=1, win_size=8,
                                           shift_size=shift, token_mlp='leff', drop_path=0.)
            # non-trivial LN affine + biases so that every parameter matters
            g = torch.Generator().manual_seed(5)
            with torch.no_grad():
                for p in blk.parameters():
                    if p.ndim == 1:
                        p.add_(0.1 * torch.randn(p.shape, generator=g))
            x = torch.randn(2, 256, 32, generator=g).requires_grad_()
            gout = torch.randn(2, 256, 32, generator=g)
            torch.manual_seed(77)
            idx = torch.randint(64, (64, 25))
            torch.manual_seed(77)
            y = blk(x)
            (y * gout).sum().backward()
            npz(f"block_{modname}_c32_shift{shift}", x=x, gout=gout, idx=idx.to(torch.int8), y=y, dx=x.grad,
                **sd_arrays(blk), **grad_arrays(blk))
        # multi-head block (C=64, H=2), shifted, 16x16
        seed_all(23)
        blk = MM.LeWinTransformerBlock(dim=64, input_resolution=(16, 16), num_heads=2, win_size=8,
                                       shift_size=4, token_mlp='leff', drop_path=0.)
        g = torch.Generator().manual_seed(6)
        x = torch.randn(1, 256, 64, generator=g).requires_grad_()
        gout = torch.randn(1, 256, 64, generator=g)
        torch.manual_seed(78)
        idx = torch.randint(64, (64, 25))
        torch.manual_seed(78)
        y = blk(x)
        (y * gout).sum().backward()
        npz(f"block_{modname}_c64_shift4", x=x, gout=gout, idx=idx.to(torch.int8), y=y, dx=x.grad,
            **sd_arrays(blk), **grad_arrays(blk))


WIDE_BLOCKS = {     # name -> (C, heads, map side, shift): the widths whose blocks run the kernel chain (C >= 256) or the widest fused instance
    "block_m1_c128_shift4": (128, 4, 16, 4),
    "block_m1_c256_shift4": (256, 8, 16, 4),
    "block_m1_c512_shift0": (512, 16, 8, 0),       # the bottleneck's geometry: one 8 x 8 window per image
    # round 5: head_dim 16 (the embed_dim = 16 model, utils/model_utils.py:96-98 'Uformer16'): its first stage (one head, every Linear
    # 16 wide) and its last decoder stage (C = 32 as TWO heads of 16)
    "block_m1_c16_shift4": (16, 1, 16, 4),
    "block_m1_c32h2_shift4": (32, 2, 16, 4),
    # round 5: token_mlp = 'ffn' (Mlp, M1:442-468) - the constructor default of Uformer / LeWinTransformerBlock's other branch (M1:778-779)
    "block_m1_c64_ffn_shift4": (64, 2, 16, 4),
}


def gen_block_wide(M1, name):
    """round 4: ProbSparse blocks at C = 128 (the widest instance of the fused window-attention forward, csrc/fused_attn.hip), C = 256
    and C = 512 (8 / 16 heads on the kernel chain) - so that they are checked against the REFERENCE block by block, not only through
    the whole-model golden"""
    C, heads, side, shift = WIDE_BLOCKS[name]
    seed_all(31)
    blk = M1.LeWinTransformerBlock(dim=C, input_resolution=(side, side), num_heads=heads, win_size=8, shift_size=shift,
                                   token_mlp='ffn' if "_ffn_" in name else 'leff', drop_path=0.)
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for p in blk.parameters():
            if p.ndim == 1:
                p.add_(0.1 * torch.randn(p.shape, generator=g))
    x = torch.randn(1, side * side, C, generator=g).requires_grad_()
    gout = torch.randn(1, side * side, C, generator=g)
    torch.manual_seed(79)
    idx = torch.randint(64, (64, 25))
    torch.manual_seed(79)
    y = blk(x)
    (y * gout).sum().backward()
    # kept small: the weights, x and gout are NOT stored (the test rebuilds them with this seed recipe - the package's init stream equals the
    # reference's, tests/test_host.py), the parameter gradients as norm + every 97th element
    gs = {}
    for k, p in blk.named_parameters():
        if p.grad is None:
            gs["gn/" + k] = torch.zeros(0)
        else:
            gs["gn/" + k] = p.grad.double().norm().reshape(1)
            gs["gs/" + k] = p.grad.reshape(-1)[::97].clone()
    first = next(iter(blk.parameters()))
    npz(name, x_probe=x.detach().reshape(-1)[:16].clone(), gout_probe=gout.reshape(-1)[:16].clone(),
        idx=idx.to(torch.int8), y=y, dx=x.grad, w_probe=first.detach().reshape(-1)[:16].clone(), **gs)


def gen_block_c128(M1):
    gen_block_wide(M1, "block_m1_c128_shift4")


def gen_masks(M1):
    m16 = ref_shift_mask(M1, 16)
    m128 = ref_shift_mask(M1, 128)
    npz("shift_mask", m16=(m16 != 0).numpy().astype(np.uint8), m128_packed=np.packbits((m128 != 0).numpy()),
        m128_shape=np.array(m128.shape), vals=np.unique(torch.cat([m16.flatten(), m128.flatten()]).numpy()))


def gen_small_modules(M1):
    g = torch.Generator().manual_seed(9)
    out = {}

    def run(tag, mod, x):
        x = x.clone().requires_grad_()
        y = mod(x)
        go = torch.randn(y.shape, generator=g)
        (y * go).sum().backward()
        out.update({f"{tag}/x": x, f"{tag}/y": y, f"{tag}/gout": go, f"{tag}/dx": x.grad})
        out.update(sd_arrays(mod, f"{tag}/sd/"))
        out.update(grad_arrays(mod, f"{tag}/g/"))

    seed_all(31)
    run("leff", M1.LeFF(32, 128), torch.randn(2, 256, 32, generator=g))
    run("down", M1.Downsample(32, 64), torch.randn(2, 256, 32, generator=g))
    run("up", M1.Upsample(64, 32), torch.randn(2, 64, 64, generator=g))
    run("inproj", M1.InputProj(3, 32, 3, 1, act_layer=nn.LeakyReLU), torch.rand(2, 3, 16, 16, generator=g))
    run("outproj", M1.OutputProj(64, 3, 3, 1), torch.randn(2, 256, 64, generator=g))
    npz("small_modules", **out)


def tensor_digest(sd):
    h = hashlib.sha256()
    for k, v in sd.items():
        h.update(k.encode())
        h.update(v.detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()


def gen_full(M1, M0, losses):
    for modname, MM in (("m1", M1), ("m0", M0)):
        seed_all(1234)
        model = MM.Uformer(img_size=128, embed_dim=32, win_size=8, token_projection='linear', token_mlp='leff')
        sd = model.state_dict()
        keys = list(sd.keys())
        stats = np.array([[float(v.double().sum()), float(v.double().abs().sum())] for v in sd.values()])
        pnames = [n for n, _ in model.named_parameters()]
        g = torch.Generator().manual_seed(7)
        gt = torch.rand(1, 3, 128, 128, generator=g)
        hazy = (0.6 * gt + 0.4 * torch.rand(1, 1, 1, 1, generator=g)).clamp(0, 1)
        # inputs are stored as fp16 to keep the fixture small: make them exactly fp16-representable
        gt, hazy = gt.half().float(), hazy.half().float()
        # per-stage activation means via hooks
        acts = {}
        hooks = []
        for name in ["input_proj", "encoderlayer_0", "dowsample_0", "encoderlayer_1", "encoderlayer_2", "encoderlayer_3",
                     "conv", "upsample_0", "decoderlayer_0", "decoderlayer_1", "decoderlayer_2", "decoderlayer_3",
                     "output_proj"]:
            hooks.append(getattr(model, name).register_forward_hook(
                lambda m, i, o, name=name: acts.__setitem__(name, [float(o.double().mean()), float(o.double().abs().mean())])))
        model.eval()
        torch.manual_seed(99)
        with torch.no_grad():
            y_eval = model(hazy)
        for h in hooks:
            h.remove()
        # train-mode (DropPath disabled through p=0 model would change the rng stream; instead keep eval-mode
        # dropout semantics but enable grad): gradients w.r.t. all parameters for a Charbonnier loss
        torch.manual_seed(99)
        y = model(hazy)
        loss = losses.CharbonnierLoss()(torch.clamp(y, 0, 1), gt)
        loss.backward()
        gnorm = np.array([float(p.grad.double().norm()) if p.grad is not None else -1.0 for p in model.parameters()])
        gsum = np.array([float(p.grad.double().sum()) if p.grad is not None else 0.0 for p in model.parameters()])
        npz(f"full_{modname}_e32", keys=np.array(keys), pnames=np.array(pnames), sd_stats=stats,
            sd_sha256=np.array(tensor_digest(sd)), gt=gt.half(), hazy=hazy.half(),
            y_eval_crop=y_eval[0, :, 40:72, 40:72], y_eval_sum=np.float64(y_eval.double().sum()),
            y_eval_abs=np.float64(y_eval.double().abs().sum()), y_eval_lowres=torch.nn.functional.avg_pool2d(y_eval, 4),
            act_names=np.array(list(acts.keys())), act_stats=np.array(list(acts.values())),
            loss=np.float64(loss.item()), gnorm=gnorm, gsum=gsum,
            shapes=np.array([str(tuple(v.shape)) for v in sd.values()]))


def gen_full_e16(M1, losses):
    """round 5: the embed_dim = 16 model that get_arch builds for --arch Uformer16 (utils/model_utils.py:96-98): eval output, loss and
    gradient norms of a Charbonnier step, same recipe as gen_full"""
    seed_all(1234)
    model = M1.Uformer(img_size=128, embed_dim=16, win_size=8, token_projection='linear', token_mlp='leff')
    sd = model.state_dict()
    g = torch.Generator().manual_seed(7)
    gt = torch.rand(1, 3, 128, 128, generator=g)
    hazy = (0.6 * gt + 0.4 * torch.rand(1, 1, 1, 1, generator=g)).clamp(0, 1)
    gt, hazy = gt.half().float(), hazy.half().float()
    model.eval()
    torch.manual_seed(99)
    with torch.no_grad():
        y_eval = model(hazy)
    torch.manual_seed(99)
    y = model(hazy)
    loss = losses.CharbonnierLoss()(torch.clamp(y, 0, 1), gt)
    loss.backward()
    gnorm = np.array([float(p.grad.double().norm()) if p.grad is not None else -1.0 for p in model.parameters()])
    gsum = np.array([float(p.grad.double().sum()) if p.grad is not None else 0.0 for p in model.parameters()])
    npz("full_m1_e16", keys=np.array(list(sd.keys())), sd_sha256=np.array(tensor_digest(sd)), gt=gt.half(), hazy=hazy.half(),
        y_eval_crop=y_eval[0, :, 40:72, 40:72], y_eval_sum=np.float64(y_eval.double().sum()),
        y_eval_abs=np.float64(y_eval.double().abs().sum()), y_eval_lowres=torch.nn.functional.avg_pool2d(y_eval, 4),
        loss=np.float64(loss.item()), gnorm=gnorm, gsum=gsum, shapes=np.array([str(tuple(v.shape)) for v in sd.values()]))


def gen_full_ctor_default(M1, losses):
    """round 5: M1.Uformer() with the CONSTRUCTOR's defaults (M1:961-967: embed_dim 32, token_mlp = 'ffn' - Mlp blocks instead of LeFF;
    options.py / get_arch pass 'leff'): state_dict keys / shapes / init stream, eval output, Charbonnier loss and gradient norms"""
    seed_all(1234)
    model = M1.Uformer()
    sd = model.state_dict()
    stats = np.array([[float(v.double().sum()), float(v.double().abs().sum())] for v in sd.values()])
    g = torch.Generator().manual_seed(7)
    gt = torch.rand(1, 3, 128, 128, generator=g)
    hazy = (0.6 * gt + 0.4 * torch.rand(1, 1, 1, 1, generator=g)).clamp(0, 1)
    gt, hazy = gt.half().float(), hazy.half().float()
    model.eval()
    torch.manual_seed(99)
    with torch.no_grad():
        y_eval = model(hazy)
    torch.manual_seed(99)
    y = model(hazy)
    loss = losses.CharbonnierLoss()(torch.clamp(y, 0, 1), gt)
    loss.backward()
    gnorm = np.array([float(p.grad.double().norm()) if p.grad is not None else -1.0 for p in model.parameters()])
    npz("full_m1_ctor_default", keys=np.array(list(sd.keys())), pnames=np.array([n for n, _ in model.named_parameters()]), sd_stats=stats,
        gt=gt.half(), hazy=hazy.half(), y_eval_crop=y_eval[0, :, 40:72, 40:72], y_eval_sum=np.float64(y_eval.double().sum()),
        y_eval_abs=np.float64(y_eval.double().abs().sum()), loss=np.float64(loss.item()), gnorm=gnorm,
        shapes=np.array([str(tuple(v.shape)) for v in sd.values()]))


def gen_dead_branches(M1):
    """round 5: token_projection = 'conv' / 'linear_concat' and se_layer = True in the ProbSparse model: WindowAttention.forward (M1:400-415)
    never runs self.qkv / self.proj / self.se_layer, so these switches change the state_dict and the init stream, not the function -
    keys, shapes, init statistics and the eval output of the (different) initial weights"""
    g = torch.Generator().manual_seed(7)
    gt = torch.rand(1, 3, 128, 128, generator=g)
    hazy = (0.6 * gt + 0.4 * torch.rand(1, 1, 1, 1, generator=g)).clamp(0, 1).half().float()
    out = {"hazy": hazy.half()}
    for tag, kw in (("conv_se", dict(token_projection='conv', se_layer=True)), ("concat", dict(token_projection='linear_concat'))):
        seed_all(1234)
        model = M1.Uformer(img_size=128, embed_dim=32, win_size=8, token_mlp='leff', **kw)
        sd = model.state_dict()
        out[tag + "/keys"] = np.array(list(sd.keys()))
        out[tag + "/shapes"] = np.array([str(tuple(v.shape)) for v in sd.values()])
        out[tag + "/sd_stats"] = np.array([[float(v.double().sum()), float(v.double().abs().sum())] for v in sd.values()])
        model.eval()
        torch.manual_seed(99)
        with torch.no_grad():
            y = model(hazy)
        out[tag + "/y_eval_crop"] = y[0, :, 40:72, 40:72]
        out[tag + "/y_eval_sum"] = np.float64(y.double().sum())
    npz("dead_branches", **out)


def gen_trajectory(M1, losses):
    seed_all(1234)
    model = M1.Uformer(img_size=128, embed_dim=32, win_size=8, token_projection='linear', token_mlp='leff')
    opt = torch.optim.AdamW(model.parameters(), lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.02)
    g = torch.Generator().manual_seed(7)
    gt = torch.rand(4, 3, 128, 128, generator=g)
    hazy = (0.6 * gt + 0.4 * torch.rand(4, 1, 1, 1, generator=g)).clamp(0, 1)
    crit = losses.CharbonnierLoss()
    model.train()
    traj = []

    def step(a, b):
        opt.zero_grad()
        out = torch.clamp(model(hazy[a:b]), 0, 1)
        loss = crit(out, gt[a:b])
        loss.backward()
        opt.step()
        return loss.item()

    step(0, 2)  # warm step on pair 0
    for ep in range(3):
        for s in range(2):
            traj.append(step(2 * s, 2 * s + 2))
    print("  trajectory:", ["%.5f" % t for t in traj])
    npz("train_trajectory", losses=np.array(traj, dtype=np.float64))


def gen_losses(losses):
    g = torch.Generator().manual_seed(3)
    x = torch.rand(2, 3, 16, 16, generator=g).requires_grad_()
    y = torch.rand(2, 3, 16, 16, generator=g)
    l = losses.CharbonnierLoss()(x, y)
    l.backward()
    out = dict(char_x=x, char_y=y, char_loss=np.float64(l.item()), char_dx=x.grad)
    # ContrastLoss through the reference's own Vgg19 slicing / loss formula over the shimmed feature stack
    import My_CR
    for ab in (False, True):
        cl = My_CR.ContrastLoss(ablation=ab)
        a = torch.rand(2, 3, 32, 32, generator=g).requires_grad_()
        p = torch.rand(2, 3, 32, 32, generator=g)
        n = torch.rand(2, 3, 32, 32, generator=g)
        loss, all_ap, all_an = cl(a, p, n)
        loss.backward()
        tag = "cr_ab" if ab else "cr"
        out.update({f"{tag}/a": a, f"{tag}/p": p, f"{tag}/n": n, f"{tag}/loss": np.float64(loss.item()),
                    f"{tag}/all_ap": np.float64(float(all_ap)), f"{tag}/all_an": np.float64(float(all_an)),
                    f"{tag}/da": a.grad})
        if not ab:
            feats = cl.vgg(a.detach())
            out["cr/feat_stats"] = np.array([[float(f.double().mean()), float(f.double().abs().max())] for f in feats])
            out["cr/feat_shapes"] = np.array([list(f.shape) for f in feats])
            out["cr/vgg_w0"] = cl.vgg.slice1[0].weight.detach()
            out["cr/vgg_wsum"] = np.array([float(pp.double().sum()) for pp in cl.vgg.parameters()])
    npz("losses", **out)


def gen_options():
    import argparse
    import options
    p = options.Options().init(argparse.ArgumentParser())
    d = {a.dest: (str(a.default), str(a.type.__name__ if a.type else None), a.__class__.__name__)
         for a in p._actions if a.dest != "help"}
    npz("options", dests=np.array(list(d.keys())), defaults=np.array([v[0] for v in d.values()]),
        types=np.array([v[1] for v in d.values()]), kinds=np.array([v[2] for v in d.values()]),
        is_relative_position_bias=np.int32(options.is_relative_position_bias))


def gen_misc():
    # MixUp (dataset_utils.py) - .cuda() patched to identity via a tensor-level shim
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_dataset_utils", os.path.join(REF, "utils/dataset_utils.py"))
    du = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(du)
    torch.Tensor.cuda = lambda self, *a, **k: self
    g = torch.Generator().manual_seed(21)
    gt = torch.rand(4, 3, 8, 8, generator=g)
    nz = torch.rand(4, 3, 8, 8, generator=g)
    torch.manual_seed(5)
    mix = du.MixUp_AUG()
    a, b = mix.aug(gt, nz)
    # warmup + cosine LR schedule (per-epoch)
    from warmup_scheduler import GradualWarmupScheduler
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([p], lr=2e-4)
    cos = torch.optim.lr_scheduler.CosineAnnealingLR(opt, 20 - 3, eta_min=1e-6)
    sch = GradualWarmupScheduler(opt, multiplier=1, total_epoch=3, after_scheduler=cos)
    lrs = []
    for ep in range(20):
        lrs.append(opt.param_groups[0]['lr'])
        opt.step()
        sch.step()
    npz("misc", mix_gt=gt, mix_nz=nz, mix_out_gt=a, mix_out_nz=b, lrs=np.array(lrs))


def gen_data():
    """utils/dataset_utils.py of the reference (imported by file path: the package __init__ pulls cv2): the 8 rotate/flip
    augmentations on a seeded tensor and one MixUp draw (torch global RNG seeded; lam.cuda() shimmed to a no-op)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_dataset_utils", os.path.join(REF, "utils", "dataset_utils.py"))
    du = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(du)
    g = torch.Generator().manual_seed(11)
    x = torch.rand(3, 6, 6, generator=g)
    aug = du.Augment_RGB_torch()
    names = [m for m in dir(aug) if callable(getattr(aug, m)) if not m.startswith('_')]
    out = {"x": x, "names": np.array(names)}
    for k, nme in enumerate(names):
        out[f"t{k}"] = getattr(aug, nme)(x).contiguous()
    had = hasattr(torch.Tensor, "_orig_cuda")
    torch.Tensor._orig_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        gt, nz = torch.rand(5, 3, 4, 4, generator=g), torch.rand(5, 3, 4, 4, generator=g)
        torch.manual_seed(2024)
        mg, mn = du.MixUp_AUG().aug(gt, nz)
    finally:
        torch.Tensor.cuda = torch.Tensor._orig_cuda
    out.update(mix_gt_in=gt, mix_noisy_in=nz, mix_gt=mg, mix_noisy=mn)
    npz("data_aug", **out)


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--only" and sys.argv[2] == "data":
        gen_data()
        return
    if len(sys.argv) > 2 and sys.argv[1] == "--only" and sys.argv[2] == "blockwide":
        install_shims()
        import warnings
        warnings.filterwarnings("ignore")
        import My_model_1 as M1
        torch.set_num_threads(8)
        for name in WIDE_BLOCKS:
            gen_block_wide(M1, name)
        return
    if len(sys.argv) > 2 and sys.argv[1] == "--only" and sys.argv[2] == "uformer16":        # the round-5 fixtures alone
        install_shims()
        import warnings
        warnings.filterwarnings("ignore")
        import My_model_1 as M1
        import ProbSparse.attn as ATT
        import losses
        torch.set_num_threads(8)
        gen_probattn(M1, ATT, only=("h2_mask_bias_d16",))
        gen_block_wide(M1, "block_m1_c16_shift4")
        gen_block_wide(M1, "block_m1_c32h2_shift4")
        gen_block_wide(M1, "block_m1_c64_ffn_shift4")
        gen_full_e16(M1, losses)
        gen_full_ctor_default(M1, losses)
        gen_dead_branches(M1)
        return
    if len(sys.argv) > 2 and sys.argv[1] == "--only" and sys.argv[2] == "block128":
        install_shims()
        import warnings
        warnings.filterwarnings("ignore")
        import My_model_1 as M1
        torch.set_num_threads(8)
        gen_block_c128(M1)
        return
    install_shims()
    import warnings
    warnings.filterwarnings("ignore")
    import My_model_1 as M1
    import My_model as M0
    import ProbSparse.attn as ATT
    import losses
    torch.set_num_threads(8)
    print("generating goldens from", REF)
    gen_rng()
    gen_probattn(M1, ATT)
    gen_masks(M1)
    gen_blocks(M1, M0)
    for name in WIDE_BLOCKS:
        gen_block_wide(M1, name)
    gen_small_modules(M1)
    gen_losses(losses)
    gen_options()
    gen_misc()
    gen_full(M1, M0, losses)
    gen_full_e16(M1, losses)
    gen_full_ctor_default(M1, losses)
    gen_dead_branches(M1)
    gen_trajectory(M1, losses)
    gen_data()


if __name__ == "__main__":
    main()
