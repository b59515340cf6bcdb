"""The CPU oracle (oracle/uformer_oracle.py) against golden vectors captured from the reference itself
(tests/golden/gen_golden.py).  This is what pins the oracle; the -m gpu tests then compare HIP to it."""
import numpy as np
import pytest
import torch

from oracle import uformer_oracle as O

T = torch.from_numpy


def sd_from(g, prefix="sd/"):
    return {k[len(prefix):]: T(g[k]) for k in g.files if k.startswith(prefix)}


def test_rng_stream(golden):
    g = golden("rng_stream")
    torch.manual_seed(0)
    batched = torch.randint(64, (3, 64, 25))
    assert np.array_equal(batched.numpy(), g["idx"].astype(np.int64))


@pytest.mark.parametrize("case", ["h1_nomask_bias", "h2_mask_bias", "h16_nomask_nobias", "h2_mask_nobias",
                                  "h2_mask_bias_d64", "h2_mask_bias_d16"])
def test_prob_attention(golden, case):
    g = golden("probattn_" + case)
    q, k, v = (T(g[n]).transpose(1, 2).contiguous().requires_grad_() for n in "qkv")   # -> B_,H,N,d
    bias = T(g["bias"]).requires_grad_()
    use_bias = bool(g["use_bias"])
    mask = T(g["mask"]) if g["mask"].size else None
    idx = T(g["idx"].astype(np.int64))
    ctx, top, Mq, scores, a = O.prob_attention(q, k, v, idx, bias if use_bias else None, mask, return_aux=True)
    # selection: same SET of queries per (window, head)
    ref_top = np.sort(g["top"].astype(np.int64), -1)
    assert np.array_equal(np.sort(top.numpy(), -1), ref_top)
    ref_ctx = T(g["ctx"]).transpose(1, 2)
    assert torch.allclose(ctx, ref_ctx, atol=2e-6, rtol=1e-5)
    (ctx * T(g["gout"]).transpose(1, 2)).sum().backward()
    for name, t in (("dq", q), ("dk", k), ("dv", v)):
        assert torch.allclose(t.grad, T(g[name]).transpose(1, 2), atol=5e-6, rtol=1e-4), name
    if use_bias:
        assert torch.allclose(bias.grad, T(g["dbias"]), atol=5e-6, rtol=1e-4)


def test_shift_mask(golden):
    g = golden("shift_mask")
    m16 = O.shift_attn_mask(16, 16, 8, 4)
    assert np.array_equal((m16 != 0).numpy().astype(np.uint8), g["m16"])
    m128 = O.shift_attn_mask(128, 128, 8, 4)
    assert tuple(g["m128_shape"]) == tuple(m128.shape)
    assert np.array_equal(np.packbits((m128 != 0).numpy()), g["m128_packed"])
    assert set(np.unique(m128.numpy())) <= {0.0, -100.0} and set(g["vals"]) == {0.0, -100.0}


@pytest.mark.parametrize("name,variant,heads,shift", [
    ("block_m1_c32_shift0", "probsparse", 1, 0), ("block_m1_c32_shift4", "probsparse", 1, 4),
    ("block_m1_c64_shift4", "probsparse", 2, 4), ("block_m0_c32_shift0", "dense", 1, 0),
    ("block_m0_c32_shift4", "dense", 1, 4), ("block_m0_c64_shift4", "dense", 2, 4)])
def test_block(golden, name, variant, heads, shift):
    g = golden(name)
    P = {k: v.requires_grad_() if v.dtype.is_floating_point else v for k, v in sd_from(g).items()}
    x = T(g["x"]).requires_grad_()
    idx = T(g["idx"].astype(np.int64))
    y = O.lewin_block(x, P, "", heads, 8, shift, variant, idx)
    assert torch.allclose(y, T(g["y"]), atol=1e-5, rtol=1e-5)
    (y * T(g["gout"])).sum().backward()
    assert torch.allclose(x.grad, T(g["dx"]), atol=2e-5, rtol=1e-4)
    for k in g.files:
        if k.startswith("g/"):
            ref = g[k]
            p = P[k[2:]]
            if ref.size == 0:
                assert p.grad is None, k
            else:
                assert torch.allclose(p.grad, T(ref), atol=5e-5, rtol=2e-4), k


@pytest.mark.parametrize("tag", ["leff", "down", "up", "inproj", "outproj"])
def test_small_modules(golden, tag):
    g = golden("small_modules")
    P = {k[len(tag) + 4:]: T(g[k]).requires_grad_() for k in g.files if k.startswith(tag + "/sd/")}
    x = T(g[tag + "/x"]).requires_grad_()
    if tag == "leff":
        y = O.leff(x, P, "")
    elif tag == "down":
        y = O.downsample(x, {"d." + k: v for k, v in P.items()}, "d")
    elif tag == "up":
        y = O.upsample(x, {"u." + k: v for k, v in P.items()}, "u")
    elif tag == "inproj":
        y = O.input_proj(x, {"input_proj." + k: v for k, v in P.items()})
    else:
        y = O.output_proj(x, {"output_proj." + k: v for k, v in P.items()})
    assert torch.allclose(y, T(g[tag + "/y"]), atol=1e-5, rtol=1e-5)
    (y * T(g[tag + "/gout"])).sum().backward()
    assert torch.allclose(x.grad, T(g[tag + "/dx"]), atol=1e-5, rtol=1e-4)
    for k, p in P.items():
        assert torch.allclose(p.grad, T(g[f"{tag}/g/{k}"]), atol=1e-4, rtol=1e-4), k


def test_losses(golden):
    g = golden("losses")
    x = T(g["char_x"]).requires_grad_()
    l = O.charbonnier(x, T(g["char_y"]))
    assert abs(l.item() - float(g["char_loss"])) < 1e-7
    l.backward()
    assert torch.allclose(x.grad, T(g["char_dx"]), atol=1e-8, rtol=1e-5)
    W = O.seeded_vgg_weights()
    assert torch.equal(W[0][0], T(g["cr/vgg_w0"]))
    for tag, ab in (("cr", False), ("cr_ab", True)):
        a = T(g[tag + "/a"]).requires_grad_()
        loss, ap, an = O.contrast_loss(a, T(g[tag + "/p"]), T(g[tag + "/n"]), W, ablation=ab)
        assert abs(loss.item() - float(g[tag + "/loss"])) < 2e-6 * max(1, abs(float(g[tag + "/loss"])))
        assert abs(float(ap) - float(g[tag + "/all_ap"])) < 1e-5
        assert abs(float(an) - float(g[tag + "/all_an"])) < 1e-5
        loss.backward()
        assert torch.allclose(a.grad, T(g[tag + "/da"]), atol=1e-7, rtol=1e-3)
    feats = O.vgg19_features(T(g["cr/a"]), W)
    assert [list(f.shape) for f in feats] == g["cr/feat_shapes"].tolist()


# ----------------------------------------------------------------------------- plain-C oracle (oracle/ps_attn_oracle.c)
def _c_oracle():
    import ctypes
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = os.path.join(root, "oracle", "libps_attn_oracle.so")
    if not os.path.exists(so):
        subprocess.run(["make", "-s", "-C", os.path.join(root, "oracle")], check=True)
    return ctypes.CDLL(so)


@pytest.mark.parametrize("case", ["h1_nomask_bias", "h2_mask_bias", "h16_nomask_nobias", "h2_mask_nobias",
                                  "h2_mask_bias_d64", "h2_mask_bias_d16"])
def test_c_oracle_prob_attention(golden, case):
    """The double-precision C restatement reproduces the reference's outputs and gradients."""
    import ctypes
    lib = _c_oracle()
    g = golden("probattn_" + case)
    q, k, v = (np.ascontiguousarray(g[n].transpose(0, 2, 1, 3)) for n in "qkv")           # B_,H,N,d
    B_, H, N, d = q.shape
    use_bias = bool(g["use_bias"])
    bias = np.ascontiguousarray(g["bias"]) if use_bias else None
    mask = np.ascontiguousarray(g["mask"]) if g["mask"].size else None
    idx = np.ascontiguousarray(g["idx"].astype(np.int32))
    ctx = np.empty_like(q)
    top = np.empty((B_, H, 25), np.int32)
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p) if a is not None else None
    nW = mask.shape[0] if mask is not None else 1
    lib.ps_attn_oracle_fwd(P(q), P(k), P(v), P(idx), P(bias), P(mask), B_, H, nW, N, d, 25, P(ctx), P(top), None)
    assert np.array_equal(np.sort(top, -1), np.sort(g["top"].astype(np.int32), -1))
    assert np.allclose(ctx, g["ctx"].transpose(0, 2, 1, 3), atol=2e-6, rtol=1e-5)
    dctx = np.ascontiguousarray(g["gout"].transpose(0, 2, 1, 3))
    dq, dk, dv = np.empty_like(q), np.empty_like(q), np.empty_like(q)
    dbias = np.zeros((H, N, N), np.float32) if use_bias else None
    lib.ps_attn_oracle_bwd(P(q), P(k), P(v), P(bias), P(mask), P(top), P(dctx), B_, H, nW, N, d, 25, P(dq), P(dk),
                           P(dv), P(dbias))
    for name, arr in (("dq", dq), ("dk", dk), ("dv", dv)):
        assert np.allclose(arr, g[name].transpose(0, 2, 1, 3), atol=5e-6, rtol=1e-4), name
    if use_bias:
        assert np.allclose(dbias, g["dbias"], atol=5e-6, rtol=1e-4)


def test_train_trajectory(golden):
    """seed recipe -> the reference's 6-step loss trajectory (pins the order in which the
    sampling indices and the DropPath masks consume the global CPU generator)."""
    import random
    import My_model_1 as M1
    g = golden("train_trajectory")
    random.seed(1234); np.random.seed(1234); torch.manual_seed(1234)
    model = M1.Uformer(img_size=128, embed_dim=32, win_size=8, token_projection='linear', token_mlp='leff')
    P = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point) for k, v in model.state_dict().items()}
    opt = torch.optim.AdamW([P[n] for n, _ in model.named_parameters()], lr=2e-4, betas=(0.9, 0.999), eps=1e-8,
                            weight_decay=0.02)
    gen = torch.Generator().manual_seed(7)
    gt = torch.rand(4, 3, 128, 128, generator=gen)
    hazy = (0.6 * gt + 0.4 * torch.rand(4, 1, 1, 1, generator=gen)).clamp(0, 1)

    def step(a, b):
        opt.zero_grad()
        loss, _ = O.train_step_loss(P, hazy[a:b], gt[a:b], training=True)
        loss.backward()
        opt.step()
        return loss.item()

    step(0, 2)
    traj = [step(2 * s, 2 * s + 2) for _ in range(3) for s in range(2)]
    assert np.allclose(traj, g["losses"], atol=2e-5), (traj, g["losses"])


@pytest.mark.parametrize("gname,variant", [("full_m1_e32", "probsparse"), ("full_m0_e32", "dense")])
def test_full_model_oracle(golden, gname, variant):
    import random
    mod = __import__("My_model_1" if variant == "probsparse" else "My_model")
    g = golden(gname)
    random.seed(1234); np.random.seed(1234); torch.manual_seed(1234)
    model = mod.Uformer(img_size=128, embed_dim=32, win_size=8, token_projection='linear', token_mlp='leff')
    P = {k: v.detach() for k, v in model.state_dict().items()}
    hazy = T(g["hazy"]).float()
    torch.manual_seed(99)
    with torch.no_grad():
        y = O.uformer_forward(P, hazy, variant=variant)
    assert torch.allclose(y[0, :, 40:72, 40:72], T(g["y_eval_crop"]), atol=1e-5, rtol=1e-4)
    assert abs(float(y.double().sum()) - float(g["y_eval_sum"])) < 1e-4 * float(g["y_eval_abs"])


def test_full_model_e16_oracle(golden):
    """the embed_dim = 16 model of --arch Uformer16 (utils/model_utils.py:96-98; head_dim 16 everywhere): oracle vs the reference"""
    import random
    import My_model_1 as M1
    g = golden("full_m1_e16")
    random.seed(1234); np.random.seed(1234); torch.manual_seed(1234)
    model = M1.Uformer(img_size=128, embed_dim=16, win_size=8, token_projection='linear', token_mlp='leff')
    P = {k: v.detach() for k, v in model.state_dict().items()}
    assert list(P.keys()) == list(g["keys"])
    assert [str(tuple(v.shape)) for v in P.values()] == list(g["shapes"])
    hazy = T(g["hazy"]).float()
    torch.manual_seed(99)
    with torch.no_grad():
        y = O.uformer_forward(P, hazy, variant="probsparse")
    assert torch.allclose(y[0, :, 40:72, 40:72], T(g["y_eval_crop"]), atol=1e-5, rtol=1e-4)
    assert abs(float(y.double().sum()) - float(g["y_eval_sum"])) < 1e-4 * float(g["y_eval_abs"])


def test_full_model_ctor_default_oracle(golden):
    """M1.Uformer() with the constructor defaults (token_mlp = 'ffn'): oracle vs the reference"""
    import random
    import My_model_1 as M1
    g = golden("full_m1_ctor_default")
    random.seed(1234); np.random.seed(1234); torch.manual_seed(1234)
    model = M1.Uformer()
    P = {k: v.detach() for k, v in model.state_dict().items()}
    hazy = T(g["hazy"]).float()
    torch.manual_seed(99)
    with torch.no_grad():
        y = O.uformer_forward(P, hazy, variant="probsparse")
    assert torch.allclose(y[0, :, 40:72, 40:72], T(g["y_eval_crop"]), atol=1e-5, rtol=1e-4)
    assert abs(float(y.double().sum()) - float(g["y_eval_sum"])) < 1e-4 * float(g["y_eval_abs"])


WIDE_BLOCKS = {     # tests/golden/gen_golden.py::WIDE_BLOCKS: name -> (C, heads, map side, shift)
    "block_m1_c128_shift4": (128, 4, 16, 4),
    "block_m1_c256_shift4": (256, 8, 16, 4),
    "block_m1_c512_shift0": (512, 16, 8, 0),
    "block_m1_c16_shift4": (16, 1, 16, 4),          # head_dim 16: the embed_dim = 16 model's first stage ...
    "block_m1_c32h2_shift4": (32, 2, 16, 4),        # ... and its last decoder stage (C = 32 as two heads of 16)
    "block_m1_c64_ffn_shift4": (64, 2, 16, 4),      # token_mlp = 'ffn' (Mlp instead of LeFF, M1:778-779)
}


def _wide_block_inputs(g, M1, name):
    """rebuild weights / x / gout of tests/golden/gen_golden.py::gen_block_wide (the golden stores probes of them, not the tensors)"""
    import random
    C, heads, side, shift = WIDE_BLOCKS[name]
    random.seed(31); np.random.seed(31); torch.manual_seed(31)
    blk = M1.LeWinTransformerBlock(dim=C, input_resolution=(side, side), num_heads=heads, win_size=8, shift_size=shift,
                                   token_mlp='ffn' if "_ffn_" in name else 'leff', drop_path=0.)
    gen = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for p in blk.parameters():
            if p.ndim == 1:
                p.add_(0.1 * torch.randn(p.shape, generator=gen))
    x = torch.randn(1, side * side, C, generator=gen)
    gout = torch.randn(1, side * side, C, generator=gen)
    first = next(iter(blk.parameters()))
    assert torch.equal(first.detach().reshape(-1)[:16], T(g["w_probe"])) and torch.equal(x.reshape(-1)[:16], T(g["x_probe"]))
    assert torch.equal(gout.reshape(-1)[:16], T(g["gout_probe"]))
    return blk, x, gout


def _c128_block_inputs(g, M1):
    return _wide_block_inputs(g, M1, "block_m1_c128_shift4")


def check_c128_grads(g, named_grads, tol_rel):
    for k in g.files:
        if not k.startswith("gn/"):
            continue
        name = k[3:]
        gr = named_grads[name]
        if g[k].size == 0:
            assert gr is None, name
            continue
        assert gr is not None, name
        ref_n = float(g[k][0])
        assert abs(float(gr.double().norm()) - ref_n) <= tol_rel * ref_n + 1e-6, name
        samp = T(g["gs/" + name])
        assert torch.allclose(gr.reshape(-1)[::97].cpu(), samp, atol=tol_rel * float(samp.abs().max()) + 1e-6, rtol=10 * tol_rel), name


@pytest.mark.parametrize("name", sorted(WIDE_BLOCKS))
def test_block_wide_oracle_vs_reference(golden, name):
    """the oracle's block at C = 128 / 256 / 512 (4 / 8 / 16 heads; shifted windows; the bottleneck's single-window geometry) against the
    reference's own numbers"""
    import My_model_1 as M1
    C, heads, side, shift = WIDE_BLOCKS[name]
    g = golden(name)
    blk, x, gout = _wide_block_inputs(g, M1, name)
    P = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point) for k, v in blk.state_dict().items()}
    x = x.requires_grad_()
    y = O.lewin_block(x, P, "", heads, 8, shift, "probsparse", T(g["idx"].astype(np.int64)))
    assert torch.allclose(y, T(g["y"]), atol=2e-5, rtol=1e-5)
    (y * gout).sum().backward()
    assert torch.allclose(x.grad, T(g["dx"]), atol=5e-5, rtol=1e-4)
    check_c128_grads(g, {k: P[k].grad for k in P if P[k].dtype.is_floating_point}, 2e-4)
