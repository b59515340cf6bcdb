"""CPU-side (-m "not gpu") checks of the host logic: the C-ABI library loads and exports every symbol the
header declares, the drop-in surface matches the reference (state_dict keys, parameter order, init
stream, options), and the product refuses CPU tensors instead of silently falling back."""
import argparse
import os
import random
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from dehaze_hip import _lib
    lib = _lib.load()
    hdr = open(os.path.join(ROOT, "include", "dehaze_hip.h")).read()
    declared = set(re.findall(r"\b(dhz_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/dehaze_hip.h but not exported"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert lib.dhz_abi_version() == 1
    assert lib.dhz_ps_attn_bwd_parts(8192, 1) == 512 and lib.dhz_ps_attn_bwd_parts(4, 16) == 64
    assert lib.dhz_ps_attn_bwd_parts_d(8192, 1, 32) == 512 and lib.dhz_ps_attn_bwd_parts_d(8192, 2, 64) == 512


def test_argument_validation_without_gpu():
    """Bad arguments are rejected before any launch (safe on a GPU-less host)."""
    from dehaze_hip import _lib
    lib = _lib.load()
    rc = lib.dhz_ps_attn_fwd(None, None, None, 32, None, None, None, None, 32, None, 1, 1, 1, 32, None)
    assert rc == -22 and b"null pointer" in lib.dhz_last_error()
    rc = lib.dhz_ps_attn_fwd(8, 8, 8, 32, 8, None, None, 8, 32, 8, 1, 1, 1, 48, None)
    assert rc == -22 and b"head_dim" in lib.dhz_last_error()
    # the convolution / loss / feed entry points: shape contracts are checked before the launch too
    assert lib.dhz_winograd_conv3x3(8, 8, None, 0, None, None, 8, 2, 24, 32, 64, 64, None) == -22     # H % 16 != 0
    assert b"unsupported shape" in lib.dhz_last_error()
    assert lib.dhz_winograd_conv3x3(8, 8, 8, 1, 8, None, 8, 2, 32, 32, 64, 64, None) == -22           # bias/relu with out_mask
    assert b"exclusive" in lib.dhz_last_error()
    assert lib.dhz_winograd_prepack(8, 8, 48, 64, 0, None) == -22                                     # Kout % 32 != 0
    assert lib.dhz_maxpool2x2_blocked_fwd(8, 8, 4, 15, 16, None) == -22
    assert lib.dhz_thin_conv3x3_fwd(8, 8, None, 8, 1, 16, 16, 32, None) == -22 and b"C=32 unsupported" in lib.dhz_last_error()
    assert lib.dhz_thin_conv3x3_wgrad(8, 8, None, None, 1, 16, 16, 64, None) == -22
    assert lib.dhz_l1_pair_fwd(8, 8, None, 8, 10, None) == -22 and b"multiple of 4" in lib.dhz_last_error()
    assert lib.dhz_crop_augment_pair(8, 8, 8, 8, 8, 4, 16, 16, 32, None) == -22 and b"bad sizes" in lib.dhz_last_error()
    assert lib.dhz_linear_wgrad(8, 32, 8, 32, 100, 32, 32, 8, None, None) == -22                      # T % 32 != 0


@pytest.mark.parametrize("modname,gname", [("My_model_1", "full_m1_e32"), ("My_model", "full_m0_e32")])
def test_state_dict_and_init_stream_match_reference(golden, modname, gname):
    mod = __import__(modname)
    g = golden(gname)
    random.seed(1234); np.random.seed(1234); torch.manual_seed(1234)
    m = mod.Uformer(img_size=128, embed_dim=32, win_size=8, token_projection='linear', token_mlp='leff')
    sd = m.state_dict()
    assert list(sd.keys()) == list(g["keys"])
    assert [n for n, _ in m.named_parameters()] == list(g["pnames"])
    assert [str(tuple(v.shape)) for v in sd.values()] == list(g["shapes"])
    stats = np.array([[float(v.double().sum()), float(v.double().abs().sum())] for v in sd.values()])
    assert np.array_equal(stats, g["sd_stats"])          # same init RNG stream, bit for bit
    assert sum(p.numel() for p in m.parameters()) == 26222685 if modname == "My_model_1" else True
    if modname == "My_model_1":
        live = m.live_parameters()
        assert sum(p.numel() for _, p in live) == 20628317 and len(list(m.parameters())) - len(live) == 108


def test_constructor_default_model_is_the_ffn_variant(golden):
    """M1.Uformer() with the constructor's own defaults (M1:961-967): token_mlp = 'ffn' - Mlp (fc1 / fc2) blocks instead of LeFF.  Keys,
    parameter order, shapes and the init RNG stream equal the reference's."""
    import My_model_1 as M1
    g = golden("full_m1_ctor_default")
    random.seed(1234); np.random.seed(1234); torch.manual_seed(1234)
    m = M1.Uformer()
    sd = m.state_dict()
    assert list(sd.keys()) == list(g["keys"]) and any(k.endswith("mlp.fc1.weight") for k in sd) and not any("dwconv" in k for k in sd)
    assert [n for n, _ in m.named_parameters()] == list(g["pnames"])
    assert [str(tuple(v.shape)) for v in sd.values()] == list(g["shapes"])
    stats = np.array([[float(v.double().sum()), float(v.double().abs().sum())] for v in sd.values()])
    assert np.array_equal(stats, g["sd_stats"])
    with pytest.raises(Exception, match="FFN error"):
        M1.LeWinTransformerBlock(dim=32, input_resolution=(16, 16), num_heads=1, token_mlp="mlp")


@pytest.mark.parametrize("tag,kw", [("conv_se", dict(token_projection='conv', se_layer=True)), ("concat", dict(token_projection='linear_concat'))])
def test_off_default_projections_are_dead_parameters_with_the_reference_layout(golden, tag, kw):
    """token_projection = 'conv' / 'linear_concat', se_layer = True (M1:384-394): in the ProbSparse model these modules are registered,
    initialised and checkpointed but never run (M1:400-415) - keys, shapes and init stream equal the reference's, they stay out of the
    live parameter set; the dense twin, which does run them, refuses."""
    import My_model as M0
    import My_model_1 as M1
    g = golden("dead_branches")
    random.seed(1234); np.random.seed(1234); torch.manual_seed(1234)
    m = M1.Uformer(img_size=128, embed_dim=32, win_size=8, token_mlp='leff', **kw)
    sd = m.state_dict()
    assert list(sd.keys()) == list(g[tag + "/keys"])
    assert [str(tuple(v.shape)) for v in sd.values()] == list(g[tag + "/shapes"])
    stats = np.array([[float(v.double().sum()), float(v.double().abs().sum())] for v in sd.values()])
    assert np.array_equal(stats, g[tag + "/sd_stats"])
    live = {n for n, _ in m.live_parameters()}
    assert not any(".attn.qkv." in n or ".attn.proj." in n or ".attn.se_layer." in n for n in live)
    assert sum(p.numel() for n, p in m.named_parameters() if n in live) == 20628317           # the same live set as the default model
    with pytest.raises(NotImplementedError):
        M0.Uformer(img_size=128, embed_dim=32, win_size=8, token_mlp='leff', **kw)


def test_options_match_reference(golden):
    import options
    g = golden("options")
    p = options.Options().init(argparse.ArgumentParser())
    mine = {a.dest: (str(a.default), str(a.type.__name__ if a.type else None), a.__class__.__name__)
            for a in p._actions if a.dest != "help"}
    assert list(mine) == list(g["dests"])
    for d, dflt, ty, kind in zip(g["dests"], g["defaults"], g["types"], g["kinds"]):
        if d in ("pretrain_weights",):                 # machine-specific absolute path in the reference
            continue
        assert mine[d] == (dflt, ty, kind), d
    assert options.is_relative_position_bias is True and int(g["is_relative_position_bias"]) == 1


def test_misc_host_helpers(golden):
    import utils
    from warmup_scheduler import GradualWarmupScheduler
    g = golden("misc")
    torch.manual_seed(5)
    mix = utils.MixUp_AUG()
    a, b = mix.aug(torch.from_numpy(g["mix_gt"]), torch.from_numpy(g["mix_nz"]))
    assert torch.allclose(a, torch.from_numpy(g["mix_out_gt"])) and torch.allclose(b, torch.from_numpy(g["mix_out_nz"]))
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([p], lr=2e-4)
    cos = torch.optim.lr_scheduler.CosineAnnealingLR(opt, 20 - 3, eta_min=1e-6)
    sch = GradualWarmupScheduler(opt, multiplier=1, total_epoch=3, after_scheduler=cos)
    lrs = []
    for ep in range(20):
        lrs.append(opt.param_groups[0]['lr'])
        opt.step()
        sch.step()
    assert np.allclose(lrs, g["lrs"], rtol=1e-9, atol=0)


def test_product_refuses_cpu_tensors():
    import My_model_1 as M1
    from losses import CharbonnierLoss
    blk = M1.LeWinTransformerBlock(dim=32, input_resolution=(16, 16), num_heads=1, win_size=8, shift_size=0, token_mlp='leff')
    with pytest.raises(RuntimeError, match="no CPU"):
        blk(torch.zeros(1, 256, 32))
    with pytest.raises(RuntimeError, match="no CPU"):
        CharbonnierLoss()(torch.zeros(4), torch.zeros(4))


def test_get_arch_and_checkpoint_roundtrip(tmp_path):
    import utils
    ns = argparse.Namespace(arch="Uformer", train_ps=128, embed_dim=32, win_size=8, token_projection="linear", token_mlp="leff")
    torch.manual_seed(0)
    m = utils.get_arch(ns)
    assert type(m).__name__ == "Uformer" and m.variant == "probsparse"
    with pytest.raises(Exception, match="Arch error"):
        utils.get_arch(argparse.Namespace(arch="nope", train_ps=128, embed_dim=32, win_size=8, token_projection="linear", token_mlp="leff"))
    # reference checkpoints carry DataParallel's 'module.' prefix (TR:294-297)
    path = str(tmp_path / "ck.pth")
    torch.save({"epoch": 7, "state_dict": {"module." + k: v for k, v in m.state_dict().items()}, "optimizer": {}}, path)
    torch.manual_seed(1)
    m2 = utils.get_arch(ns)
    utils.load_checkpoint(m2, path)
    assert utils.load_start_epoch(path) == 7
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))


def test_flat_buffer_layout_packs_qkv():
    """FlatAdamW's flat parameter buffer: values preserved, slices sorted and gap-free, and in every attention layer the three
    projection weights (and biases) sit back to back, so that ops.cat_rows returns a VIEW of the buffer instead of a copy."""
    import My_model_1 as M1
    from dehaze_hip import ops
    from dehaze_hip.train import FlatAdamW
    torch.manual_seed(0)
    model = M1.Uformer(img_size=128, embed_dim=32, win_size=8, token_projection='linear', token_mlp='leff')
    before = {n: p.detach().clone() for n, p in model.named_parameters()}
    opt = FlatAdamW(model)
    opt._ensure_flat()
    for n, p in model.named_parameters():
        assert torch.equal(p.detach(), before[n]), n
    sl = opt.param_slices()
    live = [p for _, p in model.live_parameters()]
    assert len(sl) == len(live) and {id(p) for p, _, _ in sl} == {id(p) for p in live}
    off = 0
    for p, o, k in sl:
        assert o == (off + 7) // 8 * 8 and k == p.numel()          # 32-byte aligned, back to back otherwise
        off = o + k
    layers = [m for m in model.modules() if hasattr(m, "query_projection")]
    assert len(layers) == 18
    for m in layers:
        ws = [m.query_projection.weight.detach(), m.key_projection.weight.detach(), m.value_projection.weight.detach()]
        bs = [m.query_projection.bias.detach(), m.key_projection.bias.detach(), m.value_projection.bias.detach()]
        W, b = ops.cat_rows(ws), ops.cat_rows(bs)
        assert W.data_ptr() == ws[0].data_ptr() and b.data_ptr() == bs[0].data_ptr()          # views, not copies
        assert torch.equal(W, torch.cat(ws, 0)) and torch.equal(b, torch.cat(bs, 0))
    # not adjacent (or tracked by autograd): falls back to a copy with the same values
    a, c = torch.randn(4, 3), torch.randn(2, 3)
    assert torch.equal(ops.cat_rows([a, c]), torch.cat([a, c], 0))
    # neighbouring addresses in DIFFERENT allocations must not be taken for one buffer
    import ctypes
    raw = torch.zeros(64)
    u, v = raw[:12].view(4, 3).clone(), None
    blob = torch.empty(24)
    u2, v2 = blob[:12].view(4, 3), blob[12:].view(4, 3)
    assert ops.cat_rows([u2, v2]).data_ptr() == u2.data_ptr()
    assert ops.cat_rows([u, u2]).data_ptr() not in (u.data_ptr(), u2.data_ptr())
    assert ops.cat_rows([m.query_projection.weight, m.key_projection.weight]).data_ptr() != m.query_projection.weight.data_ptr()


def test_rng_state_is_plain_tensors_and_per_rank(tmp_path):
    """The generator states of a checkpoint decode without pickle (a crafted 'host' blob executed code on --resume),
    survive torch.load(weights_only=True), are validated, and are restored PER RANK (a multi-rank resume must not hand rank 0's
    streams to every rank)."""
    import utils
    random.seed(5); np.random.seed(6); torch.manual_seed(7)
    random.gauss(0, 1); np.random.standard_normal()            # populate both cached-gaussian fields
    st0 = utils.rng_state_dict()
    assert all(isinstance(v, torch.Tensor) for v in st0.values()) and "host" not in st0
    want0 = (random.random(), np.random.rand(), np.random.standard_normal(), random.gauss(0, 1), torch.rand(1).item())
    random.seed(50); np.random.seed(60); torch.manual_seed(70)
    st1 = utils.rng_state_dict()
    want1 = (random.random(), np.random.rand(), np.random.standard_normal(), random.gauss(0, 1), torch.rand(1).item())
    path = str(tmp_path / "ck.pth")
    torch.save({"epoch": 1, "rng_state": [st0, st1]}, path)
    torch.load(path, map_location="cpu", weights_only=True)     # nothing but tensors / lists / dicts in the file
    for rank, want in ((0, want0), (1, want1)):
        random.seed(999); np.random.seed(999); torch.manual_seed(999)
        assert utils.load_rng_state(path, rank) is True
        got = (random.random(), np.random.rand(), np.random.standard_normal(), random.gauss(0, 1), torch.rand(1).item())
        assert got == want, rank
    assert utils.load_rng_state(path, 2) is False               # resumed on more ranks than the file was written with
    torch.save({"epoch": 1, "rng_state": st0}, path)            # single-process layout: a bare dict = rank 0
    assert utils.load_rng_state(path) is True and utils.load_rng_state(path, 1) is False
    bad = dict(st0, python_mt=st0["python_mt"][:100])
    with pytest.raises(ValueError):
        utils.set_rng_state(bad)
    bad = dict(st0, numpy_mt=st0["numpy_mt"].clone())
    bad["numpy_mt"][3] = -1
    with pytest.raises(ValueError):
        utils.set_rng_state(bad)
    with pytest.raises((KeyError, ValueError)):
        utils.set_rng_state({"host": torch.zeros(8, dtype=torch.uint8), "torch": torch.get_rng_state()})   # the round-2 layout


def test_pmc_traffic_is_tied_to_the_loaded_library(tmp_path):
    """bench.py reports roofline.traffic only when profiles/pmc_traffic.json carries the build id of the library that is loaded
    (the committed table went stale silently when a kernel changed)."""
    import importlib.util
    import json
    from dehaze_hip import _lib
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    bid = _lib.load().dhz_build_id().decode()
    assert re.fullmatch(r"[0-9a-f]{16}(-.+)?", bid)
    table = {"fused_window_attn_fwd_kernel<32, true>": {"launches": 2, "hbm_bytes_per_launch": 1.0e8},
             "_stamp": {"build_id": bid}}
    path = str(tmp_path / "pmc.json")
    json.dump(table, open(path, "w"))
    pmc, src = bench.load_pmc_traffic(path, bid)
    assert list(pmc) == ["fused_window_attn_fwd_kernel<32, true>"] and bid in src
    flipped = bid[:-1] + ("0" if bid[-1] != "0" else "1")
    table["_stamp"]["build_id"] = flipped
    json.dump(table, open(path, "w"))
    pmc, src = bench.load_pmc_traffic(path, bid)
    assert pmc == {} and "not reported" in src                   # -> every roofline object gets "traffic": null
    pmc, src = bench.load_pmc_traffic(str(tmp_path / "missing.json"), bid)
    assert pmc == {} and "not reported" in src


def test_pointer_arguments_keep_their_tensors_alive():
    """ops._p hands a raw device pointer to a launch; a temporary (`_p(g.contiguous())`) must not be freed the moment _p returns:
    the most recent pointer-argument tensors stay referenced (longer than the longest argument list)."""
    import gc
    import weakref
    import torch
    from dehaze_hip import ops
    t = torch.arange(12.0).view(3, 4).t()                 # non-contiguous: .contiguous() makes a temporary
    tmp = t.contiguous()
    ref = weakref.ref(tmp)
    ptr = ops._p(tmp)
    del tmp
    gc.collect()
    assert ref() is not None and ref().data_ptr() == ptr  # still alive although no caller holds it
    keep = [torch.zeros(1) for _ in range(ops._RECENT.maxlen)]
    for k in keep:
        ops._p(k)
    gc.collect()
    assert ref() is None                                  # and released once enough launches have gone by
    assert ops._p(None) is None and ops._RECENT.maxlen >= 32


def test_split_switch_values():
    from dehaze_hip import ops
    assert [ops._split_terms(v) for v in ("0", "1", "3", "6", 6)] == [0, 3, 3, 6, 6]
    import pytest
    for bad in ("2", "x", "-1"):
        with pytest.raises(ValueError, match="DHZ_SPLIT_BF16"):
            ops._split_terms(bad)


def test_round6_entry_points_validate_without_gpu():
    """the residual-epilogue GEMMs, the layout forms of the LayerNorm backward and the batched staging entry points refuse bad arguments before any
    launch (DHZ_EINVAL with a message) - no device needed"""
    import ctypes
    from dehaze_hip import _lib
    lib = _lib.load()
    p = 16      # any non-null, 16-byte-aligned "pointer": the checks run before the first dereference
    #                                    x  ldx hi mid lo bias res scale out ldo  T    N   K   HW  H  W shift win stream
    assert lib.dhz_linear_fwd_split6_res(p, 64, p, p, p, None, p, None, p, 64, 192, 64, 64, 60, 6, 10, 0, 1, None) == -22      # HW % 64
    assert b"multiple of 64" in lib.dhz_last_error()
    assert lib.dhz_linear_fwd_split6_res(p, 64, p, p, p, None, p, None, p, 64, 192, 64, 64, 64, 8, 8, 8, 1, None) == -22       # shift range
    assert lib.dhz_linear_fwd_split6_res(p, 64, p, p, p, None, 8, None, p, 64, 192, 64, 64, 64, 8, 8, 0, 1, None) == -22       # shortcut alignment
    assert lib.dhz_linear_fwd_split_res(p, 64, p, None, p, None, p, 64, 192, 64, 64, 128, 8, 16, 0, 1, 6, None) == -22         # T % HW
    assert lib.dhz_linear_fwd_bf16_res(p, 64, p, None, p, None, p, 64, 192, 64, 64, 64, 4, 16, 0, 1, None) == -22              # H % 8
    assert lib.dhz_ln_partition_bwd_lay(p, p, p, p, p, p, p, p, 1, 8, 8, 64, 0, 0, 1, 0, 0, 0, None) == -22                    # windowed dres without partition
    assert lib.dhz_ln_partition_bwd_lay(p, p, p, p, p, p, p, p, 1, 8, 8, 64, 0, 0, 0, 1, 0, 0, None) == -22                    # dx == dres with a windowed dx
    assert lib.dhz_ln_partition_bwd_lay2(p, p, p, p, None, p, p, p, 1, 12, 8, 64, 0, 0, 0, 0, 4, 32, None, 0, None) == -22     # second output on a 12-row map
    one = (ctypes.c_void_p * 1)(p)
    arr = ctypes.cast(one, ctypes.c_void_p)
    heads = ctypes.cast((ctypes.c_int * 1)(0), ctypes.c_void_p)
    assert lib.dhz_bias_gather_multi(arr, arr, heads, 1, None) == -22                                                          # H = 0
    assert lib.dhz_bias_gather_multi(arr, arr, heads, 33, None) == -22                                                         # more than 32 entries
    c48 = ctypes.cast((ctypes.c_int * 1)(48), ctypes.c_void_p)
    assert lib.dhz_fused_attn_prepack_multi(arr, arr, arr, arr, arr, arr, c48, 1, None) == -22 and b"C=48" in lib.dhz_last_error()
    assert lib.dhz_fused_attn_prepack6(p, p, p, p, p, 48, None) == -22 and lib.dhz_leff_prepack6(p, p, p, 128, None) == -22
    assert lib.dhz_fused_window_attn_fwd6(p, p, p, p, p, p, p, p, None, None, None, p, None, None, None, None, None, 1, 8, 8, 48, 0, None) == -22


def test_zero_scratch_allocator():
    """ops.zeros_f32: slices of the optimizer's pre-zeroed region, never handed out twice between two zero_grad() calls, fresh allocations when
    the region is exhausted or its owner is gone"""
    import torch
    from dehaze_hip import ops
    from dehaze_hip.train import FlatAdamW
    lin = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3), torch.nn.Linear(8, 8))
    opt = FlatAdamW(lin)
    opt.zero_grad()
    scr = ops.ZERO_SCRATCH[0]
    a = ops.zeros_f32((5, 3), torch.device("cpu"))
    b = ops.zeros_f32((7,), torch.device("cpu"))
    assert a.untyped_storage().data_ptr() == scr.untyped_storage().data_ptr() == b.untyped_storage().data_ptr()
    assert a.data_ptr() != b.data_ptr() and b.data_ptr() >= a.data_ptr() + 15 * 4 and (b.data_ptr() - scr.data_ptr()) % 32 == 0
    a.fill_(3.0); b.fill_(4.0)
    big = ops.zeros_f32((scr.numel() + 1,), torch.device("cpu"))              # does not fit: a fresh tensor
    assert big.untyped_storage().data_ptr() != scr.untyped_storage().data_ptr() and float(big.abs().sum()) == 0.0
    assert all(float(p.grad.abs().sum()) == 0.0 for p in lin.parameters())    # the scratch lies BEHIND the gradients
    opt.zero_grad()                                                           # re-zeroes the region and rewinds it
    c = ops.zeros_f32((5, 3), torch.device("cpu"))
    assert c.data_ptr() == a.data_ptr() and float(c.abs().sum()) == 0.0
    del opt
    import gc
    gc.collect()
    d = ops.zeros_f32((5, 3), torch.device("cpu"))                            # owner gone: never a slice of a dead optimizer's buffer
    assert d.untyped_storage().data_ptr() != scr.untyped_storage().data_ptr()
