"""-m gpu: fp32 GEMMs on the bf16 matrix pipe by operand splitting (csrc/linear_split.hip; dehaze_hip.ops.SPLIT_BF16).  The
six-term form (three bf16 pieces per operand, dropped terms <= 2^-24 relative) is the product's default arithmetic: against
float64 it must stay within a small factor of the fp32-pipe kernel's own error.  The three-term form (two pieces, ~16 mantissa
bits per product: |err| <= 2^-15 * sum_k |a_k||b_k|) is kept as an experiment and is never a product setting."""
import pytest
import torch

pytestmark = pytest.mark.gpu


# elementwise bound against float64, relative to sum_k |a_k||b_k|: three terms drop 2^-16-class products; six terms drop only
# 2^-24-class ones and are left with fp32 accumulation error - the class of the fp32 kernel itself
BOUND = {3: 2.0 ** -15, 6: 2.0 ** -21}


@pytest.mark.parametrize("terms", [3, 6])
@pytest.mark.parametrize("T,K,N", [(4096, 128, 128), (1000, 256, 64), (777, 128, 512), (32768, 256, 64), (64, 1024, 256),
                                   (5000, 192, 64)])
def test_split_gemm_forward_and_dgrad_vs_fp64(T, K, N, terms):
    from dehaze_hip import _lib
    dev = torch.device("cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(T + K + N)
    x = torch.randn(T, K, generator=g).to(dev)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    y = torch.empty(T, N, device=dev)
    _lib.call("dhz_linear_fwd_split", x.data_ptr(), K, w.data_ptr(), b.data_ptr(), y.data_ptr(), N, T, N, K, terms, s)
    ref = x.double() @ w.double().t() + b.double()
    mag = x.double().abs() @ w.double().abs().t() + b.double().abs()
    err = (y.double() - ref).abs()
    assert (err <= BOUND[terms] * mag).all(), (err / mag).max().item()
    y32 = torch.empty(T, N, device=dev)
    _lib.call("dhz_linear_fwd", x.data_ptr(), K, w.data_ptr(), b.data_ptr(), y32.data_ptr(), N, T, N, K, s)
    e32 = (y32.double() - ref).abs().max().item()
    # three terms: bounded loss of precision; six terms: within a small factor of the fp32 kernel's own error
    assert err.max().item() < (2000 if terms == 3 else 4) * max(e32, 1e-7), (err.max().item(), e32)
    # backward-data: dx[T,K] = dy[T,N] . w[N,K] needs a contraction (N) of a multiple of 64
    dy = torch.randn(T, N, generator=g).to(dev)
    dx = torch.empty(T, K, device=dev)
    _lib.call("dhz_linear_dgrad_split", dy.data_ptr(), N, w.data_ptr(), dx.data_ptr(), K, T, N, K, terms, s)
    ref = dy.double() @ w.double()
    mag = dy.double().abs() @ w.double().abs()
    err = (dx.double() - ref).abs()
    assert (err <= BOUND[terms] * mag).all(), (err / mag).max().item()


def test_six_term_split_is_the_default_and_the_switch_routes():
    """the product's default arithmetic is the six-term split (ops.SPLIT_BF16 == 6 unless DHZ_SPLIT_BF16 says otherwise); 0 sends
    the same call to the fp32 matrix pipe"""
    import os
    from dehaze_hip import ops, _lib
    assert ops.SPLIT_BF16 == ops._split_terms(os.environ.get("DHZ_SPLIT_BF16", "6"))
    assert ops._split_terms("6") == 6 and "DHZ_SPLIT_BF16" in os.environ or ops.SPLIT_BF16 == 6
    dev = torch.device("cuda:0")
    x = torch.randn(512, 128, device=dev)
    w = torch.randn(64, 128, device=dev) / 11.0
    calls = []
    orig = _lib.call
    _lib.call = lambda name, *a: (calls.append(name), orig(name, *a))[1]
    old = ops.SPLIT_BF16
    try:
        ops.SPLIT_BF16 = 0
        y0 = ops.gemm_fwd(x, w)
        ops.SPLIT_BF16 = 6
        y1 = ops.gemm_fwd(x, w)
    finally:
        ops.SPLIT_BF16 = old
        _lib.call = orig
    assert calls[0] == "dhz_linear_fwd" and calls[1] != "dhz_linear_fwd" and "split" in calls[1], calls
    ref = x.double() @ w.double().t()
    e0, e1 = (y0.double() - ref).abs().max().item(), (y1.double() - ref).abs().max().item()
    assert e1 < 4 * max(e0, 1e-7), (e0, e1)                    # the error class of the fp32 pipe


@pytest.mark.parametrize("terms", [3, 6])
@pytest.mark.parametrize("T,nmat,nper,K,scaled", [(4096, 1, 128, 64, False), (2048, 3, 64, 64, False), (8192, 1, 64, 256, True),
                                                  (1024, 1, 512, 128, False), (4096, 3, 128, 128, False),
                                                  # step-sized problems: many token splits per tile, the packed three-matrix form, the
                                                  # row-scaled form, a token count that does not divide evenly over the splits
                                                  (16384, 1, 256, 256, False), (65536, 1, 256, 64, True), (65536, 1, 64, 256, False),
                                                  (32768, 3, 128, 128, False), (33280, 1, 128, 256, True),
                                                  (4736, 1, 128, 128, False)])     # 18 slabs of 8 or 9 stages: paired groups, unequal trip counts
def test_split_wgrad_vs_fp64(T, nmat, nper, K, scaled, terms):
    import ctypes
    from dehaze_hip import _lib
    dev = torch.device("cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(T + nper + K)
    N = nmat * nper
    dy = torch.randn(T, N, generator=g).to(dev)
    x = torch.randn(T, K, generator=g).to(dev)
    rps = 512 if T % 512 == 0 else 32
    rs = (0.5 + torch.rand(T // rps, generator=g)).to(dev) if scaled else None
    dws = [torch.zeros(nper, K, device=dev) for _ in range(nmat)]
    dbs = [torch.zeros(nper, device=dev) for _ in range(nmat)]
    pw = (ctypes.c_void_p * nmat)(*[t.data_ptr() for t in dws])
    pb = (ctypes.c_void_p * nmat)(*[t.data_ptr() for t in dbs])
    _lib.call("dhz_linear_wgrad_split", dy.data_ptr(), N, x.data_ptr(), K, T, nmat, nper, K, ctypes.cast(pw, ctypes.c_void_p),
              ctypes.cast(pb, ctypes.c_void_p), rs.data_ptr() if scaled else None, rps if scaled else 0, terms, s)
    d64 = dy.double() * (rs.double().repeat_interleave(rps)[:, None] if scaled else 1.0)
    ref = d64.t() @ x.double()
    mag = d64.abs().t() @ x.double().abs()
    got = torch.cat(dws, 0).double()
    err = (got - ref).abs()
    assert (err <= BOUND[terms] * mag + 1e-6).all(), (err / mag).max().item()
    assert torch.allclose(torch.cat(dbs, 0).double(), d64.sum(0), rtol=1e-5, atol=1e-3 * T ** 0.5)


@pytest.mark.parametrize("terms", [3, 6])
def test_split_model_step_close_to_fp32_step(terms):
    """the whole model forward / backward with the switch on against the fp32 pipe on the same weights, batch and sampled keys:
    output PSNR, loss, parameter-gradient direction.  (The golden / oracle model tests of tests/test_gpu_model.py also pass under
    DHZ_SPLIT_BF16=1 at their fp32 tolerances - DESIGN.md section 4c; the kernel-level fp32 tolerances of tests/test_gpu_linear.py
    do not, by design.)"""
    import math
    import My_model_1 as M1
    from dehaze_hip import ops
    from dehaze_hip.train import synthetic_batch
    from losses import CharbonnierLoss
    dev = torch.device("cuda:0")
    torch.manual_seed(11)
    model = M1.Uformer(img_size=128, embed_dim=32, win_size=8, token_projection='linear', token_mlp='leff',
                       drop_path_rate=0.).to(dev).train()
    gt, hazy = synthetic_batch(2, 128, seed=5, device=dev)
    char = CharbonnierLoss()
    res = {}
    old = ops.SPLIT_BF16
    try:
        for flag in (False, True):
            ops.SPLIT_BF16 = terms if flag else 0
            model.zero_grad(set_to_none=True)
            torch.manual_seed(99)                                   # the same sampled keys
            out = model(hazy)
            loss = char(out, gt)
            loss.backward()
            res[flag] = (out.detach().clone(), loss.item(), torch.cat([p.grad.flatten() for p in model.parameters() if p.grad is not None]))
    finally:
        ops.SPLIT_BF16 = old
    (o0, l0, g0), (o1, l1, g1) = res[False], res[True]
    mse = torch.mean((o0.double() - o1.double()) ** 2).item()
    psnr = 10 * math.log10(1.0 / max(mse, 1e-30))
    cos = torch.nn.functional.cosine_similarity(g0.double(), g1.double(), dim=0).item()
    rel = ((g0 - g1).norm() / g0.norm()).item()
    print("split(%d) vs fp32: PSNR %.1f dB, loss %.7f vs %.7f, grad cos %.8f rel %.2e" % (terms, psnr, l0, l1, cos, rel))
    # measured (three terms): PSNR 132 dB, equal loss to 7 digits, gradient relative difference 1.4e-6
    assert psnr > 100 and abs(l0 - l1) < 1e-5 * abs(l0) and cos > 0.999999 and rel < 1e-4, (psnr, l0, l1, cos, rel)


# ---------------------------------------------------------------------------------------------------------------------------------
# csrc/split6_gemm.hip: the six-term GEMMs with the weight operand pre-split into three bf16 planes


def _planes(w):
    from dehaze_hip import _lib
    s = torch.cuda.current_stream().cuda_stream
    pl = torch.empty((3, w.numel()), dtype=torch.bfloat16, device=w.device)
    _lib.call("dhz_split3_planes", w.data_ptr(), w.numel(), pl[0].data_ptr(), pl[1].data_ptr(), pl[2].data_ptr(), s)
    return pl


@pytest.mark.parametrize("n", [8, 4096, 1 << 20])
def test_split3_planes_are_exact(n):
    """hi + mid + lo == x bit for bit (three bf16 pieces by truncation hold all 24 mantissa bits) for zeros, negative values, values
    whose lower pieces vanish and every magnitude whose lowest piece is still a normal number (|x| >= 2^-100; below that the
    remainders are subnormal and the split is only good to 2^-126 absolute - no weight or activation of this model comes close)"""
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(n)
    x = torch.randn(n, generator=g) * torch.exp(8 * torch.randn(n, generator=g))
    x[:4] = torch.tensor([0.0, -0.0, 1.0, -1.5])
    x[4:8] = torch.tensor([1e-30, -3e-29, 65536.0, 3.0e38])
    x = torch.where((x.abs() < 1e-30) & (x != 0), torch.full_like(x, 1e-30), x).to(dev)
    pl = _planes(x)
    rec = (pl[0].double() + pl[1].double() + pl[2].double())
    assert torch.equal(rec, x.double())
    assert torch.equal(pl[0].float(), (x.view(torch.int32) & -65536).view(torch.float32))          # hi = the top 16 bits


# tile selection of csrc/split6_gemm.hip: features % 128 -> 128-wide tiles (256 x 128 when the problem has >= one tile per CU),
# % 64 -> 128 x 64, % 32 -> 128 x 32 (forward only); ragged T; strided activations (a packed QKV buffer); no bias
@pytest.mark.parametrize("T,K,N,ldx_pad", [(128, 32, 32, 0), (1000, 64, 96, 0), (777, 128, 512, 0), (4096, 128, 128, 64),
                                           (32768, 256, 256, 0), (65536, 64, 128, 0), (5000, 192, 64, 32), (64, 1024, 256, 0),
                                           (33000, 96, 384, 0), (66000, 128, 256, 64)])
def test_split6_planes_gemm_forward_and_dgrad_vs_fp64(T, K, N, ldx_pad):
    from dehaze_hip import _lib
    dev = torch.device("cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(T + K + N)
    xb = torch.randn(T, K + ldx_pad, generator=g).to(dev)
    x = xb[:, :K]
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    pl = _planes(w)
    for bias in (b, None):
        y = torch.full((T, N), float("nan"), device=dev)
        _lib.call("dhz_linear_fwd_split6", x.data_ptr(), K + ldx_pad, pl[0].data_ptr(), pl[1].data_ptr(), pl[2].data_ptr(),
                  bias.data_ptr() if bias is not None else None, y.data_ptr(), N, T, N, K, s)
        ref = x.double() @ w.double().t() + (bias.double() if bias is not None else 0.0)
        mag = x.double().abs() @ w.double().abs().t() + (bias.double().abs() if bias is not None else 0.0)
        err = (y.double() - ref).abs()
        assert (err <= BOUND[6] * mag).all(), (err / mag).max().item()
    y32 = torch.empty(T, N, device=dev)
    _lib.call("dhz_linear_fwd", x.data_ptr(), K + ldx_pad, w.data_ptr(), None, y32.data_ptr(), N, T, N, K, s)
    e32 = (y32.double() - ref).abs().max().item()
    assert err.max().item() < 4 * max(e32, 1e-7), (err.max().item(), e32)            # the error class of the fp32 pipe
    if K % 64 == 0:                                  # backward-data: its output features are w's K columns (64- or 128-wide tiles)
        dyb = torch.randn(T, N + ldx_pad, generator=g).to(dev)
        dy = dyb[:, :N]
        dx = torch.full((T, K), float("nan"), device=dev)
        _lib.call("dhz_linear_dgrad_split6", dy.data_ptr(), N + ldx_pad, pl[0].data_ptr(), pl[1].data_ptr(), pl[2].data_ptr(),
                  dx.data_ptr(), K, T, N, K, s)
        ref = dy.double() @ w.double()
        mag = dy.double().abs() @ w.double().abs()
        err = (dx.double() - ref).abs()
        assert (err <= BOUND[6] * mag).all(), (err / mag).max().item()
        dx32 = torch.empty(T, K, device=dev)
        _lib.call("dhz_linear_dgrad", dy.data_ptr(), N + ldx_pad, w.data_ptr(), dx32.data_ptr(), K, T, N, K, s)
        e32 = (dx32.double() - ref).abs().max().item()
        assert err.max().item() < 4 * max(e32, 1e-7), (err.max().item(), e32)


def test_split6_gemm_is_deterministic_and_rejects_bad_shapes():
    from dehaze_hip import _lib
    dev = torch.device("cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    T, K, N = 8192, 256, 512
    x = torch.randn(T, K, device=dev)
    w = torch.randn(N, K, device=dev) / 16
    pl = _planes(w)
    ys = []
    for _ in range(3):
        y = torch.empty(T, N, device=dev)
        _lib.call("dhz_linear_fwd_split6", x.data_ptr(), K, pl[0].data_ptr(), pl[1].data_ptr(), pl[2].data_ptr(), None, y.data_ptr(),
                  N, T, N, K, s)
        ys.append(y)
    assert torch.equal(ys[0], ys[1]) and torch.equal(ys[0], ys[2])
    with pytest.raises(_lib.DehazeHipError, match="multiple of 32"):
        _lib.call("dhz_linear_fwd_split6", x.data_ptr(), K, pl[0].data_ptr(), pl[1].data_ptr(), pl[2].data_ptr(), None, y.data_ptr(),
                  N, T, N, 48, s)
    with pytest.raises(_lib.DehazeHipError, match="null pointer"):
        _lib.call("dhz_linear_fwd_split6", x.data_ptr(), K, None, pl[1].data_ptr(), pl[2].data_ptr(), None, y.data_ptr(), N, T, N, K, s)


def test_split_planes_shadow_follows_optimizer_and_outside_writes():
    """FlatAdamW keeps the three planes of its flat parameter buffer: (a) current after every step (hi + mid + lo == parameters),
    (b) ops.split_planes hands out VIEWS of them for weights that live in the flat buffer (the packed Q / K / V operand included),
    (c) a parameter written outside the optimizer reaches them before the next forward - train_step's or a bare model(x)."""
    import My_model_1 as M1
    from dehaze_hip import ops
    from dehaze_hip.train import FlatAdamW, synthetic_batch, train_step
    from losses import CharbonnierLoss
    if ops.SPLIT_BF16 != 6:
        pytest.skip("the six-term split is not the active arithmetic (DHZ_SPLIT_BF16)")
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    model = M1.Uformer(img_size=128, embed_dim=32, win_size=8, token_projection='linear', token_mlp='leff',
                       drop_path_rate=0.).to(dev).train()
    opt = FlatAdamW(model, lr=2e-4, weight_decay=0.02)
    gt, hazy = synthetic_batch(1, 128, seed=3, device=dev)
    train_step(model, CharbonnierLoss(), None, opt, None, hazy, gt, 1.0, 0.0)
    f = opt._flat
    rec = lambda: f["p3"][0].double() + f["p3"][1].double() + f["p3"][2].double()
    assert ops.SPLIT_SHADOW is not None and ops.SPLIT_SHADOW[0] is f["p"]
    assert torch.equal(rec(), f["p"].double())                                           # (a)
    W = model.encoderlayer_2.blocks[0].mlp.linear1[0].weight
    hi, mid, lo = ops.split_planes(W.detach())
    assert hi.data_ptr() - f["p3"][0].data_ptr() == (W.data_ptr() - f["p"].data_ptr()) // 2    # (b) a view, no launch
    assert torch.equal((hi.double() + mid.double() + lo.double()).view_as(W), W.detach().double())
    # (b') the transposed planes of the backward-data GEMMs: W^T at W's offset; the packed Q / K / V weights as one [3C, C] matrix
    ht, mt, lt = ops.split_planes_t(W.detach())
    assert torch.equal((ht.double() + mt.double() + lt.double()).view(W.shape[1], W.shape[0]), W.detach().double().t())
    al = model.encoderlayer_3.blocks[1].attn.ProbSpare
    wq, wk, wv = al.query_projection.weight, al.key_projection.weight, al.value_projection.weight
    Wp = ops.cat_rows([wq.detach(), wk.detach(), wv.detach()])
    assert Wp.data_ptr() == wq.data_ptr() and Wp.shape == (3 * wq.shape[0], wq.shape[1])          # a view of the flat buffer
    ht, mt, lt = ops.split_planes_t(Wp)
    assert torch.equal((ht.double() + mt.double() + lt.double()).view(Wp.shape[1], Wp.shape[0]), Wp.double().t())
    assert ops.split_planes_t(wq.detach()) is None                                       # (not registered on its own)
    dyy = torch.randn(8192, W.shape[0], device=dev)
    d1 = ops.gemm_dgrad(dyy, W.detach())                                                 # forward kernel on the planes of W^T
    ops._NO_TPLANES = True
    try:
        d2 = ops.gemm_dgrad(dyy, W.detach())                                             # transposed-read kernel on the planes of W
    finally:
        ops._NO_TPLANES = False
    assert torch.equal(d1, d2)                                                           # the same products in the same order
    with torch.no_grad():
        W.mul_(1.5)                                                                       # (c) an outside write ...
    assert not torch.equal(rec(), f["p"].double())
    model.eval()
    with torch.no_grad():
        model(hazy)                                                                       # ... reaches the planes in a bare forward
    assert torch.equal(rec(), f["p"].double())
    ht, mt, lt = ops.split_planes_t(W.detach())
    assert torch.equal((ht.double() + mt.double() + lt.double()).view(W.shape[1], W.shape[0]), W.detach().double().t())
    model.train()
    with torch.no_grad():
        W.mul_(0.5)
    train_step(model, CharbonnierLoss(), None, opt, None, hazy, gt, 1.0, 0.0)             # ... and in train_step
    assert torch.equal(rec(), f["p"].double())
    # (d) every LOOKUP validates itself: a bare GEMM on a weight written outside the optimizer (no forward, no train_step in
    #     between) multiplies with the current weight - also through the packed Q / K / V view, whose second and third parts are
    #     parameters with version counters of their own
    xx = torch.randn(4096, W.shape[1], device=dev)
    with torch.no_grad():
        W.mul_(-2.0)
    assert not torch.equal(rec(), f["p"].double())
    y1 = ops.gemm_fwd(xx, W.detach())
    assert torch.equal(rec(), f["p"].double())
    ref = xx.double() @ W.detach().double().t()
    assert (y1.double() - ref).abs().max().item() < 1e-5 * ref.abs().max().item()
    with torch.no_grad():
        wk.add_(0.25)                                                                      # the MIDDLE third of the packed operand
    xq = torch.randn(4096, Wp.shape[1], device=dev)
    y2 = ops.gemm_fwd(xq, ops.cat_rows([wq.detach(), wk.detach(), wv.detach()]))
    ref = xq.double() @ torch.cat([wq, wk, wv]).detach().double().t()
    assert (y2.double() - ref).abs().max().item() < 1e-5 * ref.abs().max().item()
    d3 = ops.gemm_dgrad(dyy, W.detach())
    refd = dyy.double() @ W.detach().double()
    assert (d3.double() - refd).abs().max().item() < 1e-5 * refd.abs().max().item()
    # writes that bump no counter need the forced form (documented in ops.py)
    W.data.mul_(3.0)
    assert not torch.equal(rec(), f["p"].double())
    opt.sync_shadows(force=True)
    assert torch.equal(rec(), f["p"].double())
    # (e) the registration is weak: dropping the optimizer un-registers the copies (and frees its buffers)
    import gc
    del opt, f, rec
    gc.collect()
    assert ops._shadow_owner() is None and ops.SPLIT_SHADOW is None
