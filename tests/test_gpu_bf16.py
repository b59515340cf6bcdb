"""-m gpu: the bf16 path of BASELINE config 4 (bf16 activations / bf16 weight copies in HBM, fp32 accumulation, fp32 sparsity
measure / ranking / softmax).  Every check runs the fp64 (or fp32-kernel) reference on the SAME bf16-rounded inputs, so the
tolerance only has to cover the fp32 accumulation order and the final rounding of the outputs to bf16 (relative 2^-8)."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
EPS = 2.0 ** -8            # bf16 unit round-off (8 significant bits)


def _s():
    return torch.cuda.current_stream().cuda_stream


@pytest.mark.parametrize("T,N,K", [(4096, 64, 64), (8192, 192, 64), (4096, 256, 64), (2048, 64, 256), (1024, 384, 128),
                                   (1000, 512, 128), (512, 1536, 512), (64, 2048, 512), (300, 512, 2048), (131072, 128, 128),
                                   (1, 64, 64), (65536, 256, 64),
                                   # the software-pipelined 256 x 128 kernel (csrc/gemm_bf16_pipe.hip: a tile per CU and more): a ragged
                                   # last row block, one stage per tile, sixteen stages per tile, a result written with non-temporal stores
                                   (70000, 256, 128), (66000, 128, 64), (33000, 256, 1024), (131072, 768, 64)])
def test_linear_bf16_c_abi(T, N, K):
    """dhz_linear_fwd_bf16 / dhz_linear_dgrad_bf16 through the raw C-ABI against fp64 matmuls of the same bf16 operands."""
    from dehaze_hip import _lib
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(T + 3 * N + K)
    x = torch.randn(T, K + 64, generator=g).to(dev).to(BF)                 # x = first K columns of a wider buffer
    W = (torch.randn(N, K, generator=g) * 0.1).to(dev).to(BF)
    b = (torch.randn(N, generator=g) * 0.1).to(dev)
    y = torch.full((T, N + 64), 7.0, device=dev, dtype=BF)
    _lib.call("dhz_linear_fwd_bf16", x.data_ptr(), x.stride(0), W.data_ptr(), b.data_ptr(), y.data_ptr() + 2 * 32, y.stride(0), T,
              N, K, _s())
    ref = x[:, :K].double() @ W.double().t() + b.double()
    err = (y[:, 32:32 + N].double() - ref).abs()
    assert (err <= 1.1 * EPS * ref.abs() + 2e-5 * K ** 0.5).all(), err.max().item()
    assert (y[:, :32] == 7.0).all() and (y[:, 32 + N:] == 7.0).all()
    dy = torch.randn(T, N, generator=g).to(dev).to(BF)
    dx = torch.empty(T, K, device=dev, dtype=BF)
    _lib.call("dhz_linear_dgrad_bf16", dy.data_ptr(), N, W.data_ptr(), dx.data_ptr(), K, T, N, K, _s())
    refd = dy.double() @ W.double()
    errd = (dx.double() - refd).abs()
    assert (errd <= 1.1 * EPS * refd.abs() + 2e-5 * N ** 0.5).all(), errd.max().item()
    lib = _lib.load()
    assert lib.dhz_linear_fwd_bf16(x.data_ptr(), x.stride(0), W.data_ptr(), None, y.data_ptr(), y.stride(0), T, N + 32, K, _s()) == -22


@pytest.mark.parametrize("T,n,N,K", [(4096, 1, 64, 64), (8192, 3, 64, 64), (4096, 1, 256, 64), (2048, 1, 64, 256),
                                     (8192, 3, 128, 128), (1024, 1, 512, 128), (512, 1, 1536, 512), (65536, 1, 128, 128),
                                     (320, 1, 2048, 512), (131072, 3, 64, 64),
                                     (4736, 1, 128, 128)])      # 18 slabs of 4 or 5 stages: paired groups with unequal trip counts
def test_linear_wgrad_bf16_c_abi(T, n, N, K):
    """dW_i += dy[:, iN:(i+1)N]^T x, db_i += column sums, for n parameters sharing x: fp32 accumulation of bf16 products."""
    from dehaze_hip import _lib
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(T + N + K + n)
    dy = torch.randn(T, n * N, generator=g).to(dev).to(BF)
    x = torch.randn(T, K, generator=g).to(dev).to(BF)
    dws = [torch.full((N, K), 0.5, device=dev) for _ in range(n)]        # accumulated INTO
    dbs = [torch.full((N,), -1.0, device=dev) for _ in range(n)]
    aw = (ctypes.c_void_p * n)(*[w.data_ptr() for w in dws])
    ab = (ctypes.c_void_p * n)(*[b.data_ptr() for b in dbs])
    _lib.call("dhz_linear_wgrad_bf16", dy.data_ptr(), dy.stride(0), x.data_ptr(), K, T, n, N, K, ctypes.cast(aw, ctypes.c_void_p),
              ctypes.cast(ab, ctypes.c_void_p), _s())
    ref = dy.double().t() @ x.double()
    refb = dy.double().sum(0)
    tol = 3e-6 * T ** 0.5 + 1e-5
    for i in range(n):
        assert (dws[i].double() - 0.5 - ref[i * N:(i + 1) * N]).abs().max() < tol * max(1.0, ref.abs().max().item() / T ** 0.5)
        assert (dbs[i].double() + 1.0 - refb[i * N:(i + 1) * N]).abs().max() < tol * 4


def _bf(t):
    return t.to(BF)


def test_streaming_kernels_bf16_equal_fp32_kernels_on_rounded_inputs():
    """LayerNorm+partition, reverse+residual, LeFF depthwise stage, ProbSparse core - forward and backward - with bf16 token
    tensors: every kernel converts to fp32 on load and runs the fp32 kernel's arithmetic, so on bf16-representable inputs the
    result must be the fp32 kernel's result rounded to bf16 - bit for bit (parameter gradients, fp32 atomics: to rounding)."""
    from dehaze_hip import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    B, H, W, C, shift = 2, 16, 16, 64, 4

    def both(fn, *tensors):
        """run fn on bf16 tensors and on their fp32 values; returns (outs_bf16, outs_fp32_rounded)"""
        tb = [t.to(dev).to(BF).requires_grad_() for t in tensors]
        tf = [t.detach().float().requires_grad_() for t in tb]
        return fn(*tb), fn(*tf), tb, tf

    x = torch.randn(B, H * W, C, generator=g)
    gamma, beta = (1 + 0.1 * torch.randn(C, generator=g)).to(dev), (0.1 * torch.randn(C, generator=g)).to(dev)
    gout = torch.randn(B * H * W, C, generator=g).to(dev)
    # LayerNorm + roll + partition, fwd + bwd (dx; dgamma/dbeta fp32)
    gm_b, gm_f = gamma.clone().requires_grad_(), gamma.clone().requires_grad_()
    bt_b, bt_f = beta.clone().requires_grad_(), beta.clone().requires_grad_()
    xb = x.to(dev).to(BF).requires_grad_()
    xf = xb.detach().float().requires_grad_()
    yb = ops.ln_partition(xb, gm_b, bt_b, H, W, shift)
    yf = ops.ln_partition(xf, gm_f, bt_f, H, W, shift)
    # (the bf16 LayerNorm kernels give a lane 8 channels instead of 4, so their row sums add up in another order than the fp32
    # kernels': equal up to one bf16 step on a few elements whose fp32 value sits on a rounding boundary)
    def one_step(a, b):
        d = (a.float() - b.float()).abs()
        return (a != b).float().mean().item() < 2e-3 and bool((d <= 2.0 ** -7 * torch.maximum(a.float().abs(), b.float().abs())).all())
    assert yb.dtype == BF and one_step(yb, yf.to(BF))
    gb = gout.to(BF)
    yb.backward(gb)
    yf.backward(gb.float())
    assert one_step(xb.grad, xf.grad.to(BF))
    assert torch.allclose(gm_b.grad, gm_f.grad, rtol=1e-4, atol=1e-4) and torch.allclose(bt_b.grad, bt_f.grad, rtol=1e-4, atol=1e-4)
    # window reverse + un-roll + residual with a DropPath vector
    sc = torch.tensor([1.0 / 0.9, 0.0], device=dev)
    (ob, of, tb, tf) = both(lambda a, s_: ops.reverse_residual(a, s_, sc, H, W, shift), torch.randn(B * H * W, C, generator=g), x)
    assert ob.dtype == BF and torch.equal(ob, of.to(BF))
    ob.backward(gb.view(B, H * W, C)); of.backward(gb.float().view(B, H * W, C))
    assert torch.equal(tb[0].grad, tf[0].grad.to(BF)) and torch.equal(tb[1].grad, tf[1].grad.to(BF))
    # LeFF depthwise stage
    Ch = 4 * C
    wd, bd = (0.3 * torch.randn(Ch, 1, 3, 3, generator=g)).to(dev), (0.1 * torch.randn(Ch, generator=g)).to(dev)
    wd_b, wd_f, bd_b, bd_f = (t.clone().requires_grad_() for t in (wd, wd, bd, bd))
    u = torch.randn(B, H * W, Ch, generator=g)
    ub = u.to(dev).to(BF).requires_grad_()
    uf = ub.detach().float().requires_grad_()
    zb, zf = ops.leff_dwconv(ub, wd_b, bd_b, H, W), ops.leff_dwconv(uf, wd_f, bd_f, H, W)
    assert zb.dtype == BF and torch.equal(zb, zf.to(BF))
    gz = torch.randn(B, H * W, Ch, generator=g).to(dev).to(BF)
    zb.backward(gz)
    zf.backward(gz.float())
    # the bf16 chain also rounds the saved gelu'(t) to bf16: du agrees to that rounding, not bit for bit
    assert (ub.grad.float() - uf.grad).abs().max() <= 3 * EPS * uf.grad.abs().max()
    assert torch.allclose(wd_b.grad, wd_f.grad, rtol=2e-2, atol=2e-2 * wd_f.grad.abs().max().item())
    # ProbSparse core on a packed QKV buffer, head_dim 64 (config 4), bias table + shift mask
    heads, d = 2, 64
    Cq = heads * d
    qkv = torch.randn(B * H * W, 3 * Cq, generator=g)
    table = (0.3 * torch.randn(225, heads, generator=g)).to(dev)
    tb_, tf_ = table.clone().requires_grad_(), table.clone().requires_grad_()
    idx = torch.randint(64, (64, 25), generator=g).to(torch.uint8).to(dev)
    mask = ops.shift_mask(H, W, shift, dev)
    qb = qkv.to(dev).to(BF).requires_grad_()
    qf = qb.detach().float().requires_grad_()
    cb, cf = ops.ps_window_attention(qb, tb_, idx, mask, heads, d), ops.ps_window_attention(qf, tf_, idx, mask, heads, d)
    # (with bf16 storage the products of two stored tensors - Q K^T, dO V^T - run on the bf16 matrix pipe: exact products, another
    # order of the fp32 sums than the fp32 kernel's: equal up to one bf16 step on a few elements)
    assert cb.dtype == BF and one_step(cb, cf.to(BF))                         # same selection, same context
    gc = torch.randn(B * H * W, Cq, generator=g).to(dev).to(BF)
    cb.backward(gc)
    cf.backward(gc.float())
    assert one_step(qb.grad, qf.grad.to(BF))
    assert torch.allclose(tb_.grad, tf_.grad, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("C,heads,shift", [(64, 1, 0), (128, 2, 4)])
def test_block_bf16_vs_oracle_fp32_on_rounded_inputs(C, heads, shift):
    """One LeWin block of config 4 (head_dim 64) with bf16 activations against the fp32 CPU oracle given the SAME bf16-rounded
    input and the bf16-rounded weights the GEMMs actually use.  Stated tolerance: every kernel boundary rounds its output
    to bf16 (relative 2^-8 = 3.9e-3), ~10 boundaries in a block -> norm-wise relative error <= 2 % on the block output and
    <= 4 % on dx, 6 % on parameter gradients (15 % on those that only flow through the 25 selected rows of a window: bias table,
    query / key projections - fewer terms to average the rounding over, and a near-tie of the measure between two bf16-rounded
    rows can still be broken differently by the accumulation order of the fp32 matrix pipe and of the oracle's matmul)."""
    import My_model_1 as M1
    from oracle import uformer_oracle as O
    dev = torch.device("cuda:0")
    torch.manual_seed(C + shift)
    res, B = 16, 2
    blk = M1.LeWinTransformerBlock(dim=C, input_resolution=(res, res), num_heads=heads, win_size=8, shift_size=shift,
                                   token_mlp='leff', drop_path=0.)
    with torch.no_grad():
        for n, p in blk.named_parameters():
            if p.ndim >= 2 and "table" not in n:
                p.copy_(p.to(BF).float())                 # weights bf16-representable: the bf16 copies are exact
    x = torch.randn(B, res * res, C).to(BF).float()
    gout = torch.randn(B, res * res, C).to(BF).float()
    idx = torch.randint(64, (64, 25))
    P = {"b." + k: v.detach().clone().requires_grad_(v.dtype.is_floating_point) for k, v in blk.state_dict().items()}
    xo = x.clone().requires_grad_()
    # The HIP path hands Q, K, V to the attention core ROUNDED to bf16 (the packed QKV buffer is bf16), and the core ranks what it
    # is given.  The oracle is fed the same rounding at that boundary (straight-through for the gradient), so that both rank the
    # same numbers: a top-25 set then differs only at a genuine tie, and the selection-path gradients can be held to the same
    # bound as the others instead of the 25 % a flipped selection needs.
    real_core = O.prob_attention

    class _RoundBF(torch.autograd.Function):
        @staticmethod
        def forward(ctx, t):
            return t.to(BF).float()

        @staticmethod
        def backward(ctx, g):
            return g

    def rounded_core(q, k, v, *a, **kw):
        return real_core(_RoundBF.apply(q), _RoundBF.apply(k), _RoundBF.apply(v), *a, **kw)
    O.prob_attention = rounded_core
    try:
        yo = O.lewin_block(xo, P, "b.", heads, win=8, shift=shift, idx=idx, drop_path=0.0, training=True)
    finally:
        O.prob_attention = real_core
    (yo * gout).sum().backward()
    blk.to(dev).train()
    xd = x.to(dev).to(BF).requires_grad_()
    blk._staged_idx = idx.to(torch.uint8).to(dev)
    y = blk(xd)
    assert y.dtype == BF
    (y.float() * gout.to(dev)).sum().backward()

    def rel(a, b):
        return ((a.double().cpu() - b.double()).norm() / b.double().norm()).item()
    assert rel(y.float(), yo.detach()) < 2e-2, rel(y.float(), yo.detach())
    assert rel(xd.grad.float(), xo.grad) < 4e-2, rel(xd.grad.float(), xo.grad)
    for n, p in blk.named_parameters():
        go = P["b." + n].grad
        if go is not None and p.grad is not None and go.abs().max() > 1e-5:     # (d key bias == 0 analytically)
            assert p.grad.dtype == torch.float32
            # gradients that only flow through the 25 selected query rows of each window (bias table, query / key projections)
            # are the quantities most sensitive to a selection flipped at a near-tie: the bf16 rounding of Q and K moves the
            # sparsity measure by ~2^-8 of the scores, the oracle ranks the unrounded ones
            # (the oracle ranks the same bf16-rounded Q and K, see above; before that alignment these needed 25 %)
            sel = any(k in n for k in ("table", "query_projection", "key_projection"))
            assert rel(p.grad, go) < (0.15 if sel else 6e-2), (n, rel(p.grad, go))


def test_config4_train_step_bf16():
    """BASELINE config 4's model (E = 64, head_dim 64) taking training steps with bf16 activations: fp32 master weights and
    fp32 gradients in the flat buffers, loss within 2 % of the fp32 step on the same weights / batch / sampled keys, loss going
    down over a few steps.  (ps = 128 and two patches here; bench.py --embed_dim 64 --ps 256 --batch 8 --dtype bf16 runs the full size.)"""
    import My_model_1 as M1
    from dehaze_hip.train import FlatAdamW, synthetic_batch, train_step
    from losses import CharbonnierLoss
    dev = torch.device("cuda:0")
    gt, hazy = synthetic_batch(2, 128, seed=3, device=dev)
    losses = {}
    for dt in (torch.float32, BF):
        torch.manual_seed(1234)
        model = M1.Uformer(img_size=128, embed_dim=64, win_size=8, token_projection='linear', token_mlp='leff',
                           drop_path_rate=0.).to(dev).train()
        model.act_dtype = dt
        opt = FlatAdamW(model, lr=2e-4)
        torch.manual_seed(7)
        losses[dt] = [train_step(model, CharbonnierLoss(), None, opt, None, hazy, gt, 1.0, 0.0)[0].item() for _ in range(4)]
        assert all(p.grad is None or p.grad.dtype == torch.float32 for p in model.parameters())
    f, b = losses[torch.float32], losses[BF]
    assert abs(b[0] - f[0]) < 2e-2 * f[0], (b, f)
    assert b[-1] < b[0] and abs(b[-1] - f[-1]) < 5e-2 * f[-1], (b, f)


def test_config4_model_bf16_vs_oracle_psnr():
    """Model level, against the ORACLE (not against this build's fp32 path): the E = 64 model's eval forward with bf16 activations
    vs the fp32 CPU oracle on the same weights, input and sampled keys.  Stated tolerance: bf16 storage rounds every kernel
    boundary to 8 significant bits; through the 18 blocks the restored image stays within PSNR >= 35 dB of the fp32 reference
    image, and its PSNR against the ground truth differs from the reference's by < 0.1 dB (the 0.01 dB contract of the fp32 path
    does not apply to reduced-precision storage; the reference's own fp16 autocast, TR:224, moves it as much)."""
    import My_model_1 as M1
    from oracle import uformer_oracle as O
    from dehaze_hip.train import synthetic_batch
    dev = torch.device("cuda:0")
    torch.manual_seed(1234)
    model = M1.Uformer(img_size=128, embed_dim=64, win_size=8, token_projection='linear', token_mlp='leff').to(dev).eval()
    P = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    gt, hazy = synthetic_batch(2, 128, seed=11)
    torch.manual_seed(77)
    with torch.no_grad():
        ref = O.uformer_forward(P, hazy).clamp(0, 1)
    model.act_dtype = BF
    torch.manual_seed(77)
    with torch.no_grad():
        out = model(hazy.to(dev)).float().clamp(0, 1).cpu()

    def psnr(a, b):
        return (10 * torch.log10(1.0 / ((a - b) ** 2).mean())).item()
    assert psnr(out, ref) > 35.0, psnr(out, ref)
    assert abs(psnr(out, gt) - psnr(ref, gt)) < 0.1, (psnr(out, gt), psnr(ref, gt))


def test_config4_full_size_bf16_properties():
    """BASELINE config 4 AT FULL SIZE and in its dtype (E = 64, 256 x 256 patches, 8 per GPU, bf16 activations), where the CPU
    oracle is too slow to be the checker - size-independent properties instead: (1) a patch restored inside the batch equals the
    same patch restored alone to within bf16 rounding of a [0, 1] image: max |difference| <= 2^-6, mean < 1e-3 (this build's own
    kernels accumulate per token in an order that does not depend on the batch size, but the GEMM tile shape follows the token
    count, so the summation order of a product can differ between the two runs: an fp32 ulp there becomes a bf16 step downstream
    - measured max 4.4e-3, mean 4.8e-4); (2) the fp32
    gradient of the 8-patch batch equals the mean of the gradients of its halves (the loss scale differs by a factor of two, which
    commutes with bf16 rounding; bound 2 % norm-wise, for the same reason as (1)); (3) one finite step that moves the weights."""
    import My_model_1 as M1
    from dehaze_hip.train import FlatAdamW, synthetic_batch, train_step
    from losses import CharbonnierLoss
    dev = torch.device("cuda:0")
    torch.manual_seed(1234)
    model = M1.Uformer(img_size=256, embed_dim=64, win_size=8, token_projection='linear', token_mlp='leff', drop_path_rate=0.).to(dev)
    model.act_dtype = BF
    gt, x = synthetic_batch(8, 256, seed=23, device=dev)
    model.eval()
    with torch.no_grad():
        torch.manual_seed(5)
        y = model(x)
        for i in (0, 7):
            torch.manual_seed(5)
            yi = model(x[i:i + 1])
            d = (y[i:i + 1].float() - yi.float()).abs()
            assert d.max().item() <= 2.0 ** -6 and d.mean().item() < 1e-3, (i, d.max().item(), d.mean().item())
    assert torch.isfinite(y.float()).all() and y.shape == (8, 3, 256, 256)
    model.train()
    char = CharbonnierLoss().to(dev)
    params = [p for _, p in model.live_parameters()]

    def grad_of(sl):
        for p in params:
            p.grad = None
        torch.manual_seed(9)                                  # same sampled keys for every call
        loss = char(model(x[sl]).float().clamp(0, 1), gt[sl])
        loss.backward()
        return torch.cat([p.grad.detach().float().reshape(-1) for p in params]), loss.item()
    g_full, l_full = grad_of(slice(0, 8))
    g_a, l_a = grad_of(slice(0, 4))
    g_b, l_b = grad_of(slice(4, 8))
    assert abs(l_full - 0.5 * (l_a + l_b)) < 2e-4 * l_full      # (library convolutions by batch size, see above)
    rel = ((g_full - 0.5 * (g_a + g_b)).norm() / g_full.norm()).item()
    assert rel < 2e-2, rel
    for p in params:
        p.grad = None
    opt = FlatAdamW(model, lr=2e-4, weight_decay=0.02)
    before = [p.detach().clone() for p in params[:4]]
    loss, _, _ = train_step(model, char, None, opt, None, x, gt, w_cr=0.0)
    assert torch.isfinite(loss) and 0.0 < loss.item() < 1.0
    assert any(not torch.equal(a, b.detach()) for a, b in zip(before, params[:4]))


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 16, 16, 64, 128), (1, 32, 16, 128, 256)])
def test_downsample_bf16_vs_fp64_conv(B, H, W, Cin, Cout):
    """Downsample (Conv2d k4 s2 p1, M1:606-622) on bf16 tokens - patch matrix (dhz_im2col_k4s2_bf16), bf16-MFMA GEMMs, dhz_col2im_k4s2_bf16 -
    against torch's fp64 conv2d on the same bf16-rounded input / weights: output within bf16 rounding (2^-8 relative + accumulation
    slack), dx within two roundings (dcol is stored in bf16 before the 4-term sum), dw / db (fp32 accumulation) to 1e-3."""
    import torch.nn.functional as F
    from dehaze_hip import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(B + H + Cin)
    x = torch.randn(B, H * W, Cin, generator=g).to(BF)
    w = (torch.randn(Cout, Cin, 4, 4, generator=g) * 0.05).to(BF).float()
    b = torch.randn(Cout, generator=g) * 0.1
    gy = torch.randn(B, (H // 2) * (W // 2), Cout, generator=g).to(BF)
    x64 = x.double().view(B, H, W, Cin).permute(0, 3, 1, 2).requires_grad_()
    w64, b64 = w.double().requires_grad_(), b.double().requires_grad_()
    y64 = F.conv2d(x64, w64, b64, stride=2, padding=1)
    (y64 * gy.double().view(B, H // 2, W // 2, Cout).permute(0, 3, 1, 2)).sum().backward()
    ref_y = y64.detach().permute(0, 2, 3, 1).reshape(B, -1, Cout)
    ref_dx = x64.grad.permute(0, 2, 3, 1).reshape(B, H * W, Cin)
    xd = x.to(dev).requires_grad_()
    wd, bd = w.to(dev).requires_grad_(), b.to(dev).requires_grad_()
    y = ops.conv4s2_tokens(xd, wd, bd, H, W)
    assert y.dtype == BF and y.shape == (B, (H // 2) * (W // 2), Cout)
    y.backward(gy.to(dev))
    err = (y.double().cpu() - ref_y).abs()
    assert (err <= 1.1 * EPS * ref_y.abs() + 1e-4 * (16 * Cin) ** 0.5).all(), err.max().item()
    edx = (xd.grad.double().cpu() - ref_dx).abs()
    assert (edx <= 4 * EPS * ref_dx.abs() + 4 * EPS * ref_dx.abs().max()).all(), edx.max().item()
    assert torch.allclose(wd.grad.double().cpu(), w64.grad, rtol=1e-3, atol=1e-3 * w64.grad.abs().max().item())
    assert torch.allclose(bd.grad.double().cpu(), b64.grad, rtol=1e-3, atol=1e-3 * b64.grad.abs().max().item())


def test_projections_bf16_tokens_match_fp32_kernels():
    """InputProj / OutputProj with bf16 token storage: the same arithmetic as the fp32 kernels, rounded once on the token side.
    InputProj: bf16 output == fp32 output rounded; weight gradients from a bf16 dy / saved bf16 y equal the fp32 kernel's on the
    same (bf16-representable) values.  OutputProj: fp32 image from bf16 tokens == fp32 kernel on the same values; dx == rounded."""
    from dehaze_hip import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    B, H, W, E = 2, 32, 48, 64
    img = torch.rand(B, 3, H, W, generator=g).to(dev)
    w = (torch.randn(E, 3, 3, 3, generator=g) * 0.2).to(dev).requires_grad_()
    b = (torch.randn(E, generator=g) * 0.1).to(dev).requires_grad_()
    y32 = ops.input_proj(img, w, b, 0.01)
    y16 = ops.input_proj(img, w, b, 0.01, BF)
    assert y16.dtype == BF and torch.equal(y16, y32.to(BF))
    gy = torch.randn(B, H * W, E, generator=g).to(dev).to(BF)
    w2, b2 = w.detach().clone().requires_grad_(), b.detach().clone().requires_grad_()
    y16b = ops.input_proj(img, w2, b2, 0.01, BF)
    y16b.backward(gy)
    # fp32 kernel on the same numbers: its saved output must carry the same signs -> feed it the rounded output's pre-image
    w3, b3 = w.detach().clone().requires_grad_(), b.detach().clone().requires_grad_()
    y32b = ops.input_proj(img, w3, b3, 0.01)
    same_sign = ((y32b > 0) == (y16b.float() > 0)).all().item()
    y32b.backward(gy.float())
    if same_sign:
        assert torch.allclose(w2.grad, w3.grad, rtol=1e-4, atol=1e-4 * w3.grad.abs().max().item())
        assert torch.allclose(b2.grad, b3.grad, rtol=1e-4, atol=1e-4 * b3.grad.abs().max().item())
    C = 128
    x = torch.randn(B, H * W, C, generator=g).to(dev).to(BF)
    wo = (torch.randn(3, C, 3, 3, generator=g) * 0.05).to(dev)
    bo = (torch.randn(3, generator=g) * 0.1).to(dev)
    outs = {}
    for dt in (torch.float32, BF):
        xx = x.to(dt).requires_grad_()
        ww, bb = wo.clone().requires_grad_(), bo.clone().requires_grad_()
        yy = ops.thin_conv3x3(xx, ww, bb, H, W)
        assert yy.dtype == torch.float32
        gi = torch.randn(B, 3, H, W, generator=torch.Generator().manual_seed(9)).to(dev)
        yy.backward(gi)
        outs[dt] = (yy.detach(), xx.grad, ww.grad, bb.grad)
    assert torch.equal(outs[BF][0], outs[torch.float32][0])
    assert outs[BF][1].dtype == BF and torch.equal(outs[BF][1], outs[torch.float32][1].to(BF))
    assert torch.allclose(outs[BF][2], outs[torch.float32][2], rtol=1e-5, atol=1e-5 * outs[torch.float32][2].abs().max().item())
    assert torch.allclose(outs[BF][3], outs[torch.float32][3], rtol=1e-5, atol=1e-4)


def test_bf16_shadow_written_by_optimizer_and_resynced_after_outside_writes():
    """The bf16 weight copy the GEMMs read: (a) the AdamW kernel writes it in its own pass - bit-equal to a cast of the updated
    parameters, no separate cast launch; (b) a parameter written in place by anything else (a loaded checkpoint, a landscape
    probe) reaches the shadow at the start of the next train_step; (c) without such a write the start-of-step check does nothing."""
    import My_model_1 as M1
    from dehaze_hip import ops
    from dehaze_hip.train import FlatAdamW, synthetic_batch, train_step
    from losses import CharbonnierLoss
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    model = M1.Uformer(img_size=128, embed_dim=64, win_size=8, token_projection='linear', token_mlp='leff',
                       drop_path_rate=0.).to(dev).train()
    model.act_dtype = BF
    opt = FlatAdamW(model, lr=2e-4, weight_decay=0.02)
    gt, hazy = synthetic_batch(1, 128, seed=3, device=dev)
    train_step(model, CharbonnierLoss(), None, opt, None, hazy, gt, 1.0, 0.0)
    f = opt._flat
    assert ops.BF16_SHADOW is not None and ops.BF16_SHADOW[0] is f["p"]
    assert torch.equal(f["p16"], f["p"].to(BF))                                    # (a)
    p = next(model.parameters())
    with torch.no_grad():
        p.add_(0.5)                                                                # (b) an outside write
    assert not torch.equal(f["p16"], f["p"].to(BF))
    seen = []
    orig = ops.refresh_bf16_shadow
    ops.refresh_bf16_shadow = lambda: (seen.append(1), orig())[1]
    try:
        opt.sync_shadows()
        assert seen == [1] and torch.equal(f["p16"], f["p"].to(BF))
        seen.clear()
        train_step(model, CharbonnierLoss(), None, opt, None, hazy, gt, 1.0, 0.0)  # (c) nothing written outside: no cast launch
        assert seen == [], seen
        assert torch.equal(f["p16"], f["p"].to(BF))
        # (d) a forward OUTSIDE train_step after an outside write (eval after load_state_dict, a landscape probe): the parameters
        # are views of the flat buffer with version counters of their own - Uformer.forward asks the optimizer to re-derive
        with torch.no_grad():
            p.add_(0.25)
        model.eval()
        with torch.no_grad():
            model(hazy)
        assert seen == [1] and torch.equal(f["p16"], f["p"].to(BF))
    finally:
        ops.refresh_bf16_shadow = orig


def test_bf16_transposed_shadow_and_backward_data_on_the_forward_kernel():
    """The optimizer also keeps bf16 copies of every Linear weight's TRANSPOSE (dhz_bf16_transpose_batched; Q | K | V as one packed
    matrix): they equal the transposed bf16 copy after a step and after an outside write, and ops.gemm_dgrad - which runs the forward
    kernel on them - returns what the transposed-read backward-data kernel returns."""
    import My_model_1 as M1
    from dehaze_hip import _lib, ops
    from dehaze_hip.train import FlatAdamW, synthetic_batch, train_step
    from losses import CharbonnierLoss
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    model = M1.Uformer(img_size=128, embed_dim=64, win_size=8, token_projection='linear', token_mlp='leff',
                       drop_path_rate=0.).to(dev).train()
    model.act_dtype = BF
    opt = FlatAdamW(model, lr=2e-4, weight_decay=0.02)
    gt, hazy = synthetic_batch(1, 128, seed=5, device=dev)
    train_step(model, CharbonnierLoss(), None, opt, None, hazy, gt, 1.0, 0.0)
    f = opt._flat
    assert "p16t" in f and len(f["p16t_index"]) > 50

    def check_all():
        for off, R, Cc in sorted(f["p16t_index"]):
            assert torch.equal(f["p16t"][off: off + R * Cc].view(Cc, R), f["p16"][off: off + R * Cc].view(R, Cc).t()), (off, R, Cc)
    check_all()
    blk = model.encoderlayer_1.blocks[0]
    with torch.no_grad():
        blk.mlp.linear1[0].weight.mul_(1.5)                                       # an outside write
    opt.sync_shadows()
    check_all()
    for W in (blk.mlp.linear1[0].weight, blk.mlp.linear2[0].weight,
              ops.cat_rows([blk.attn.ProbSpare.query_projection.weight.detach(), blk.attn.ProbSpare.key_projection.weight.detach(),
                            blk.attn.ProbSpare.value_projection.weight.detach()])):
        W = W.detach()
        N, K = W.shape
        assert ops.bf16_copy_t(W) is not None, (N, K)
        dy = torch.randn(70000, N, device=dev).to(BF)
        dx = ops.gemm_dgrad(dy, W)
        ref = torch.empty_like(dx)
        _lib.call("dhz_linear_dgrad_bf16", dy.data_ptr(), N, ops.bf16_copy(W).data_ptr(), ref.data_ptr(), K, dy.shape[0], N, K, _s())
        d = (dx.float() - ref.float()).abs()
        assert (d <= EPS * ref.float().abs() + 1e-6).all(), d.max().item()       # at most the last rounding of a different summation order


@pytest.mark.parametrize("n", [1003, 4096, 7])
def test_adamw_shadow_equals_cast_of_updated_parameters(n):
    """dhz_adamw_step_shadow: the bf16 copy written in the optimizer's pass == a cast of the updated fp32 parameters (vector body
    and scalar tail), and the fp32 update == dhz_adamw_step's."""
    from dehaze_hip import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(n)
    p = torch.randn(n, generator=g).to(dev)
    gr = torch.randn(n, generator=g).to(dev)
    m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev)
    p2, m2, v2 = p.clone(), m.clone(), v.clone()
    p16 = torch.zeros(n, device=dev, dtype=BF)
    for step in (1, 2, 3):
        ops.adamw_step_(p, gr, m, v, 2e-4, 0.9, 0.999, 1e-8, 0.02, step, 1.0, p16=p16)
        ops.adamw_step_(p2, gr, m2, v2, 2e-4, 0.9, 0.999, 1e-8, 0.02, step, 1.0)
    assert torch.equal(p, p2) and torch.equal(m, m2) and torch.equal(v, v2)
    assert torch.equal(p16, p.to(BF))
