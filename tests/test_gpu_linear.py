"""-m gpu: dhz_linear_wgrad (skinny TN GEMM over tokens) and the token-major Linear wrapper vs torch."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("T,N,K", [(4096, 32, 32), (8192, 96, 32), (4096, 128, 32), (4096, 32, 128), (2048, 192, 64),
                                   (2048, 256, 64), (1024, 384, 128), (1024, 512, 128), (512, 1536, 512),
                                   (256, 2048, 512), (512, 512, 2048), (64, 64, 64),
                                   # long token slabs: the two-token-group variant of the kernel (>= 8 stages per workgroup),
                                   # the last two with an odd number of stages per workgroup (groups of unequal length)
                                   (65536, 128, 128), (65536, 96, 64), (32768, 384, 128), (8192, 1536, 512), (2336 * 32, 32, 64),
                                   # 16-wide forms (the embed_dim = 16 model: 16 -> 16 / 64, 64 -> 16, 32 -> 48); a ragged token count
                                   (4096, 16, 16), (32768, 64, 16), (32768, 16, 64), (8192, 48, 32), (128032, 16, 16), (96, 80, 48)])
def test_linear_tokens_grads(T, N, K):
    from dehaze_hip import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(T + N + K)
    x = torch.randn(T, K, generator=g)
    W = torch.randn(N, K, generator=g) * 0.1
    b = torch.randn(N, generator=g) * 0.1
    go = torch.randn(T, N, generator=g)
    xr, Wr, br = x.double().requires_grad_(), W.double().requires_grad_(), b.double().requires_grad_()
    (torch.nn.functional.linear(xr, Wr, br) * go.double()).sum().backward()
    xd = x.to(dev).requires_grad_()
    Wd = torch.nn.Parameter(W.to(dev))
    bd = torch.nn.Parameter(b.to(dev))
    y = ops.linear_tokens(xd, Wd, bd)
    (y * go.to(dev)).sum().backward()
    assert torch.allclose(y.detach().cpu().double(), torch.nn.functional.linear(xr, Wr, br).detach(), atol=1e-4, rtol=1e-4)
    scale = (T ** 0.5)
    assert torch.allclose(xd.grad.cpu().double(), xr.grad, atol=1e-4, rtol=1e-4)
    assert (Wd.grad.cpu().double() - Wr.grad).abs().max() < 2e-5 * scale
    assert (bd.grad.cpu().double() - br.grad).abs().max() < 2e-5 * scale
    # second backward accumulates in place
    y2 = ops.linear_tokens(xd, Wd, bd)
    (y2 * go.to(dev)).sum().backward()
    assert (Wd.grad.cpu().double() - 2 * Wr.grad).abs().max() < 4e-5 * scale


@pytest.mark.parametrize("T,C", [(2048, 64), (65536, 32), (32768, 128), (4096, 512), (16384, 16), (2048, 48)])
def test_packed_qkv_grads(T, C):
    """Q / K / V share their input: one dhz_linear_wgrad_multi launch fills the three separate .grad buffers."""
    from dehaze_hip import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    x = torch.randn(T, C, generator=g)
    Ws = [torch.randn(C, C, generator=g) * 0.1 for _ in range(3)]
    bs = [torch.randn(C, generator=g) * 0.1 for _ in range(3)]
    go = torch.randn(T, 3 * C, generator=g)
    xr = x.double().requires_grad_()
    Wr = [w.double().requires_grad_() for w in Ws]
    br = [b.double().requires_grad_() for b in bs]
    (torch.nn.functional.linear(xr, torch.cat(Wr), torch.cat(br)) * go.double()).sum().backward()
    Wd = [torch.nn.Parameter(w.to(dev)) for w in Ws]
    bd = [torch.nn.Parameter(b.to(dev)) for b in bs]
    xd = x.to(dev).requires_grad_()
    y = ops.linear_tokens(xd, Wd[0], bd[0], Wd[1], bd[1], Wd[2], bd[2])
    (y * go.to(dev)).sum().backward()
    tol = 2e-5 * T ** 0.5 + 1e-4
    for i in range(3):
        assert (Wd[i].grad.cpu().double() - Wr[i].grad).abs().max() < tol
        assert (bd[i].grad.cpu().double() - br[i].grad).abs().max() < tol
    assert torch.allclose(xd.grad.cpu().double(), xr.grad, atol=1e-4, rtol=1e-4)


def test_wgrad_row_scale_c_abi():
    """dhz_linear_wgrad_rs: rows of dy scaled per image on the way in == the plain kernel on a pre-scaled dy."""
    from dehaze_hip import _lib
    dev = torch.device("cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    B, L, N, K = 4, 256, 96, 64
    g = torch.Generator().manual_seed(11)
    dy = torch.randn(B * L, N, generator=g).to(dev); x = torch.randn(B * L, K, generator=g).to(dev)
    sc = torch.tensor([1.25, 0.0, 1.0, 1.0526316], device=dev)
    dw = torch.zeros(N, K, device=dev); db = torch.zeros(N, device=dev)
    _lib.call("dhz_linear_wgrad_rs", dy.data_ptr(), N, x.data_ptr(), K, B * L, N, K, dw.data_ptr(), db.data_ptr(), sc.data_ptr(), L, s)
    dys = (dy.view(B, L, N) * sc.view(B, 1, 1)).reshape(B * L, N)
    ref_w = dys.double().t() @ x.double(); ref_b = dys.double().sum(0)
    assert (dw.double() - ref_w).abs().max().item() < 1e-3 and (db.double() - ref_b).abs().max().item() < 1e-3
    lib = _lib.load()
    assert lib.dhz_linear_wgrad_rs(dy.data_ptr(), N, x.data_ptr(), K, B * L, N, K, dw.data_ptr(), db.data_ptr(), sc.data_ptr(), 48, s) == -22


def test_wgrad_multi_c_abi():
    """dhz_linear_wgrad_multi through the C-ABI: 4 parameters, no bias gradients, strided dy; and its argument checks."""
    import ctypes
    from dehaze_hip import _lib
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(9)
    T, n, N, K = 8192, 4, 96, 64
    dy = torch.randn(T, n * N + 32, generator=g).to(dev)          # 32 unused trailing columns: ldy > n*N
    x = torch.randn(T, K, generator=g).to(dev)
    dws = [torch.zeros(N, K, device=dev) for _ in range(n)]
    arr = (ctypes.c_void_p * n)(*[w.data_ptr() for w in dws])
    s = torch.cuda.current_stream().cuda_stream
    _lib.call("dhz_linear_wgrad_multi", dy.data_ptr(), dy.stride(0), x.data_ptr(), K, T, n, N, K,
              ctypes.cast(arr, ctypes.c_void_p), None, s)
    ref = dy[:, :n * N].double().t() @ x.double()
    for i in range(n):
        assert (dws[i].double() - ref[i * N:(i + 1) * N]).abs().max() < 2e-5 * T ** 0.5
    lib = _lib.load()
    assert lib.dhz_linear_wgrad_multi(dy.data_ptr(), dy.stride(0), x.data_ptr(), K, T, 5, N, K,
                                      ctypes.cast(arr, ctypes.c_void_p), None, s) == -22
    assert b"nmat" in lib.dhz_last_error()


@pytest.mark.parametrize("T,N,K", [(4096, 96, 32), (4096, 32, 32), (8192, 128, 32), (8192, 32, 128), (2048, 192, 64),
                                   (2048, 256, 64), (1000, 384, 128), (1024, 512, 128), (512, 1536, 512), (64, 2048, 512),
                                   (300, 512, 2048), (64, 64, 64), (1, 32, 32), (131072, 64, 256), (70000, 256, 64)])
def test_linear_gemm_c_abi(T, N, K):
    """dhz_linear_fwd / dhz_linear_dgrad through the raw C-ABI vs an fp64 matmul: every tile shape of the dispatch
    (32*WN features, 64 / 128 tokens), ragged token counts, a packed input with row stride > K and an output written into
    the middle of a wider buffer."""
    from dehaze_hip import _lib
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(T * 7 + N + K)
    xw = torch.randn(T, K + 32, generator=g).to(dev)               # x = the first K columns of a wider buffer
    W = (torch.randn(N, K, generator=g) * 0.1).to(dev)
    b = (torch.randn(N, generator=g) * 0.1).to(dev)
    s = torch.cuda.current_stream().cuda_stream
    yw = torch.full((T, N + 64), 7.0, device=dev)                  # y = columns 32 .. 32+N
    _lib.call("dhz_linear_fwd", xw.data_ptr(), xw.stride(0), W.data_ptr(), b.data_ptr(), yw.data_ptr() + 4 * 32, yw.stride(0),
              T, N, K, s)
    ref = xw[:, :K].double() @ W.double().t() + b.double()
    tol = 2e-6 * K ** 0.5 + 1e-5
    assert (yw[:, 32:32 + N].double() - ref).abs().max() < tol * max(1.0, ref.abs().max().item())
    assert (yw[:, :32] == 7.0).all() and (yw[:, 32 + N:] == 7.0).all()          # nothing outside the output columns
    y0 = torch.empty(T, N, device=dev)
    _lib.call("dhz_linear_fwd", xw.data_ptr(), xw.stride(0), W.data_ptr(), None, y0.data_ptr(), N, T, N, K, s)
    assert (y0.double() - (ref - b.double())).abs().max() < tol * max(1.0, ref.abs().max().item())
    dy = torch.randn(T, N, generator=g).to(dev)
    dx = torch.empty(T, K, device=dev)
    _lib.call("dhz_linear_dgrad", dy.data_ptr(), N, W.data_ptr(), dx.data_ptr(), K, T, N, K, s)
    refd = dy.double() @ W.double()
    assert (dx.double() - refd).abs().max() < (2e-6 * N ** 0.5 + 1e-5) * max(1.0, refd.abs().max().item())
    lib = _lib.load()
    assert lib.dhz_linear_fwd(xw.data_ptr(), xw.stride(0), W.data_ptr(), None, y0.data_ptr(), N, T, N + 1, K, s) == -22
    assert b"multiples of 16" in lib.dhz_last_error()          # (round 5: the 16-wide forms of the embed_dim = 16 model widened the contract from 32)
