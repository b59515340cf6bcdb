"""N>1 path on CPU: world_size-2 gloo run of the bucketed gradient reducer (dehaze_hip.train.GradReducer) -
the same code that drives RCCL over xGMI on the GPUs.  Checks that the reduced gradients equal the
full-batch gradients, that bucket bookkeeping survives parameters without gradients (the reference's dead
attn.qkv/attn.proj tensors) and in-place (hook-bypassing) gradient producers."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build():
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(16, 64), torch.nn.GELU(), torch.nn.Linear(64, 64), torch.nn.GELU(),
                              torch.nn.Linear(64, 8))
    dead = torch.nn.Linear(8, 8)          # never used in forward: gets no gradient
    return net, dead


def _worker(rank, world, port, q, overlap=True):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dehaze_hip import ops
    from dehaze_hip.train import GradReducer
    net, dead = _build()
    params = list(net.parameters())
    red = GradReducer(params=params, bucket_mb=0.002, overlap=overlap)      # tiny buckets -> several collectives
    assert len(red.buckets) >= 3
    g = torch.Generator().manual_seed(5)
    x = torch.randn(8, 16, generator=g)
    y = torch.randn(8, 8, generator=g)
    xs, ys = x[rank * 4:(rank + 1) * 4], y[rank * 4:(rank + 1) * 4]
    for it in range(2):                                     # two steps: bookkeeping must reset
        red.flat.zero_()
        loss = ((net(xs) - ys) ** 2).sum()
        loss.backward()
        assert overlap or not red._handles                  # --no-overlap: nothing is launched before wait()
        red.wait()
        red.average_()
    grads = [p.grad.clone() for p in params]
    # in-place producer path (wgrad kernels bypass autograd hooks and call ops.GRAD_READY themselves)
    assert ops.GRAD_READY is not None
    red.flat.zero_()
    for p in params:
        p.grad.add_(float(rank + 1))
        ops.GRAD_READY(p)
    red.wait()
    ok_inplace = bool(torch.allclose(red.flat, torch.full_like(red.flat, 3.0)))
    # the product's shape of that path: an autograd node that accumulates a leaf's gradient in place, announces it and
    # returns None for it - PyTorch still runs the leaf's post-accumulate hook, so the parameter is announced twice and
    # must be counted once (a double count fires the bucket before its other parameters are done)
    class InPlace(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, w, b):
            ctx.save_for_backward(x)
            ctx.w = w
            return x @ w.t() + b

        @staticmethod
        def backward(ctx, g):
            (x,) = ctx.saved_tensors
            ctx.w.grad.add_(g.t() @ x)
            ops.GRAD_READY(ctx.w)
            return g @ ctx.w, None, g.sum(0)
    red.wait()
    red.flat.zero_()
    red.calls = {}
    h = InPlace.apply(xs, params[0], params[1])                     # params[0]: first Linear's weight, announced in place + by hook
    for layer in list(net)[1:]:
        h = layer(h)
    ((h - ys) ** 2).sum().backward()
    red.wait()
    red.average_()
    ok_inplace = ok_inplace and red.calls[id(params[0])] >= 1 and all(
        bool(torch.allclose(p.grad, g_, atol=1e-5, rtol=1e-5)) for p, g_ in zip(params[1:], grads[1:]))
    w_grad_inplace = params[0].grad.clone()
    grads = grads + [w_grad_inplace]
    # the `--optimizer adam` path of My_train.py: a torch optimizer's zero_grad() drops the .grad views (set_to_none); the
    # reducer's zero_grad() re-binds them to its flat buffer, and one averaged step leaves the replicas identical
    opt = torch.optim.Adam(params, lr=1e-2)
    opt.zero_grad()
    assert all(p.grad is None for p in params)
    red.zero_grad()
    ok_inplace = ok_inplace and all(p.grad is not None and red.flat.data_ptr() <= p.grad.data_ptr() < red.flat.data_ptr() + 4 * red.flat.numel()
                                    for p in params)
    ((net(xs) - ys) ** 2).sum().backward()
    red.wait()
    red.average_()
    opt.step()
    sums = torch.stack([p.detach().double().sum() for p in params])
    both = [torch.zeros_like(sums) for _ in range(world)]
    dist.all_gather(both, sums)
    ok_inplace = ok_inplace and bool(torch.equal(both[0], both[1]))
    q.put((rank, [g.numpy().copy() for g in grads], ok_inplace))     # by value: torch tensors would travel as shm handles
                                                                     # that die with this process
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("overlap", [True, False])
def test_grad_reducer_world2_gloo(overlap):
    """overlap=False is bench.py --no-overlap: every bucket's collective is launched by wait(), after backward"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, overlap)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    net, _ = _build()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(8, 16, generator=g)
    y = torch.randn(8, 8, generator=g)
    (((net(x) - y) ** 2).sum() / 2).backward()              # mean over the 2 ranks of per-rank sums
    for (r, grads, ok_inplace) in res:
        assert ok_inplace
        for gp, p in zip(grads, net.parameters()):
            assert torch.allclose(torch.from_numpy(gp), p.grad, atol=1e-5, rtol=1e-5)
        assert torch.allclose(torch.from_numpy(grads[-1]), list(net.parameters())[0].grad, atol=1e-5, rtol=1e-5)


def test_flat_adamw_layout_and_state_dict_cpu():
    """Flat layout, live/dead split and the torch-compatible positional state_dict (no kernel launch)."""
    import My_model_1 as M1
    from dehaze_hip.train import FlatAdamW, GradReducer
    torch.manual_seed(0)
    m = M1.Uformer(img_size=128, embed_dim=32, win_size=8, token_projection='linear', token_mlp='leff')
    opt = FlatAdamW(m)
    opt.zero_grad()
    # 20,628,317 live elements; every parameter starts on a 32-byte boundary (<= 7 pad floats each)
    nlive = len(opt.param_slices())
    assert 20628317 <= opt.flat_grad.numel() <= 20628317 + 7 * nlive and opt.flat_grad.numel() % 8 == 0
    assert all(off % 8 == 0 for _, off, _ in opt.param_slices())
    dead = [p for n, p in m.named_parameters() if "attn.qkv." in n or "attn.proj." in n]
    assert len(dead) == 108 and all(p.grad is None for p in dead)
    live = [p for _, p in m.live_parameters()]
    assert all(p.grad is not None and p.grad.data_ptr() >= opt.flat_grad.data_ptr() for p in live)
    # buckets cover the flat gradient exactly once, in order
    red = GradReducer(opt, bucket_mb=25)
    last = opt.param_slices()[-1]
    assert red.buckets[0][0] == 0 and red.buckets[-1][1] == last[1] + last[2]
    assert all(a[1] == b[0] for a, b in zip(red.buckets, red.buckets[1:])) and 3 <= len(red.buckets) <= 5
    # the exchange plan of the 8-GPU run (config 3), checked without the hardware: buckets in reverse registration order - the last
    # decoder stage (whose gradients arrive first) in bucket 0, the input projection (last to arrive) in the last one -, the
    # 108 dead tensors in none, ~25 MB each, and a single-ring xGMI time far below the step time
    plan = red.plan(world=8)
    names = {id(p): n for n, p in m.named_parameters()}
    first = [names[id(p)] for p, off, k in opt.param_slices() if off + k <= plan["buckets"][0]["hi"]]
    lastb = [names[id(p)] for p, off, k in opt.param_slices() if off >= plan["buckets"][-1]["lo"]]
    assert any(n.startswith("decoderlayer_3.") for n in first) and any(n.startswith("input_proj.") for n in lastb)
    assert not any(id(p) in red.bucket_of for p in dead)
    assert sum(b["params"] for b in plan["buckets"]) == len(live)
    assert all(b["bytes"] >= 25 * 2 ** 20 for b in plan["buckets"][:-1]) and plan["buckets"][-1]["bytes"] > 0
    assert abs(plan["payload_bytes"] - 4 * 20628317) < 4 * 8 * len(live)
    assert 0.8 < plan["ring_time_ms"] < 1.1                      # 2 * 7/8 * 82.5 MB / 153 GB/s = 0.94 ms
    assert red.plan(world=1)["ring_time_ms"] == 0.0
    sd = opt.state_dict()
    assert sd["param_groups"][0]["params"] == list(range(len(list(m.parameters()))))
    ref = torch.optim.AdamW(m.parameters(), lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.02)
    for k in ("lr", "betas", "eps", "weight_decay"):
        assert sd["param_groups"][0][k] == ref.state_dict()["param_groups"][0][k]
