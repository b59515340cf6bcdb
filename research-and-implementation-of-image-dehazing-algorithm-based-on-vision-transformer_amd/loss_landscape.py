"""Loss-landscape caller of the hot path: the model forward + both losses evaluated on a 2-D slice of weight
space spanned by two filter-normalised random directions.

Follows how-do-vits-work-transformer/ops/loss_landscapes.py:11-124 (bases, grid), ops/My_tests.py:26-106 (the loss of one
grid point), ops/norm.py:4-21 (the l1 / l2 columns) and My_losslandscape.py:190-215 (the call and the CSV), with the same
names and argument meaning.  What is organised differently, because the model lives in HBM:

  * the reference deep-copies the state_dict and calls load_state_dict() for every grid point; here the unperturbed weights and
    the two directions are three lists of device tensors and a grid point is  w0 + x*b0 + y*b1  written in place with three
    multi-tensor calls (no host copies, no allocation proportional to the grid);
  * only floating-point state entries are perturbed.  The reference draws directions for integer buffers as well and relies on
    the keyword filter (`kws=["pos_embed", "relative_position"]`, My_losslandscape.py:199) to zero them;
  * fp32 instead of CUDA autocast (My_tests.py:72).

Quirks of the reference that are kept, because the numbers in its CSV depend on them (My_tests.py:53-86): the loader yields
(clean, hazy, ...) and the FIRST tensor is what the model restores - `restored = model(xs)` with xs = clean - the
Charbonnier target is the second tensor, and the contrastive loss is called as (anchor=restored, positive=xs,
negative=ys).  `evaluate_loss` takes `restore_first=False` to get the training step's roles instead (TR:212-247).
"""
import csv
import re

import numpy as np
import torch


# ----------------------------------------------------------------------------- directions (loss_landscapes.py:11-72)
def normalize_filter(bs, ws):
    """Scale every direction tensor so that each 'filter' has the norm of the matching weight filter: norms are taken over
    dim 0 (keepdim), as the reference does (loss_landscapes.py:16-19), eps 1e-7 in the denominator."""
    out = {}
    for k, b in bs.items():
        b = b.float()
        w = ws[k].float()
        out[k] = torch.norm(w, dim=0, keepdim=True) / (torch.norm(b, dim=0, keepdim=True) + 1e-7) * b
    return out


def ignore_bn(ws):
    """Directions of vectors and scalars (biases, LayerNorm affine) are zero (loss_landscapes.py:24-31)."""
    return {k: (torch.zeros_like(v) if v.dim() < 2 else v) for k, v in ws.items()}


def ignore_kw(ws, kws=None):
    """Directions of entries whose name matches any regular expression in `kws` are zero (loss_landscapes.py:38-47)."""
    kws = kws or []
    return {k: (torch.zeros_like(v) if any(re.search(kw, k) for kw in kws) else v) for k, v in ws.items()}


def rand_basis(ws, generator=None):
    return {k: torch.randn(v.shape, device=v.device, dtype=torch.float32, generator=generator) for k, v in ws.items()}


def _float_state(model):
    return {k: v for k, v in model.state_dict().items() if v.is_floating_point()}


def create_bases(model, kws=None, generator=None):
    """Two random directions over the floating-point state of `model`: N(0,1) draws, filter-normalised against the
    weights, vectors/scalars and keyword matches zeroed (loss_landscapes.py:54-72).  `generator`: optional generator on the
    model's device for reproducible directions."""
    ws0 = _float_state(model)
    bases = [rand_basis(ws0, generator) for _ in range(2)]
    bases = [normalize_filter(bs, ws0) for bs in bases]
    bases = [ignore_bn(bs) for bs in bases]
    return [ignore_kw(bs, kws) for bs in bases]


# ----------------------------------------------------------------------------- one grid point (My_tests.py:26-106)
def l1(model):
    return sum(torch.norm(p.detach(), 1) for p in model.parameters())


def l2(model):
    return sum(torch.norm(p.detach()) for p in model.parameters())


@torch.no_grad()
def evaluate_loss(model, dataset, criterion, transform=None, w_char=1.0, w_cr=1.0, restore_first=True):
    """Mean over the batches of `dataset` of  w_char*Charbonnier + w_cr*Contrast  in eval mode (My_tests.py:44-92; the
    reference averages per-batch losses with equal weight).  `criterion` = (CharbonnierLoss, ContrastLoss or None);
    batches are tuples whose first two entries are image tensors."""
    model.eval()
    dev = next(model.parameters()).device
    total, n = torch.zeros((), device=dev), 0
    for batch in dataset:
        xs, ys = batch[0].to(dev, non_blocking=True), batch[1].to(dev, non_blocking=True)
        if transform is not None:
            xs, ys = transform(xs, ys)
        # restore_first: the reference's landscape script (module docstring); else the training step's roles,
        # batch = (target, hazy input)
        inp, tgt = (xs, ys) if restore_first else (ys, xs)
        restored = torch.clamp(model(inp), 0, 1)
        loss = w_char * criterion[0](restored, tgt) if w_char > 0 else 0
        if w_cr > 0 and criterion[1] is not None:
            loss = loss + w_cr * criterion[1](restored, xs, ys)[0]
        total += loss
        n += 1
    return (total / max(n, 1)).item()


# ----------------------------------------------------------------------------- the grid (loss_landscapes.py:75-124)
def get_loss_landscape(model, dataset, criterion, transform=None, bases=None, kws=None, x_min=-1.0, x_max=1.0, n_x=11,
                       y_min=-1.0, y_max=1.0, n_y=11, w_char=1.0, w_cr=1.0, restore_first=True, verbose=False,
                       generator=None):
    """{(x, y): (l1, l2, loss)} over the n_x * n_y grid, in the reference's order (np.meshgrid, x fastest).  The model's
    weights are restored before returning (also when a grid point raises)."""
    state = _float_state(model)
    keys = list(state)
    live = [state[k] for k in keys]
    w0 = [t.detach().clone() for t in live]
    bases = create_bases(model, kws, generator) if bases is None else bases
    b0 = [bases[0][k].to(t.dtype) for k, t in zip(keys, live)]
    b1 = [bases[1][k].to(t.dtype) for k, t in zip(keys, live)]
    xs = np.linspace(x_min, x_max, n_x)
    ys = np.linspace(y_min, y_max, n_y)
    ratio_grid = np.stack(np.meshgrid(xs, ys), axis=0).transpose((1, 2, 0)).reshape(-1, 2)
    was_training = model.training
    metrics_grid = {}
    try:
        for rx, ry in ratio_grid:
            with torch.no_grad():
                torch._foreach_copy_(live, w0)
                torch._foreach_add_(live, b0, alpha=float(rx))
                torch._foreach_add_(live, b1, alpha=float(ry))
            loss = evaluate_loss(model, dataset, criterion, transform, w_char, w_cr, restore_first)
            metrics_grid[(float(rx), float(ry))] = (float(l1(model)), float(l2(model)), loss)
            if verbose:
                print("Grid: [%g %g], loss_value: %.4f" % (rx, ry, loss), flush=True)
    finally:
        with torch.no_grad():
            torch._foreach_copy_(live, w0)
        model.train(was_training)
    return metrics_grid


def save_metrics(path, metrics_grid):
    """CSV rows  x, y, l1, l2, loss_value  (My_tests.py:208-229; read back with names=["x","y","l1","l2","loss_value"],
    My_losslandscape.py:226)."""
    with open(path, "w", newline="") as f:
        wr = csv.writer(f)
        for grid, metrics in metrics_grid.items():
            wr.writerow([*grid, *metrics])


def load_surface(path):
    """(xs, ys, zs) square arrays of a saved landscape, zs shifted to min 0 (My_losslandscape.py:226-240)."""
    rows = np.loadtxt(path, delimiter=",", ndmin=2)
    p = int(round(np.sqrt(len(rows))))
    xs, ys, zs = rows[:, 0].reshape(p, p), rows[:, 1].reshape(p, p), rows[:, 4].reshape(p, p)
    zs = zs - zs[np.isfinite(zs)].min()
    return xs, ys, zs
