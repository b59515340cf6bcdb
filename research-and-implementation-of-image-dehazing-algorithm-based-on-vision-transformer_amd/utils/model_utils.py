"""Model factory + checkpoint I/O (utils/model_utils.py:22-105).  Checkpoints are
{'epoch', 'state_dict', 'optimizer'} with keys possibly prefixed 'module.' (DataParallel)."""
import os
from collections import OrderedDict

import torch


def freeze(model):
    for p in model.parameters():
        p.requires_grad = False


def unfreeze(model):
    for p in model.parameters():
        p.requires_grad = True


def is_frozen(model):
    return not all(p.requires_grad for p in model.parameters())


def save_checkpoint(model_dir, state, session):
    torch.save(state, os.path.join(model_dir, "model_epoch_{}_{}.pth".format(state['epoch'], session)))


def _strip_module(state_dict):
    out = OrderedDict()
    for k, v in state_dict.items():
        out[k[7:] if k.startswith('module.') else k] = v
    return out


def _target(model):
    return model.module if hasattr(model, "module") and not hasattr(model, "input_proj") else model


def load_checkpoint(model, weights, map_location=None):
    checkpoint = torch.load(weights, map_location=map_location)
    print('load weight path:' + weights)
    _target(model).load_state_dict(_strip_module(checkpoint["state_dict"]))


def load_checkpoint_CPU(model, weights):
    load_checkpoint(model, weights, map_location=torch.device('cpu'))


def load_checkpoint_multigpu(model, weights):
    load_checkpoint(model, weights)


def load_start_epoch(weights):
    return torch.load(weights, map_location="cpu")["epoch"]


def load_optim(optimizer, weights):
    checkpoint = torch.load(weights, map_location="cpu")
    optimizer.load_state_dict(checkpoint['optimizer'])
    lr = None
    for p in optimizer.param_groups:
        lr = p['lr']
    return lr


def rng_state_dict():
    """State of every generator a training step draws from (python, numpy, torch CPU - the sampled-key tables and the
    epoch permutation - and the current HIP device - DropPath).  The reference's checkpoints hold none of this (TR:296-333:
    epoch, state_dict, optimizer), so a resumed run there re-draws from fresh seeds; My_train.py stores it under the extra
    key 'rng_state' (ignored by the reference's loaders) and restores it on --resume when present.  Plain tensors only -
    the Mersenne-Twister words as int64, the two cached-gaussian fields as float64 - so the file loads under torch.load's
    default weights_only=True and decoding it executes nothing (no pickle on either side)."""
    import random
    import numpy as np
    ver, words, gauss = random.getstate()                      # (3, 625 ints, None | float)
    name, key, pos, has_gauss, cached = np.random.get_state()   # ('MT19937', uint32[624], int, int, float)
    assert ver == 3 and name == "MT19937"
    st = {"python_mt": torch.tensor(words, dtype=torch.int64),
          "python_gauss": torch.tensor([0.0 if gauss is None else 1.0, 0.0 if gauss is None else float(gauss)],
                                       dtype=torch.float64),
          "numpy_mt": torch.tensor(np.asarray(key, dtype=np.int64)),
          "numpy_aux": torch.tensor([float(pos), float(has_gauss), float(cached)], dtype=torch.float64),
          "torch": torch.get_rng_state()}
    if torch.cuda.is_available():
        st["cuda"] = torch.cuda.get_rng_state()
    return st


def set_rng_state(st):
    """Inverse of rng_state_dict(); validates types and sizes and rebuilds the state tuples by hand."""
    import random
    import numpy as np

    def vec(key, n, dtype):
        t = st[key]
        if not isinstance(t, torch.Tensor) or t.dtype != dtype or t.dim() != 1 or t.numel() != n:
            raise ValueError(f"rng_state['{key}']: expected a {dtype} tensor of {n} elements")
        return t.cpu()

    words = vec("python_mt", 625, torch.int64).tolist()
    pg = vec("python_gauss", 2, torch.float64).tolist()
    key = vec("numpy_mt", 624, torch.int64).numpy()
    aux = vec("numpy_aux", 3, torch.float64).tolist()
    if min(words) < 0 or max(words[:624]) > 0xFFFFFFFF or not 0 <= words[624] <= 624 or key.min() < 0 \
            or key.max() > 0xFFFFFFFF or not 0 <= int(aux[0]) <= 624:
        raise ValueError("rng_state: Mersenne-Twister words out of range")
    random.setstate((3, tuple(int(w) for w in words), pg[1] if pg[0] else None))
    np.random.set_state(("MT19937", key.astype(np.uint32), int(aux[0]), int(aux[1]), float(aux[2])))
    torch.set_rng_state(st["torch"].cpu())
    if "cuda" in st and torch.cuda.is_available():
        torch.cuda.set_rng_state(st["cuda"].cpu())


def load_rng_state(weights, rank=0):
    """Restore the generators from a checkpoint written by this My_train.py; False for a reference checkpoint (no such key).
    The checkpoint holds one state per rank of the run that wrote it (a list indexed by rank; a bare dict = rank 0 of a
    single-process run).  A rank the file has no entry for (resumed on more GPUs than it was written with) returns False and
    keeps the rank-specific seeds My_train.py gave it - every rank restoring rank 0's state would make all of them draw the
    same crops, MixUp lambdas, DropPath masks and sampled keys."""
    checkpoint = torch.load(weights, map_location="cpu")
    if "rng_state" not in checkpoint:
        return False
    st = checkpoint["rng_state"]
    if isinstance(st, dict):
        st = [st]
    if rank >= len(st) or st[rank] is None:
        return False
    if not isinstance(st[rank], dict) or "python_mt" not in st[rank]:
        # a checkpoint of an earlier commit of this repository ('host': pickled bytes) or a foreign format: never unpickled;
        # the run continues on fresh seeds, as it does for a reference checkpoint
        import warnings
        warnings.warn(f"{weights}: 'rng_state' is not in this version's format (legacy or foreign file): generator states not restored")
        return False
    set_rng_state(st[rank])
    return True


def get_arch(opt):
    from My_model_1 import UNet, Uformer          # the ProbSparse model, as model_utils.py:81
    arch = opt.arch
    print('You choose ' + arch + '...')
    if arch == 'UNet':
        return UNet(dim=opt.embed_dim)
    if arch == 'Uformer':
        return Uformer(img_size=opt.train_ps, embed_dim=opt.embed_dim, win_size=opt.win_size,
                       token_projection=opt.token_projection, token_mlp=opt.token_mlp)
    if arch == 'Uformer16':
        # model_utils.py:96-98: embed_dim 16 -> head_dim 16 in every block (heads 1, 2, 4, 8, 16, 16, 8, 4, 2).  The window-attention
        # kernels carry head_dim-16 instances, the C = 16 stage's Linears (16 -> 48 / 16 / 64, 64 -> 16) run the 16-wide GEMM /
        # weight-gradient forms; its two thin convolutions (Downsample 16 -> 32, OutputProj 32 -> 3) run the hand-written kernels on zero-padded channels
        return Uformer(img_size=opt.train_ps, embed_dim=16, win_size=8, token_projection='linear', token_mlp='leff')
    if arch == 'Uformer32':
        return Uformer(img_size=opt.train_ps, embed_dim=32, win_size=8, token_projection='linear', token_mlp='leff')
    raise Exception("Arch error!")
