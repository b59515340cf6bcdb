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
    key 'rng_state' (ignored by the reference's loaders) and restores it on --resume when present.  Tensors only, so that
    the file still loads with torch.load's default weights_only=True: the python / numpy states travel as a pickled byte
    tensor that only set_rng_state() decodes."""
    import pickle
    import random
    import numpy as np
    blob = pickle.dumps({"python": random.getstate(), "numpy": np.random.get_state()})
    st = {"host": torch.frombuffer(bytearray(blob), dtype=torch.uint8).clone(), "torch": torch.get_rng_state()}
    if torch.cuda.is_available():
        st["cuda"] = torch.cuda.get_rng_state()
    return st


def set_rng_state(st):
    import pickle
    import random
    import numpy as np
    host = pickle.loads(st["host"].cpu().numpy().tobytes())
    random.setstate(host["python"])
    np.random.set_state(host["numpy"])
    torch.set_rng_state(st["torch"].cpu())
    if "cuda" in st and torch.cuda.is_available():
        torch.cuda.set_rng_state(st["cuda"].cpu())


def load_rng_state(weights):
    """Restore the generators from a checkpoint written by this My_train.py; False for a reference checkpoint (no such key)."""
    checkpoint = torch.load(weights, map_location="cpu")
    if "rng_state" not in checkpoint:
        return False
    set_rng_state(checkpoint["rng_state"])
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
        return Uformer(img_size=opt.train_ps, embed_dim=16, win_size=8, token_projection='linear', token_mlp='leff')
    if arch == 'Uformer32':
        return Uformer(img_size=opt.train_ps, embed_dim=32, win_size=8, token_projection='linear', token_mlp='leff')
    raise Exception("Arch error!")
