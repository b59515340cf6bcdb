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
