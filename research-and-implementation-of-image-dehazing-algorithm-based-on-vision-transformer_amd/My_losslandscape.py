"""Drop-in for how-do-vits-work-transformer/My_losslandscape.py:100-215 on MI355X: build the model (options.py flags), load
`--pretrain_weights` if given, evaluate the loss on an n x n grid around the weights and write the CSV the reference plots.

  python My_losslandscape.py --arch Uformer --embed_dim 32 --train_dir <dir with gt/ hazy/> --n_grid 21 --scale 1.0
  python My_losslandscape.py --synthetic 64 --n_grid 5            # synthetic haze pairs, no files needed
"""
import argparse
import os
import sys

dir_name = os.path.dirname(os.path.abspath(__file__))
if dir_name not in sys.path:
    sys.path.insert(0, dir_name)

import torch  # noqa: E402

import loss_landscape as lls  # noqa: E402
import options  # noqa: E402
import utils  # noqa: E402
from dehaze_hip.train import synthetic_batch  # noqa: E402
from losses import CharbonnierLoss  # noqa: E402
from My_CR import ContrastLoss  # noqa: E402


def main(argv=None):
    parser = options.Options().init(argparse.ArgumentParser(description='loss landscape of the dehazing model'))
    parser.add_argument('--synthetic', type=int, default=0, help='evaluate on N synthetic pairs instead of --train_dir')
    parser.add_argument('--n_grid', type=int, default=21)
    parser.add_argument('--scale', type=float, default=1.0)
    parser.add_argument('--mixup', action='store_true', help='apply MixUp to every batch as the reference script does')
    parser.add_argument('--out', type=str, default='')
    parser.add_argument('--seed', type=int, default=1234)
    opt = parser.parse_args(argv)
    dev = torch.device("cuda", 0)
    torch.manual_seed(opt.seed)

    model = utils.get_arch(opt).to(dev)
    if opt.pretrain_weights and os.path.exists(opt.pretrain_weights):
        utils.load_checkpoint(model, opt.pretrain_weights)
    criterion = [CharbonnierLoss().to(dev), ContrastLoss(ablation=opt.is_ab).to(dev) if opt.w_loss_vgg7 > 0 else None]

    if opt.synthetic:
        bs = opt.batch_size
        batches = []
        for i in range(0, opt.synthetic, bs):
            gt, hazy = synthetic_batch(min(bs, opt.synthetic - i), opt.train_ps, seed=opt.seed + i, device=dev)
            batches.append((gt, hazy))
    else:
        from dataset import PatchStoreHBM
        store = PatchStoreHBM.from_dir(opt.train_dir, dev)
        import random
        import numpy as np
        random.seed(opt.seed); np.random.seed(opt.seed)     # crop origin / augmentation draws (dataset.draw_crop_aug)
        batches = [store.batch(range(i, min(i + opt.batch_size, len(store))), opt.train_ps)
                   for i in range(0, len(store), opt.batch_size)]

    transform = utils.MixUp_AUG().aug if opt.mixup else None
    gen = torch.Generator(device=dev).manual_seed(opt.seed)
    grid = lls.get_loss_landscape(model, batches, criterion, transform=transform, kws=["pos_embed", "relative_position"],
                                  x_min=-opt.scale, x_max=opt.scale, n_x=opt.n_grid, y_min=-opt.scale, y_max=opt.scale,
                                  n_y=opt.n_grid, w_char=opt.w_loss_CharbonnierLoss, w_cr=opt.w_loss_vgg7, verbose=True,
                                  generator=gen)
    out = opt.out or os.path.join(opt.save_dir, "%s_x%s_losslandscape.csv" % (opt.arch, int(1 / opt.scale)))
    os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
    lls.save_metrics(out, grid)
    print("saved", out)
    return out


if __name__ == "__main__":
    main()
