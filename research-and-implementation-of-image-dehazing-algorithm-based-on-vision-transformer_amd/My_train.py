"""Drop-in for Uformer_ProbSparse/My_train.py (TR) on MI355X.

Same flags (options.py), seeds (TR:72-75), model factory (TR:78), AdamW hyper-parameters (TR:90-92),
warm-up + cosine schedule (TR:121-126), losses (TR:145-147), step body (TR:212-250), 4x/epoch validation
with best-PSNR checkpointing (TR:258-310) and per-epoch checkpoints (TR:330-333), log layout
log/<arch><env>/{models,results} (TR:61-69).

What is different by design:
  * one process per GPU (`python -m torch.distributed.run --nproc-per-node N My_train.py ...`) with a bucketed
    RCCL gradient all-reduce over xGMI instead of nn.DataParallel (TR:97); checkpoints are written by rank 0
    with the 'module.' key prefix so that they stay loadable by the reference scripts;
  * fp32 (the reference wraps the forward in CUDA fp16 autocast + GradScaler, TR:224,249 - BASELINE configs
    2/3 ask for fp32);
  * data lives in HBM: `--train_dir/--val_dir` directories of PNG pairs (<dir>/gt, <dir>/hazy; dataset.py) are decoded
    once into uint8 tensors on the device and every batch (random crop, 8 rotate/flip augmentations, /255) is one kernel
    launch (dataset.PatchStoreHBM, dhz_crop_augment_pair) instead of DataLoader workers + PCIe per step; a `.pt` file
    holding {'target': [N,3,H,W], 'input': [N,3,H,W]} float tensors in [0,1] is accepted too; `--synthetic N` trains
    on N synthetic haze pairs per epoch.
"""
import argparse
import datetime
import os
import random
import sys
import time

dir_name = os.path.dirname(os.path.abspath(__file__))
if dir_name not in sys.path:
    sys.path.insert(0, dir_name)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import options  # noqa: E402
import utils  # noqa: E402
from dehaze_hip.train import FlatAdamW, GradReducer, psnr, synthetic_batch, train_step  # noqa: E402
from losses import CharbonnierLoss  # noqa: E402
from My_CR import ContrastLoss  # noqa: E402
from warmup_scheduler import GradualWarmupScheduler  # noqa: E402


def load_pairs(path, device):
    blob = torch.load(path, map_location="cpu")
    return blob["target"].float().to(device), blob["input"].float().to(device)


def main():
    parser = options.Options().init(argparse.ArgumentParser(description='remove the haze'))
    parser.add_argument('--synthetic', type=int, default=0, help='train on N synthetic pairs per epoch (HBM resident)')
    parser.add_argument('--val_synthetic', type=int, default=8)
    parser.add_argument('--log_every', type=int, default=10, help='host sync cadence for the progress line')
    opt = parser.parse_args()
    torch.backends.cudnn.benchmark = True      # TR:35 - on ROCm: MIOpen measures its convolution algorithms at first use

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)
    is_main = rank == 0
    if is_main:
        print(opt)

    log_dir = os.path.join(dir_name, 'log', opt.arch + opt.env)
    result_dir, model_dir = os.path.join(log_dir, 'results'), os.path.join(log_dir, 'models')
    logname = os.path.join(log_dir, datetime.datetime.now().isoformat() + '.txt')
    if is_main:
        utils.mkdir(result_dir)
        utils.mkdir(model_dir)

    random.seed(1234)
    np.random.seed(1234)
    torch.manual_seed(1234)
    torch.cuda.manual_seed_all(1234)

    model = utils.get_arch(opt)
    if is_main:
        with open(logname, 'a') as f:
            f.write(str(opt) + '\n' + str(model) + '\n')

    if opt.optimizer.lower() == 'adamw':
        optimizer = FlatAdamW(model, lr=opt.lr_initial, betas=(0.9, 0.999), eps=1e-8, weight_decay=opt.weight_decay)
    elif opt.optimizer.lower() == 'adam':
        optimizer = torch.optim.Adam(model.parameters(), lr=opt.lr_initial, betas=(0.9, 0.999), eps=1e-8,
                                     weight_decay=opt.weight_decay)
    else:
        raise Exception("Error optimizer...")
    model.to(dev)

    start_epoch = 1
    scheduler = None
    resumed_rng = None
    if opt.resume:
        utils.load_checkpoint(model, opt.pretrain_weights, map_location=dev)
        start_epoch = utils.load_start_epoch(opt.pretrain_weights) + 1
        lr = utils.load_optim(optimizer, opt.pretrain_weights)
        for p in optimizer.param_groups:
            p['lr'] = lr
        scheduler = torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, opt.nepoch - start_epoch + 1, eta_min=1e-6)
        resumed_rng = opt.pretrain_weights
    elif opt.warmup:
        cosine = torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, opt.nepoch - opt.warmup_epochs, eta_min=1e-6)
        scheduler = GradualWarmupScheduler(optimizer, multiplier=1, total_epoch=opt.warmup_epochs, after_scheduler=cosine)
    else:
        scheduler = torch.optim.lr_scheduler.StepLR(optimizer, step_size=50, gamma=0.5)

    reducer = None
    if world > 1:
        if isinstance(optimizer, FlatAdamW):
            optimizer.zero_grad()
            reducer = GradReducer(optimizer)
        else:
            # `--optimizer adam`: the stand-alone reducer owns a flat gradient buffer, averages it after the all-reduce
            # (train_step) and leaves the update to torch.optim.Adam - the replicas stay identical like under DataParallel
            reducer = GradReducer(params=[p for _, p in model.live_parameters()] if hasattr(model, "live_parameters")
                                  else model.parameters())

    char = CharbonnierLoss()
    if opt.w_loss_vgg7 > 0 and not os.environ.get("DEHAZE_VGG19_WEIGHTS") and not opt.synthetic \
            and not os.environ.get("DEHAZE_ALLOW_RANDOM_VGG"):
        raise SystemExit("My_train: the contrastive loss (w_loss_vgg7 > 0) needs the ImageNet VGG19 checkpoint the reference "
                         "downloads (My_CR.py:59): set DEHAZE_VGG19_WEIGHTS=/path/to/vgg19-dcbb9e9d.pth (no network here), "
                         "or DEHAZE_ALLOW_RANDOM_VGG=1 to train against SEEDED RANDOM VGG features (not the reference's loss)")
    cr = ContrastLoss(ablation=opt.is_ab).to(dev) if opt.w_loss_vgg7 > 0 else None
    if world > 1:
        # identical replicas above (same seed -> same initial weights); from here on every rank draws its own permutations,
        # crops, augmentations, MixUp lambdas, DropPath masks and sampled keys
        random.seed(1234 + rank)
        np.random.seed(1234 + rank)
        torch.manual_seed(1234 + rank)
        torch.cuda.manual_seed_all(1234 + rank)

    # ---- data (HBM resident)
    store = None
    if opt.synthetic > 0:
        tgt_all, inp_all = synthetic_batch(opt.synthetic, opt.train_ps, seed=1234 + rank, device=dev)
        vt, vi = synthetic_batch(opt.val_synthetic, opt.train_ps, seed=4321, device=dev)
    elif os.path.isdir(opt.train_dir):
        from dataset import PatchStoreHBM, DataLoaderVal
        store = PatchStoreHBM.from_dir(opt.train_dir, dev, rank, world)       # this rank's share of the patch pairs
        val_set = DataLoaderVal(opt.val_dir)
        val_items = [val_set[i] for i in range(len(val_set))]
        vt = torch.stack([v[0] for v in val_items]).to(dev)
        vi = torch.stack([v[1] for v in val_items]).to(dev)
        tgt_all = inp_all = None
    else:
        tgt_all, inp_all = load_pairs(opt.train_dir, dev)
        vt, vi = load_pairs(opt.val_dir, dev)
        tgt_all, inp_all = tgt_all[rank::world], inp_all[rank::world]
    n_train = len(store) if store is not None else tgt_all.shape[0]
    n_epoch = n_train
    if world > 1:
        # rank::world shards differ by one item when N % world != 0: every rank must run the SAME number of steps (a rank
        # with an extra step would wait in its bucket all-reduce for ever), so an epoch uses the smallest shard's item count
        nmin = torch.tensor([n_train], device=dev)
        dist.all_reduce(nmin, op=dist.ReduceOp.MIN)
        n_epoch = int(nmin.item())
    steps_per_epoch = max(1, -(-n_epoch // opt.batch_size))      # drop_last=False (TR:160): the last batch may be partial
    eval_now = max(1, steps_per_epoch // 4)
    mixup = utils.MixUp_AUG()

    def evaluate():
        model.eval()
        vals = []
        with torch.no_grad():
            for s in range(0, vt.shape[0], opt.batch_size):
                restored = torch.clamp(model(vi[s:s + opt.batch_size]), 0, 1)
                vals += [psnr(restored[i], vt[s + i]) for i in range(restored.shape[0])]
        model.train()
        return sum(vals) / len(vals)

    # every rank restores ITS OWN streams (the checkpoint holds one state per rank of the writing run); a rank without an entry
    # keeps the rank-specific seeds from above
    if resumed_rng is not None and utils.load_rng_state(resumed_rng, rank) and is_main:
        print("==> generator states restored from the checkpoint (continues the interrupted run's random streams)")

    def all_rng_states():
        """One generator state per rank (collective: every rank calls it at the same points); rank 0 writes the list."""
        st = utils.rng_state_dict()
        if world == 1:
            return [st]
        out = [None] * world
        dist.all_gather_object(out, st)
        return out
    best_psnr, best_epoch, best_iter = 0, 0, 0
    model.train()
    for epoch in range(start_epoch, opt.nepoch + 1):
        t0 = time.time()
        perm = torch.randperm(n_train)
        epoch_loss = torch.zeros((), device=dev)
        for i in range(steps_per_epoch):
            sel = perm[:n_epoch][i * opt.batch_size:(i + 1) * opt.batch_size]
            if store is not None:
                target, input_ = store.batch(sel.tolist(), opt.train_ps)
            else:
                target, input_ = tgt_all[sel.to(dev)], inp_all[sel.to(dev)]
            if epoch > 5:
                target, input_ = mixup.aug(target, input_)
            loss, loss_rec, loss_cr = train_step(model, char, cr, optimizer, reducer, input_, target,
                                                 opt.w_loss_CharbonnierLoss, opt.w_loss_vgg7)
            epoch_loss += loss
            if is_main and (i % opt.log_every == 0):
                print(f'\r{i}/{steps_per_epoch}: loss:{loss.item():.5f} = Charbonnier:{loss_rec.item():.5f}; '
                      f'contrast:{(loss_cr.item() if loss_cr is not None else 0):.5f} |time_used (Min):'
                      f'{(time.time() - t0) / 60:.1f}', end='', flush=True)
            if (i + 1) % eval_now == 0 and i > 0:
                val = evaluate()
                rng_states = all_rng_states()
                if is_main:
                    if val > best_psnr:
                        best_psnr, best_epoch, best_iter = val, epoch, i
                        torch.save({'epoch': epoch, 'state_dict': {'module.' + k: v for k, v in model.state_dict().items()},
                                    'optimizer': optimizer.state_dict(), 'rng_state': rng_states},
                                   os.path.join(model_dir, "model_best.pth"))
                    line = "[Ep %d it %d/%d\t PSNR: %.4f\t] ----  [best_Ep: %d, best_it: %d, Best_PSNR: %.4f]" % (
                        epoch, i, steps_per_epoch, val, best_epoch, best_iter, best_psnr)
                    print("\n" + line)
                    with open(logname, 'a') as f:
                        f.write(line + '\n')
        scheduler.step()
        rng_states = all_rng_states()
        if is_main:
            line = "Epoch: {}\tTime: {:.4f}\tLoss: {:.4f}\tLearningRate {:.6f}".format(
                epoch, time.time() - t0, epoch_loss.item(), scheduler.get_last_lr()[0])
            print("\n" + line)
            with open(logname, 'a') as f:
                f.write(line + '\n')
            torch.save({'epoch': epoch, 'state_dict': {'module.' + k: v for k, v in model.state_dict().items()},
                        'optimizer': optimizer.state_dict(), 'rng_state': rng_states},
                       os.path.join(model_dir, "epoch_model_{}.pth".format(epoch)))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
