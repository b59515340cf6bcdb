"""-m gpu: checkpoint / resume (TR:99-117, TR:294-333, utils/model_utils.py:28-77).

A run interrupted after k steps and resumed from its checkpoint - written in the REFERENCE's format ('module.'-prefixed
state_dict, torch.optim.AdamW's positional optimizer state over all 488 parameters, epoch) plus the generator states the
reference omits - must continue like the uninterrupted run: same sampled keys, same DropPath draws, same AdamW moments and
step count.  Bit-identity is not on offer (weight gradients are summed with fp32 atomics whose order differs from launch to
launch), so the check is at rounding level: two uninterrupted runs differ by the same amount."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _seed(s):
    import random
    import numpy as np
    random.seed(s); np.random.seed(s); torch.manual_seed(s); torch.cuda.manual_seed_all(s)


def _make(dev, seed):
    import My_model_1 as M1
    from dehaze_hip.train import FlatAdamW
    _seed(seed)
    model = M1.Uformer(img_size=128, embed_dim=32, win_size=8, token_projection='linear', token_mlp='leff').to(dev).train()
    return model, FlatAdamW(model, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.02)


def test_resume_continues_the_interrupted_run(tmp_path):
    import utils
    from dehaze_hip.train import synthetic_batch, train_step
    from losses import CharbonnierLoss
    dev = torch.device("cuda:0")
    gt, hazy = synthetic_batch(2, 128, seed=11, device=dev)
    char = CharbonnierLoss()
    path = str(tmp_path / "epoch_model_1.pth")

    def steps(model, opt, n):
        return [train_step(model, char, None, opt, None, hazy, gt, 1.0, 0.0)[0].item() for _ in range(n)]

    # uninterrupted: 2 + 2 steps, checkpoint after the first two
    model, opt = _make(dev, 1234)
    first = steps(model, opt, 2)
    torch.save({'epoch': 1, 'state_dict': {'module.' + k: v for k, v in model.state_dict().items()},
                'optimizer': opt.state_dict(), 'rng_state': utils.rng_state_dict()}, path)
    ref_losses = steps(model, opt, 2)
    ref_sd = {k: v.detach().clone() for k, v in model.state_dict().items()}

    # the file is what the reference's loaders expect
    ck = torch.load(path, map_location="cpu")                       # default weights_only=True: tensors and plain containers only
    assert all(k.startswith("module.") for k in ck["state_dict"]) and len(ck["state_dict"]) == 488
    nparams = len(list(model.parameters()))
    assert ck["optimizer"]["param_groups"][0]["params"] == list(range(nparams))
    assert all(float(st["step"]) == 2.0 for st in ck["optimizer"]["state"].values())
    live = len(model.live_parameters())
    assert len(ck["optimizer"]["state"]) == live == nparams - 108          # the dead attn.qkv / attn.proj tensors have no state

    # resumed: a differently initialised model and random streams, everything restored from the file
    model2, opt2 = _make(dev, 999)
    utils.load_checkpoint(model2, path, map_location=dev)
    assert utils.load_start_epoch(path) == 1
    assert utils.load_optim(opt2, path) == pytest.approx(2e-4)
    assert opt2._step == 2
    sd_file = opt.state_dict()["state"]
    for i, st in opt2.state_dict()["state"].items():                 # moments round-trip exactly (compare with the file)
        assert torch.equal(st["exp_avg"].cpu(), ck["optimizer"]["state"][i]["exp_avg"])
        assert torch.equal(st["exp_avg_sq"].cpu(), ck["optimizer"]["state"][i]["exp_avg_sq"])
    del sd_file
    torch.rand(7); torch.rand(3, device=dev)                         # disturb the streams before restoring them
    assert utils.load_rng_state(path)
    got_losses = steps(model2, opt2, 2)
    assert got_losses == pytest.approx(ref_losses, rel=2e-5, abs=2e-6), (got_losses, ref_losses, first)
    worst = max((model2.state_dict()[k] - ref_sd[k]).abs().max().item() for k in ref_sd if ref_sd[k].dtype.is_floating_point)
    assert worst < 2e-5, worst       # AdamW's normalised update amplifies ulp-level gradient differences to ~lr * 1e-2

    # without the generator states the continuation diverges (different sampled keys / DropPath) - the states matter
    model3, opt3 = _make(dev, 999)
    utils.load_checkpoint(model3, path, map_location=dev)
    utils.load_optim(opt3, path)
    other = steps(model3, opt3, 2)
    assert abs(other[1] - ref_losses[1]) > 1e-6 or abs(other[0] - ref_losses[0]) > 1e-6

    # a reference checkpoint (no 'rng_state' key) loads the same way and reports that there is nothing to restore
    ck.pop("rng_state")
    path2 = str(tmp_path / "reference_format.pth")
    torch.save(ck, path2)
    assert utils.load_rng_state(path2) is False
    model4, opt4 = _make(dev, 5)
    utils.load_checkpoint(model4, path2, map_location=dev)
    utils.load_optim(opt4, path2)
    assert opt4._step == 2
