"""not-gpu: the callers either side of the hot path: augmentations + MixUp vs the reference's own
outputs (golden), the dataset classes on a temporary PNG tree, the wrap-copy padding, the restated scikit-image metrics."""
import os
import random

import numpy as np
import pytest
import torch

from oracle import data_oracle as DO

T = torch.from_numpy


def test_augmentations_vs_reference_golden(golden):
    g = golden("data_aug")
    import utils
    from dataset import transforms_aug, augment
    assert list(g["names"]) == transforms_aug
    x = T(g["x"])
    for k, name in enumerate(transforms_aug):
        assert torch.equal(getattr(augment, name)(x).contiguous(), T(g[f"t{k}"])), name
        assert torch.equal(DO.augment(x, k).contiguous(), T(g[f"t{k}"])), name


def test_mixup_vs_reference_golden(golden):
    g = golden("data_aug")
    import utils
    torch.manual_seed(2024)
    mg, mn = utils.MixUp_AUG().aug(T(g["mix_gt_in"]), T(g["mix_noisy_in"]))
    assert torch.equal(mg, T(g["mix_gt"])) and torch.equal(mn, T(g["mix_noisy"]))


def _png_tree(tmp_path, n=3, H=20, W=24, sub=("gt", "hazy")):
    import utils
    rng = np.random.default_rng(5)
    imgs = {}
    for s in sub:
        os.makedirs(tmp_path / s)
        for i in range(n):
            a = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
            utils.save_img(str(tmp_path / s / f"{i + 1}_1.png"), a)
            imgs[(s, i)] = a
    (tmp_path / sub[0] / "notes.txt").write_text("ignored")
    return imgs


def test_png_roundtrip_and_datasets(tmp_path):
    import utils
    from dataset import DataLoaderTrain, DataLoaderVal
    imgs = _png_tree(tmp_path)
    a = utils.load_img(str(tmp_path / "gt" / "1_1.png"))
    assert a.dtype == np.float32 and np.array_equal(a, imgs[("gt", 0)].astype(np.float32) / 255.)
    val = DataLoaderVal(str(tmp_path))
    assert len(val) == 3
    clean, noisy, fc, fn = val[1]
    assert fc == "2_1.png" and fn == "2_1.png" and clean.shape == (3, 20, 24)
    assert torch.equal(noisy, T(imgs[("hazy", 1)].astype(np.float32) / 255.).permute(2, 0, 1))
    tr = DataLoaderTrain(str(tmp_path), {"patch_size": 8})
    np.random.seed(3); random.seed(3)
    clean, noisy, _, _ = tr[2]
    np.random.seed(3); random.seed(3)
    r, c = np.random.randint(0, 20 - 8), np.random.randint(0, 24 - 8)
    k = random.getrandbits(3)
    want = DO.train_item(imgs[("gt", 2)], imgs[("hazy", 2)], r, c, k, 8)
    assert torch.equal(clean, want[0]) and torch.equal(noisy, want[1])
    full = DataLoaderTrain(str(tmp_path), {"patch_size": 20})       # H - ps == 0 -> r = c = 0 (dataset.py:56-58)
    np.random.seed(1); random.seed(1)
    clean, _, _, _ = full[0]
    assert clean.shape == (3, 20, 20)


@pytest.mark.parametrize("H,W,ps", [(40, 56, 16), (1200, 1600, 128), (24, 24, 16)])
def test_pad_wrap(H, W, ps):
    import test_long_GPU as TL
    x = torch.rand(1, 3, H, W)
    big = TL.pad_wrap(x, ps)
    assert torch.equal(big, DO.pad_wrap(x, ps))
    L = TL.padded_size(H, W, ps)
    assert big.shape[-1] == L and L % ps == 0 and L > max(H, W)
    if (H, W) == (1200, 1600):
        assert L == 1664                                            # the value test_long_GPU.py:81 hard-codes
    assert torch.equal(big[:, :, :H, W:], x[:, :, :, :L - W]) and torch.equal(big[:, :, H:, :W], x[:, :, :L - H, :])


def test_metrics_restatement():
    from utils import metrics as M
    rng = np.random.default_rng(0)
    a = rng.random((20, 23, 3)).astype(np.float32)
    b = np.clip(a + 0.1 * rng.standard_normal(a.shape).astype(np.float32), 0, 1)
    mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    assert abs(M.peak_signal_noise_ratio(a, b) - 10 * np.log10(1.0 / mse)) < 1e-4
    want = np.mean([DO.ssim_bruteforce(a[..., c], b[..., c]) for c in range(3)])
    assert abs(M.structural_similarity(a, b, multichannel=True) - want) < 1e-9
    assert abs(M.structural_similarity(a, a, multichannel=True) - 1.0) < 1e-12
    u8 = M.img_as_ubyte(np.array([[0.0, 0.5, 1.0, 0.49803922]], dtype=np.float32))
    assert u8.dtype == np.uint8 and u8.tolist() == [[0, 128, 255, 127]]
    with pytest.raises(ValueError):
        M.img_as_ubyte(np.array([1.5], dtype=np.float32))


def test_torch_ssim_identity():
    import utils
    x = torch.rand(2, 3, 32, 32)
    assert abs(float(utils.SSIM(x, x)) - 1.0) < 1e-5
    assert float(utils.SSIM(x, torch.rand(2, 3, 32, 32))) < 0.2


def test_generate_patches_and_store_roundtrip(tmp_path):
    """generate_patches_SIDD.py -> patch tree -> dataset.PatchStoreHBM.from_dir (CPU tensors here): every patch is a crop of
    its source image at the drawn origin, hazy/gt crops are aligned, names follow <image>_<patch>.png in natural order."""
    import generate_patches_SIDD as GP
    import utils
    from dataset import PatchStoreHBM
    src = tmp_path / "src"
    os.makedirs(src)
    imgs = _png_tree(src, n=2, H=40, W=52)
    n = GP.main(["--src_dir", str(src), "--tar_dir", str(tmp_path / "patches"), "--ps", "16", "--num_patches", "11",
                 "--num_cores", "1", "--seed", "4"])
    assert n == 22
    names = utils.natsorted(os.listdir(tmp_path / "patches" / "gt"))
    assert names[:3] == ["1_1.png", "1_2.png", "1_3.png"] and names[10] == "1_11.png" and names[11] == "2_1.png"
    np.random.seed(4 + 1)
    for j in range(11):
        rr, cc = np.random.randint(0, 40 - 16), np.random.randint(0, 52 - 16)
        for sub in ("gt", "hazy"):
            got = utils.load_img_u8(str(tmp_path / "patches" / sub / f"2_{j + 1}.png"))
            assert np.array_equal(got, imgs[(sub, 1)][rr:rr + 16, cc:cc + 16])
    store = PatchStoreHBM.from_dir(str(tmp_path / "patches"), "cpu")
    assert len(store) == 22 and store.gt.shape == (22, 16, 16, 3) and store.gt.dtype == torch.uint8
    half = PatchStoreHBM.from_dir(str(tmp_path / "patches"), "cpu", rank=1, world=2)
    assert len(half) == 11
