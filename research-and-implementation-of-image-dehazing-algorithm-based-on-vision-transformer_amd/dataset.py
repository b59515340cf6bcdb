"""Drop-in for Uformer_ProbSparse/dataset.py (PNG pair datasets over <dir>/gt and <dir>/hazy) plus the MI355X-native
feed for training: `PatchStoreHBM`.

The reference's DataLoaderTrain (dataset.py:17-77) decodes two PNGs per item on the host, crops a random ps x ps
window, applies one of 8 rotate/flip augmentations and ships float32 through DataLoader workers and PCIe every step.
`PatchStoreHBM` decodes every patch pair ONCE into two uint8 [N,H,W,3] tensors resident in HBM (NH-HAZE train
patches: 2 x 5.4 GB of 288 GB) and produces a batch with one kernel launch (dhz_crop_augment_pair: gather + crop +
rotate/flip + /255 to float32 CHW), drawing (r, c, augmentation) per item from the same host generators in the same
order as DataLoaderTrain.__getitem__, so a batch equals what the reference dataset returns for those items and draws.
"""
import os
import random

import numpy as np
import torch
from torch.utils.data import Dataset

from utils import is_png_file, load_img, load_img_u8, Augment_RGB_torch

augment = Augment_RGB_torch()
transforms_aug = [method for method in dir(augment) if callable(getattr(augment, method)) if not method.startswith('_')]


def _pair_files(rgb_dir, gt_dir='gt', input_dir='hazy'):
    clean_files = sorted(os.listdir(os.path.join(rgb_dir, gt_dir)))
    noisy_files = sorted(os.listdir(os.path.join(rgb_dir, input_dir)))
    clean = [os.path.join(rgb_dir, gt_dir, x) for x in clean_files if is_png_file(x)]
    noisy = [os.path.join(rgb_dir, input_dir, x) for x in noisy_files if is_png_file(x)]
    return clean, noisy


def draw_crop_aug(H, W, ps):
    """(r, c, augmentation index) exactly as dataset.py:56-70 draws them (numpy global RNG, then `random`)."""
    if H - ps == 0:
        r, c = 0, 0
    else:
        r = np.random.randint(0, H - ps)
        c = np.random.randint(0, W - ps)
    return r, c, random.getrandbits(3)


def _chw(path):
    """One PNG as float32 CHW in [0, 1] (the reference's load_img + permute)."""
    return torch.from_numpy(np.float32(load_img(path))).permute(2, 0, 1)


class _PngFolder(Dataset):
    """Host-side datasets of the reference's evaluation scripts: item -> decoded tensors + file names.  `pairs` = gt / hazy
    sub-directories (training, validation), otherwise a HAZY directory only (test).  Attribute names are the reference's."""

    def __init__(self, rgb_dir, pairs=True, target_transform=None):
        super().__init__()
        self.target_transform = target_transform
        if pairs:
            self.clean_filenames, self.noisy_filenames = _pair_files(rgb_dir)
        else:
            self.clean_filenames = None
            self.noisy_filenames = [os.path.join(rgb_dir, 'HAZY', f) for f in sorted(os.listdir(os.path.join(rgb_dir, 'HAZY')))
                                    if is_png_file(f)]
        self.tar_size = len(self.noisy_filenames)

    def __len__(self):
        return self.tar_size

    def _pair(self, index):
        i = index % self.tar_size
        cf, nf = self.clean_filenames[i], self.noisy_filenames[i]
        return _chw(cf), _chw(nf), os.path.basename(cf), os.path.basename(nf)


class DataLoaderVal(_PngFolder):                       # dataset.py:82-110: whole image pairs
    def __init__(self, rgb_dir, target_transform=None):
        super().__init__(rgb_dir, True, target_transform)

    def __getitem__(self, index):
        return self._pair(index)


class DataLoaderTrain(_PngFolder):                     # dataset.py:17-77: random ps x ps crop + one of 8 rotations / flips
    def __init__(self, rgb_dir, img_options=None, target_transform=None):
        super().__init__(rgb_dir, True, target_transform)
        self.img_options = img_options

    def __getitem__(self, index):
        clean, noisy, cn, nn_ = self._pair(index)
        ps = self.img_options['patch_size']
        r, c, k = draw_crop_aug(clean.shape[1], clean.shape[2], ps)        # the reference's draw order (numpy, then random)
        aug = getattr(augment, transforms_aug[k])
        return aug(clean[:, r:r + ps, c:c + ps]), aug(noisy[:, r:r + ps, c:c + ps]), cn, nn_


class DataLoaderTest(_PngFolder):                      # dataset.py:115-138: hazy images only
    def __init__(self, rgb_dir, target_transform=None):
        super().__init__(rgb_dir, False, target_transform)

    def __getitem__(self, index):
        f = self.noisy_filenames[index % self.tar_size]
        return _chw(f), os.path.basename(f)


class DataLoaderTestSR(Dataset):                       # dataset.py:205-232: PNGs directly under rgb_dir -> (tensor, file name)
    def __init__(self, rgb_dir, target_transform=None):
        super().__init__()
        self.target_transform = target_transform
        self.LR_filenames = [os.path.join(rgb_dir, f) for f in sorted(os.listdir(rgb_dir)) if is_png_file(f)]
        self.tar_size = len(self.LR_filenames)

    def __len__(self):
        return self.tar_size

    def __getitem__(self, index):
        f = self.LR_filenames[index % self.tar_size]
        return _chw(f), os.path.split(f)[-1]


class PatchStoreHBM:
    """All (gt, hazy) patch pairs of a directory as uint8 [N,H,W,3] tensors in HBM + the batch kernel."""

    def __init__(self, gt_u8, hazy_u8, device):
        assert gt_u8.dtype == torch.uint8 and gt_u8.shape == hazy_u8.shape and gt_u8.dim() == 4 and gt_u8.shape[-1] == 3
        self.device = torch.device(device)
        self.gt = gt_u8.contiguous().to(self.device)
        self.hazy = hazy_u8.contiguous().to(self.device)
        self.N, self.H, self.W = gt_u8.shape[0], gt_u8.shape[1], gt_u8.shape[2]
        self._table = None

    @classmethod
    def from_dir(cls, rgb_dir, device, rank=0, world=1):
        clean, noisy = _pair_files(rgb_dir)
        clean, noisy = clean[rank::world], noisy[rank::world]
        assert clean, f"no PNG pairs under {rgb_dir}/gt and {rgb_dir}/hazy"
        first = load_img_u8(clean[0])
        gt = torch.empty((len(clean),) + first.shape, dtype=torch.uint8).pin_memory() if torch.cuda.is_available() \
            else torch.empty((len(clean),) + first.shape, dtype=torch.uint8)
        hz = torch.empty_like(gt)
        for i, (a, b) in enumerate(zip(clean, noisy)):
            ia, ib = load_img_u8(a), load_img_u8(b)
            assert ia.shape == first.shape and ib.shape == first.shape, "patches must share one size (generate_patches_SIDD.py)"
            gt[i] = torch.from_numpy(ia)
            hz[i] = torch.from_numpy(ib)
        return cls(gt, hz, device)

    def __len__(self):
        return self.N

    def batch(self, indices, ps):
        """indices: iterable of patch ids.  Returns (clean, noisy) float32 [n,3,ps,ps] on the device; per item the crop
        origin and the augmentation are drawn like DataLoaderTrain.__getitem__ does, in item order."""
        from dehaze_hip import _lib
        from dehaze_hip.ops import _p, _stream
        idx = [int(i) % self.N for i in indices]
        n = len(idx)
        tab = torch.empty((n, 4), dtype=torch.int32)
        if self.device.type == "cuda":
            tab = tab.pin_memory()
        for j, i in enumerate(idx):
            r, c, k = draw_crop_aug(self.H, self.W, ps)
            tab[j, 0], tab[j, 1], tab[j, 2], tab[j, 3] = i, r, c, k
        self._table = tab                                       # keep the pinned buffer alive until the copy ran
        tab_d = tab.to(self.device, non_blocking=True)
        clean = torch.empty((n, 3, ps, ps), device=self.device, dtype=torch.float32)
        noisy = torch.empty_like(clean)
        _lib.call("dhz_crop_augment_pair", self.gt.data_ptr(), self.hazy.data_ptr(), tab_d.data_ptr(), _p(clean), _p(noisy),
                  n, self.H, self.W, ps, _stream())
        return clean, noisy
