"""Drop-in for Uformer_ProbSparse/utils/image_utils.py: file predicates, PNG load/save, PSNR and the Gaussian-window
SSIM.  The reference decodes with cv2 (BGR -> RGB, /255, image_utils.py:43-49); cv2 is not a dependency here - PIL
decodes the same 8-bit RGB values, so load_img returns bit-identical float32 arrays for PNG input."""
import pickle
from math import exp

import numpy as np
import torch
import torch.nn.functional as F


def _has_ext(filename, *exts):
    return str(filename).endswith(exts)


def is_numpy_file(filename):
    return _has_ext(filename, ".npy")


def is_image_file(filename):
    return _has_ext(filename, ".jpg")


def is_png_file(filename):
    return _has_ext(filename, ".png")


def is_pkl_file(filename):
    return _has_ext(filename, ".pkl")


def load_pkl(filename_):
    with open(filename_, 'rb') as fh:
        return pickle.load(fh)


def save_dict(dict_, filename_):
    with open(filename_, 'wb') as fh:
        pickle.dump(dict_, fh)


def load_npy(filepath):
    return np.load(filepath)


def load_img_u8(filepath):
    """[H, W, 3] uint8 RGB."""
    from PIL import Image
    with Image.open(filepath) as im:
        return np.asarray(im.convert("RGB"), dtype=np.uint8)


def load_img(filepath):
    """[H, W, 3] float32 RGB in [0, 1] (image_utils.py:43-49)."""
    return load_img_u8(filepath).astype(np.float32) / 255.


def save_img(filepath, img):
    """img: [H, W, 3] uint8 RGB (image_utils.py:52-53)."""
    from PIL import Image
    Image.fromarray(np.asarray(img, dtype=np.uint8), "RGB").save(filepath)


# ---- PSNR (image_utils.py:57-74: clamp to [0,1], MAX_I = 1)
def myPSNR(tar_img, prd_img):
    imdff = torch.clamp(prd_img, 0, 1) - torch.clamp(tar_img, 0, 1)
    rmse = (imdff ** 2).mean().sqrt()
    return 20 * torch.log10(1 / rmse)


def batch_PSNR(img1, img2, average=True):
    vals = [myPSNR(a, b) for a, b in zip(img1, img2)]
    return sum(vals) / len(vals) if average else sum(vals)


# ---- SSIM with an 11x11 Gaussian window, sigma 1.5 (image_utils.py:78-127): local moments by depthwise convolution
def gaussian(window_size, sigma):
    x = torch.arange(window_size, dtype=torch.float32) - window_size // 2
    g = torch.tensor([exp(-float(v) ** 2 / float(2 * sigma ** 2)) for v in x])
    return g / g.sum()


def create_window(window_size, channel):
    g1 = gaussian(window_size, 1.5)
    g2 = torch.outer(g1, g1).float()
    return g2.expand(channel, 1, window_size, window_size).contiguous()


def _local_mean(t, window, channel):
    return F.conv2d(t, window, padding=window.shape[-1] // 2, groups=channel)


def _ssim(img1, img2, window, window_size, channel, size_average=True):
    m1, m2 = _local_mean(img1, window, channel), _local_mean(img2, window, channel)
    v1 = _local_mean(img1 * img1, window, channel) - m1 * m1
    v2 = _local_mean(img2 * img2, window, channel) - m2 * m2
    cov = _local_mean(img1 * img2, window, channel) - m1 * m2
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    smap = ((2 * m1 * m2 + c1) * (2 * cov + c2)) / ((m1 * m1 + m2 * m2 + c1) * (v1 + v2 + c2))
    return smap.mean() if size_average else smap.mean(dim=(1, 2, 3))


def SSIM(img1, img2, window_size=11, size_average=True):
    """Both images are clamped to [0, 1] first, like the PSNR helpers."""
    a, b = img1.clamp(0, 1), img2.clamp(0, 1)
    channel = a.size(1)
    window = create_window(window_size, channel).to(device=a.device, dtype=a.dtype)
    return _ssim(a, b, window, window_size, channel, size_average)
