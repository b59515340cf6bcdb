"""Drop-in for Uformer_ProbSparse/utils/image_utils.py: file predicates, PNG load/save, PSNR and the Gaussian-window
SSIM.  The reference decodes with cv2 (BGR -> RGB, /255, image_utils.py:43-49); cv2 is not a dependency here - PIL
decodes the same 8-bit RGB values, so load_img returns bit-identical float32 arrays for PNG input."""
import pickle
from math import exp

import numpy as np
import torch
import torch.nn.functional as F


def is_numpy_file(filename):
    return any(filename.endswith(extension) for extension in [".npy"])


def is_image_file(filename):
    return any(filename.endswith(extension) for extension in [".jpg"])


def is_png_file(filename):
    return any(filename.endswith(extension) for extension in [".png"])


def is_pkl_file(filename):
    return any(filename.endswith(extension) for extension in [".pkl"])


def load_pkl(filename_):
    with open(filename_, 'rb') as f:
        return pickle.load(f)


def save_dict(dict_, filename_):
    with open(filename_, 'wb') as f:
        pickle.dump(dict_, f)


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


# ---- SSIM with an 11x11 Gaussian window, sigma 1.5 (image_utils.py:78-127)
def gaussian(window_size, sigma):
    gauss = torch.tensor([exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(window_size)])
    return gauss / gauss.sum()


def create_window(window_size, channel):
    w1 = gaussian(window_size, 1.5).unsqueeze(1)
    w2 = w1.mm(w1.t()).float().unsqueeze(0).unsqueeze(0)
    return w2.expand(channel, 1, window_size, window_size).contiguous()


def _ssim(img1, img2, window, window_size, channel, size_average=True):
    pad = window_size // 2
    mu1 = F.conv2d(img1, window, padding=pad, groups=channel)
    mu2 = F.conv2d(img2, window, padding=pad, groups=channel)
    mu1_sq, mu2_sq, mu1_mu2 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    sigma1_sq = F.conv2d(img1 * img1, window, padding=pad, groups=channel) - mu1_sq
    sigma2_sq = F.conv2d(img2 * img2, window, padding=pad, groups=channel) - mu2_sq
    sigma12 = F.conv2d(img1 * img2, window, padding=pad, groups=channel) - mu1_mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    ssim_map = ((2 * mu1_mu2 + C1) * (2 * sigma12 + C2)) / ((mu1_sq + mu2_sq + C1) * (sigma1_sq + sigma2_sq + C2))
    return ssim_map.mean() if size_average else ssim_map.mean(1).mean(1).mean(1)


def SSIM(img1, img2, window_size=11, size_average=True):
    img1 = torch.clamp(img1, min=0, max=1)
    img2 = torch.clamp(img2, min=0, max=1)
    channel = img1.size(1)
    window = create_window(window_size, channel).to(img1.device).type_as(img1)
    return _ssim(img1, img2, window, window_size, channel, size_average)
