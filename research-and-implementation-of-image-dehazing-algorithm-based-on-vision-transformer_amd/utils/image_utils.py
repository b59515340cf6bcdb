"""PSNR helpers (utils/image_utils.py:57-74 semantics: clamp to [0,1], MAX_I = 1)."""
import torch


def myPSNR(tar_img, prd_img):
    imdff = torch.clamp(prd_img, 0, 1) - torch.clamp(tar_img, 0, 1)
    rmse = (imdff ** 2).mean().sqrt()
    return 20 * torch.log10(1 / rmse)


def batch_PSNR(img1, img2, average=True):
    vals = [myPSNR(a, b) for a, b in zip(img1, img2)]
    return sum(vals) / len(vals) if average else sum(vals)
