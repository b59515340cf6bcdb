"""PSNR / SSIM / uint8 conversion as test_long_GPU.py:15-17,94-98 uses them from scikit-image
(`peak_signal_noise_ratio`, `structural_similarity(..., multichannel=True)`, `img_as_ubyte`).

scikit-image is a third-party dependency that is absent from this image, so these are restatements of its published
algorithm (the 0.16-0.18 API generation the reference's `multichannel=True` call belongs to) - PARITY UNPINNED against
the package itself; tests pin them against brute-force evaluations of the same formulas.  Points that matter:
  * PSNR: data_range = 1 for float images with min >= 0 (2 otherwise); squared error in the input float type, mean in
    float64.
  * SSIM: 7x7 uniform window, sample covariance (N/(N-1)), K1 = 0.01, K2 = 0.03, per channel in float64, borders of
    (win-1)/2 cropped before the mean, channel mean; for float images data_range = dmax - dmin of the dtype = 2
    (yes, 2 - the package's documented behaviour when data_range is not passed, which the reference does not).
"""
import numpy as np
from scipy.ndimage import uniform_filter


def peak_signal_noise_ratio(image_true, image_test, data_range=None):
    image_true, image_test = np.asarray(image_true), np.asarray(image_test)
    assert image_true.shape == image_test.shape
    if data_range is None:
        if np.issubdtype(image_true.dtype, np.floating):
            tmin, tmax = float(image_true.min()), float(image_true.max())
            if tmax > 1 or tmin < -1:
                raise ValueError("image_true has intensity values outside the range expected for its data type")
            data_range = 1.0 if tmin >= 0 else 2.0
        else:
            info = np.iinfo(image_true.dtype)
            data_range = float(info.max) if image_true.min() >= 0 else float(info.max - info.min)
    ft = np.result_type(image_true.dtype, image_test.dtype, np.float32)
    err = np.mean((image_true.astype(ft) - image_test.astype(ft)) ** 2, dtype=np.float64)
    return 10 * np.log10((data_range ** 2) / err)


def _ssim_plane(im1, im2, win_size, data_range, K1=0.01, K2=0.03):
    im1, im2 = im1.astype(np.float64), im2.astype(np.float64)
    NP = win_size ** im1.ndim
    cov_norm = NP / (NP - 1)
    ux, uy = uniform_filter(im1, size=win_size), uniform_filter(im2, size=win_size)
    uxx = uniform_filter(im1 * im1, size=win_size)
    uyy = uniform_filter(im2 * im2, size=win_size)
    uxy = uniform_filter(im1 * im2, size=win_size)
    vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
    C1, C2 = (K1 * data_range) ** 2, (K2 * data_range) ** 2
    S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux ** 2 + uy ** 2 + C1) * (vx + vy + C2))
    pad = (win_size - 1) // 2
    return S[pad:S.shape[0] - pad, pad:S.shape[1] - pad].mean()


def structural_similarity(im1, im2, win_size=7, data_range=None, multichannel=False):
    im1, im2 = np.asarray(im1), np.asarray(im2)
    assert im1.shape == im2.shape
    if data_range is None:
        if np.issubdtype(im1.dtype, np.floating):
            data_range = 2.0
        else:
            info = np.iinfo(im1.dtype)
            data_range = float(info.max - info.min)
    if multichannel:
        return float(np.mean([_ssim_plane(im1[..., c], im2[..., c], win_size, data_range) for c in range(im1.shape[-1])]))
    return float(_ssim_plane(im1, im2, win_size, data_range))


def img_as_ubyte(image):
    image = np.asarray(image)
    if image.dtype == np.uint8:
        return image
    if image.min() < -1.0 or image.max() > 1.0:
        raise ValueError("Images of type float must be between -1 and 1.")
    out = np.multiply(image, 255, dtype=np.float32)
    np.rint(out, out=out)
    np.clip(out, 0, 255, out=out)
    return out.astype(np.uint8)
