"""Drop-in for Uformer_ProbSparse/test_long_GPU.py (BASELINE config 5): whole-image evaluation - each test image is
padded to an L x L square by wrap-copying its left / top strips (test_long_GPU.py:76-89), restored in ONE forward on
[1,3,L,L] (L = 1664 for the 1200 x 1600 NH-HAZE test set: 43,264 windows per block at full resolution), cropped,
clamped and scored with PSNR / SSIM (:91-95), optionally saved as PNG (:97-98).

On the reference this forward needs a 48 GB card because ProbAttention materialises K_sample = 17.7 GB in the last
decoder stage (test_long_GPU.py:19); here the sampled scores are read out of the dense S tile in LDS, nothing of that
size exists; the whole forward peaks at about 16 GB of the 288 GB HBM (the 4C-wide LeFF tensors of the full-resolution stages).

Differences by design: one process / one GPU, no nn.DataParallel wrapper (checkpoints with or without the `module.`
prefix load, utils.load_checkpoint); PNG decode through PIL instead of cv2; scikit-image's metrics restated in
utils/metrics.py; `--synthetic N` evaluates N synthetic haze pairs of --height x --width instead of reading --input_dir;
L follows the general rule of :79-80 ((max(H,W) // ps + 1) * ps), which IS 1664 for the reference's 1200 x 1600 data
(the reference hard-codes that value in :81).
"""
import argparse
import os
import sys
import time

dir_name = os.path.dirname(os.path.abspath(__file__))
if dir_name not in sys.path:
    sys.path.insert(0, dir_name)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import utils  # noqa: E402
from utils.metrics import img_as_ubyte, peak_signal_noise_ratio as psnr_loss, structural_similarity as ssim_loss  # noqa: E402


def padded_size(H, W, train_ps):
    L = max(H, W)
    return (L // train_ps + 1) * train_ps


def pad_wrap(img, train_ps):
    """test_long_GPU.py:72-89: [B,C,H,W] -> [B,C,L,L]; right strip = the image's left L-W columns, bottom strip = the top
    L-H rows of the (already widened) canvas."""
    B, C, H, W = img.shape
    L = padded_size(H, W, train_ps)
    L_H, L_W = L - H, L - W
    assert L_W <= W and L_H <= H, "the wrap-copy padding needs L - W <= W and L - H <= H"
    big = torch.zeros((B, C, L, L), dtype=img.dtype, device=img.device)
    big[:, :, :H, :W] = img
    big[:, :, :H, W:W + L_W] = img[:, :, :, :L_W]
    big[:, :, H:H + L_H, :] = big[:, :, :L_H, :]
    return big


def restore_image(model, rgb_noisy, train_ps):
    """:72-93 for one batch: pad, one forward, crop, clamp."""
    B, C, H, W = rgb_noisy.shape
    restored = model(pad_wrap(rgb_noisy, train_ps))
    return torch.clamp(restored[:, :, :H, :W], 0, 1)


def build_parser():
    parser = argparse.ArgumentParser(description='Whole-image dehazing evaluation (NH-HAZE test set)')
    parser.add_argument('--input_dir', default='../datasets/NH_haze/test/', type=str, help='Directory of validation images')
    parser.add_argument('--result_dir', default='./results/long_NH/', type=str, help='Directory for results')
    parser.add_argument('--weights', default='', type=str, help='Path to weights (empty: random init)')
    parser.add_argument('--gpus', default='0', type=str, help='device index')
    parser.add_argument('--arch', default='Uformer', type=str, help='arch')
    parser.add_argument('--batch_size', default=1, type=int, help='Batch size for dataloader')
    parser.add_argument('--save_images', default='True', help='Save restored images in result directory')
    parser.add_argument('--embed_dim', type=int, default=32)
    parser.add_argument('--win_size', type=int, default=8)
    parser.add_argument('--token_projection', type=str, default='linear', help='linear/conv token projection')
    parser.add_argument('--token_mlp', type=str, default='leff', help='ffn/leff token mlp')
    parser.add_argument('--train_ps', type=int, default=128, help='patch size of training sample')
    parser.add_argument('--synthetic', type=int, default=0, help='evaluate N synthetic haze pairs instead of --input_dir')
    parser.add_argument('--height', type=int, default=1200)
    parser.add_argument('--width', type=int, default=1600)
    return parser


def main(argv=None):
    args = build_parser().parse_args(argv)
    if not torch.cuda.is_available():
        raise SystemExit("test_long_GPU.py needs a HIP device")
    dev = torch.device("cuda", int(str(args.gpus).split(",")[0]))
    torch.cuda.set_device(dev)
    save_images = str(args.save_images).lower() in ("true", "1", "yes")
    if save_images:
        utils.mkdir(args.result_dir)

    model_restoration = utils.get_arch(args)
    if args.weights:
        utils.load_checkpoint(model_restoration, args.weights)
        print("===>Testing using weights: ", args.weights)
    else:
        print("===>Testing with randomly initialised weights (no --weights)")
    model_restoration.to(dev).eval()

    if args.synthetic > 0:
        from dehaze_hip.train import synthetic_batch

        def items():
            for i in range(args.synthetic):
                gt, hazy = synthetic_batch(1, (args.height, args.width), seed=900 + i, device="cpu")
                yield gt, hazy, ["synthetic_%03d.png" % i]
        n_items = args.synthetic
        loader = items()
    else:
        from torch.utils.data import DataLoader
        from utils.loader import get_validation_data
        test_dataset = get_validation_data(args.input_dir)
        n_items = len(test_dataset)
        loader = DataLoader(dataset=test_dataset, batch_size=1, shuffle=False, num_workers=0, drop_last=False)

    psnr_val_rgb, ssim_val_rgb, secs = [], [], []
    with torch.no_grad():
        for data_test in loader:
            rgb_gt = data_test[0].numpy().squeeze().transpose((1, 2, 0))
            rgb_noisy = data_test[1].to(dev)
            filenames = data_test[2]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            rgb_restored = restore_image(model_restoration, rgb_noisy, args.train_ps)
            torch.cuda.synchronize()
            secs.append(time.perf_counter() - t0)
            rgb_restored = rgb_restored.cpu().numpy().squeeze().transpose((1, 2, 0))
            psnr_val_rgb.append(psnr_loss(rgb_restored, rgb_gt))
            ssim_val_rgb.append(ssim_loss(rgb_restored, rgb_gt, multichannel=True))
            if save_images:
                name = filenames[0] if isinstance(filenames[0], str) else filenames[0][0]
                utils.save_img(os.path.join(args.result_dir, name), img_as_ubyte(rgb_restored))
    psnr_val_rgb = sum(psnr_val_rgb) / n_items
    ssim_val_rgb = sum(ssim_val_rgb) / n_items
    print("PSNR: %f, SSIM: %f " % (psnr_val_rgb, ssim_val_rgb))
    steady = secs[1:] if len(secs) > 1 else secs
    print("forward (pad + model + crop): %.3f s/image over %d image(s) after warm-up" % (sum(steady) / len(steady), len(steady)))
    return psnr_val_rgb, ssim_val_rgb


if __name__ == "__main__":
    main()
