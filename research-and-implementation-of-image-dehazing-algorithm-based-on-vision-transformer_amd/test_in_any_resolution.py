"""Drop-in for Uformer_ProbSparse/test_in_any_resolution.py: evaluation of images of ANY size - each image is centred in
a zero canvas whose side is a multiple of 128 (window 8 x 2^4 down-scalings; expand2square, :67-81), restored with the
padding announced to every LeWin block through the `mask` argument (1 - mask marks the padding: the blocks turn it
into a -100 attention mask per window, My_model_1.py:791-800), the valid region is cut back out (masked_select, :108) and
scored twice: with the repository's own torch metrics (utils.batch_PSNR / utils.SSIM, :113-119) and with the scikit-image
style metrics (:125-126, restated in utils/metrics.py).

Differences by design: one process / one GPU without nn.DataParallel; PIL instead of cv2; `--synthetic N` evaluates N
synthetic pairs of --height x --width.  Blocks that receive a mask run the unfused kernel chain (the fused window-attention
kernel covers the mask-free training / whole-image paths).
"""
import argparse
import math
import os
import sys

dir_name = os.path.dirname(os.path.abspath(__file__))
if dir_name not in sys.path:
    sys.path.insert(0, dir_name)

import torch  # noqa: E402

import utils  # noqa: E402
from utils.metrics import img_as_ubyte, peak_signal_noise_ratio as psnr_loss, structural_similarity as ssim_loss  # noqa: E402


def expand2square(timg, factor=16.0):
    """:67-81 - centre `timg` [1,3,h,w] in a zero X x X canvas, X = ceil(max(h,w) / factor) * factor; mask = 1 on the image."""
    _, _, h, w = timg.size()
    X = int(math.ceil(max(h, w) / float(factor)) * factor)
    img = torch.zeros(1, 3, X, X).type_as(timg)
    mask = torch.zeros(1, 1, X, X).type_as(timg)
    img[:, :, ((X - h) // 2):((X - h) // 2 + h), ((X - w) // 2):((X - w) // 2 + w)] = timg
    mask[:, :, ((X - h) // 2):((X - h) // 2 + h), ((X - w) // 2):((X - w) // 2 + w)].fill_(1.0)
    return img, mask


def restore_any(model, rgb_noisy, factor=128):
    """:101-108 for one image: pad to a square, forward with the padding mask, cut the valid region back out."""
    _, _, h, w = rgb_noisy.shape
    sq, mask = expand2square(rgb_noisy, factor=factor)
    restored = model(sq, 1 - mask)
    return torch.masked_select(restored, mask.bool()).reshape(1, 3, h, w)


def build_parser():
    parser = argparse.ArgumentParser(description='Dehazing evaluation at any resolution')
    parser.add_argument('--input_dir', default='../datasets/NH_haze/test/', type=str, help='Directory of validation images')
    parser.add_argument('--result_dir', default='./results/any_resolution/', type=str, help='Directory for results')
    parser.add_argument('--weights', default='', type=str, help='Path to weights (empty: random init)')
    parser.add_argument('--gpus', default='0', type=str, help='device index')
    parser.add_argument('--arch', default='Uformer', type=str, help='arch')
    parser.add_argument('--batch_size', default=1, type=int)
    parser.add_argument('--save_images', action='store_true', help='Save restored images in result directory')
    parser.add_argument('--embed_dim', type=int, default=32)
    parser.add_argument('--win_size', type=int, default=8)
    parser.add_argument('--token_projection', type=str, default='linear')
    parser.add_argument('--token_mlp', type=str, default='leff')
    parser.add_argument('--train_ps', type=int, default=128, help='patch size of training sample')
    parser.add_argument('--synthetic', type=int, default=0, help='evaluate N synthetic haze pairs instead of --input_dir')
    parser.add_argument('--height', type=int, default=300)
    parser.add_argument('--width', type=int, default=420)
    return parser


def main(argv=None):
    args = build_parser().parse_args(argv)
    if not torch.cuda.is_available():
        raise SystemExit("test_in_any_resolution.py needs a HIP device")
    dev = torch.device("cuda", int(str(args.gpus).split(",")[0]))
    torch.cuda.set_device(dev)
    if args.save_images:
        utils.mkdir(args.result_dir)
    model_restoration = utils.get_arch(args)
    if args.weights:
        utils.load_checkpoint(model_restoration, args.weights)
        print("===>Testing using weights: ", args.weights)
    model_restoration.to(dev).eval()

    if args.synthetic > 0:
        from dehaze_hip.train import synthetic_batch
        items = [synthetic_batch(1, (args.height, args.width), seed=700 + i, device="cpu") + (["synthetic_%03d.png" % i],)
                 for i in range(args.synthetic)]
    else:
        from utils.loader import get_validation_data
        ds = get_validation_data(args.input_dir)
        items = [(ds[i][0][None], ds[i][1][None], [ds[i][2]]) for i in range(len(ds))]

    psnr_val_rgb, ssim_val_rgb, psnr_val_rgb2, ssim_val_rgb2 = [], [], [], []
    with torch.no_grad():
        for gt, noisy, filenames in items:
            rgb_gt = gt.numpy().squeeze().transpose((1, 2, 0))
            rgb_restored = restore_any(model_restoration, noisy.to(dev), factor=128)
            ssim_val_rgb2.append(utils.SSIM(torch.clamp(rgb_restored, 0, 1).cpu(), torch.clamp(gt, 0, 1)).item())
            psnr_val_rgb2.append(utils.batch_PSNR(torch.clamp(rgb_restored, 0, 1).cpu(), torch.clamp(gt, 0, 1), False).item())
            rgb_restored = torch.clamp(rgb_restored, 0, 1).cpu().numpy().squeeze().transpose((1, 2, 0))
            psnr_val_rgb.append(psnr_loss(rgb_restored, rgb_gt))
            ssim_val_rgb.append(ssim_loss(rgb_restored, rgb_gt, multichannel=True))
            if args.save_images:
                utils.save_img(os.path.join(args.result_dir, filenames[0]), img_as_ubyte(rgb_restored))
    n = len(items)
    out = (sum(psnr_val_rgb) / n, sum(ssim_val_rgb) / n, sum(psnr_val_rgb2) / n, sum(ssim_val_rgb2) / n)
    print("PSNR: %f, SSIM: %f " % out[:2])
    print("PSNR2: %f, SSIM2: %f " % out[2:])
    return out


if __name__ == "__main__":
    main()
