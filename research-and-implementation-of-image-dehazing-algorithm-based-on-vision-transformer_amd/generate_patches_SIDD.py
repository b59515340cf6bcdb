"""Drop-in for Uformer_ProbSparse/generate_patches_SIDD.py: cut --num_patches random --ps x --ps patch pairs out of every
full-resolution (hazy, gt) PNG pair of --src_dir into --tar_dir/{hazy,gt}/<image>_<patch>.png (generate_patches_SIDD.py:
44-80).  Same crop draws (np.random.randint(0, H-PS), then (0, W-PS), per patch); PIL instead of cv2, a process pool
instead of joblib.  The output tree is what dataset.DataLoaderTrain / dataset.PatchStoreHBM read."""
import argparse
import os
import shutil
import sys
from glob import glob
from multiprocessing import Pool

dir_name = os.path.dirname(os.path.abspath(__file__))
if dir_name not in sys.path:
    sys.path.insert(0, dir_name)

import numpy as np  # noqa: E402

from utils import load_img_u8, save_img, natsorted  # noqa: E402


def save_files(job):
    i, noisy_file, clean_file, noisy_dir, clean_dir, PS, num_patches, seed = job
    if seed is not None:
        np.random.seed(seed + i)
    noisy_img, clean_img = load_img_u8(noisy_file), load_img_u8(clean_file)
    H, W = noisy_img.shape[0], noisy_img.shape[1]
    for j in range(num_patches):
        rr = np.random.randint(0, H - PS)
        cc = np.random.randint(0, W - PS)
        save_img(os.path.join(noisy_dir, '{}_{}.png'.format(i + 1, j + 1)), noisy_img[rr:rr + PS, cc:cc + PS, :])
        save_img(os.path.join(clean_dir, '{}_{}.png'.format(i + 1, j + 1)), clean_img[rr:rr + PS, cc:cc + PS, :])


def main(argv=None):
    parser = argparse.ArgumentParser(description='Generate patches from Full Resolution images')
    parser.add_argument('--src_dir', default='../datasets/NH_haze/train', type=str, help='Directory for full resolution images')
    parser.add_argument('--tar_dir', default='../datasets/NH_haze/train_patches', type=str, help='Directory for image patches')
    parser.add_argument('--ps', default=256, type=int, help='Image Patch Size')
    parser.add_argument('--num_patches', default=500, type=int, help='Number of patches per image')
    parser.add_argument('--num_cores', default=8, type=int, help='Number of CPU Cores')
    parser.add_argument('--seed', default=None, type=int, help='seed + image index per image (reproducible); default: unseeded like the reference')
    args = parser.parse_args(argv)
    noisy_dir, clean_dir = os.path.join(args.tar_dir, 'hazy'), os.path.join(args.tar_dir, 'gt')
    if os.path.exists(args.tar_dir):
        shutil.rmtree(args.tar_dir)
    os.makedirs(noisy_dir)
    os.makedirs(clean_dir)
    clean_files = natsorted(glob(os.path.join(args.src_dir, 'gt', '*.png')))
    noisy_files = natsorted(glob(os.path.join(args.src_dir, 'hazy', '*.png')))
    assert len(clean_files) == len(noisy_files) and clean_files, "need matching PNGs under <src_dir>/gt and <src_dir>/hazy"
    jobs = [(i, noisy_files[i], clean_files[i], noisy_dir, clean_dir, args.ps, args.num_patches, args.seed)
            for i in range(len(noisy_files))]
    if args.num_cores > 1 and len(jobs) > 1:
        with Pool(min(args.num_cores, len(jobs))) as pool:
            pool.map(save_files, jobs)
    else:
        for job in jobs:
            save_files(job)
    return len(jobs) * args.num_patches


if __name__ == "__main__":
    main()
