"""PSNR / SSIM scorer with the reference's command line (/root/reference/PSNRSSIM.py):

    python PSNRSSIM.py --gt_dir A --result_dir B

prints one line per image pair and the 4-decimal means.  Numerics follow the reference:
PSNR on float/255 after a 1-pixel border strip; SSIM per channel on uint8 with a sigma-1.5 Gaussian
window (scipy.ndimage.gaussian_filter, 13 taps, reflect), population covariance, data range 255,
5-pixel crop; pairs are formed from the two sorted *.png listings.  The reference computes every
pair twice (:262-264) -- not reproduced.  CPU-only, like the reference.
"""
import argparse
import os
from decimal import Decimal

import numpy as np
from scipy.ndimage import gaussian_filter

SCALE = 1


def output_psnr_mse(img_orig, img_out):
    """PSNRSSIM.py:201-205."""
    mse = np.mean(np.square(img_orig - img_out))
    return 10 * np.log10(1.0 / mse)


def compare_ssim(X, Y, sigma=1.5, K1=0.01, K2=0.03, win_size=11, data_range=None):
    """The configuration the reference calls (gaussian_weights=True, use_sample_covariance=False),
    PSNRSSIM.py:46-194, for one 2-D channel."""
    if X.dtype != Y.dtype:
        raise ValueError('Input images must have the same dtype.')
    if X.shape != Y.shape:
        raise ValueError('Input images must have the same dimensions.')
    if min(X.shape) < win_size:
        raise ValueError('win_size exceeds image extent.')
    if data_range is None:
        data_range = 255 if X.dtype == np.uint8 else 2
    X, Y = X.astype(np.float64), Y.astype(np.float64)
    ux, uy = gaussian_filter(X, sigma), gaussian_filter(Y, sigma)
    vx = gaussian_filter(X * X, sigma) - ux * ux
    vy = gaussian_filter(Y * Y, sigma) - uy * uy
    vxy = gaussian_filter(X * Y, sigma) - ux * uy
    C1, C2 = (K1 * data_range) ** 2, (K2 * data_range) ** 2
    S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux ** 2 + uy ** 2 + C1) * (vx + vy + C2))
    pad = (win_size - 1) // 2
    return S[pad:-pad, pad:-pad].mean()


def _strip(F):
    h, w, _ = F.shape
    F = F[:h - h % SCALE, :w - w % SCALE, :]
    return F[SCALE:-SCALE, SCALE:-SCALE, :]


def _imread(path):
    from PIL import Image
    return np.asarray(Image.open(path).convert('RGB'))


def psnr_images(ref_u8, res_u8):
    """PSNRSSIM.py:208-230 on decoded uint8 HWC images."""
    return output_psnr_mse(_strip(ref_u8.astype(float) / 255.0), _strip(res_u8.astype(float) / 255.0))


def mssim_images(ref_u8, res_u8):
    """PSNRSSIM.py:233-240."""
    a, b = _strip(ref_u8), _strip(res_u8)
    return np.mean([compare_ssim(a[:, :, i], b[:, :, i]) for i in range(3)])


def score_dirs(gt_dir, result_dir, verbose=True):
    # the reference swaps the two names (:245-246); both metrics are symmetric
    res_dir, ref_dir = gt_dir, result_dir
    ref_pngs = sorted(p for p in os.listdir(ref_dir) if p.lower().endswith('png'))
    res_pngs = sorted(p for p in os.listdir(res_dir) if p.lower().endswith('png'))
    scores, scores_ssim = [], []
    for ref_im, res_im in zip(ref_pngs, res_pngs):
        a, b = _imread(os.path.join(ref_dir, ref_im)), _imread(os.path.join(res_dir, res_im))
        p, s = psnr_images(a, b), mssim_images(a, b)
        if verbose:
            print(ref_im, res_im, 'psnr:', p, 'ssim:', s)
        scores.append(p)
        scores_ssim.append(s)
    return scores, scores_ssim


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument('--gt_dir', default='', help="path to GT images")
    parser.add_argument('--result_dir', default='', help="path to dehazed images")
    opt = parser.parse_args(argv)
    scores, scores_ssim = score_dirs(opt.gt_dir, opt.result_dir)
    psnr = Decimal(float(np.mean(scores))).quantize(Decimal('0.0000'))
    mssim = Decimal(float(np.mean(scores_ssim))).quantize(Decimal('0.0000'))
    print("\n psnr:\n", psnr, '\n compute ssim:\n', mssim)
    return float(psnr), float(mssim)


if __name__ == '__main__':
    main()
