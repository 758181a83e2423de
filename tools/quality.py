#!/usr/bin/env python3
"""fp16-vs-fp32 image metrics the way the reference's evaluations.ipynb computes them (cell "calculate_metrics": 8-bit PPM values,
cv2 RGB->gray, skimage structural_similarity with its defaults, cv2.PSNR) — restated with numpy/scipy, for the frames this build
renders on MI355X.  The reference reports SSIM 0.4317 / PSNR 12.4 dB for its fp16 build against its fp32 build.  GPU box only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dd2360-raytracing_amd"))
import numpy as np
from scipy.ndimage import uniform_filter
import torch
import rt_amd as rt


def ppm_levels(fb):
    """int(255.99 * c) per channel as output_to_stream writes it (main.cu:321-333), top row first"""
    a = np.nan_to_num(np.asarray(fb, np.float64), nan=0.0, posinf=1.0, neginf=0.0)
    return np.clip((255.99 * a).astype(np.int64), 0, 255)[::-1]


def gray8(rgb):
    # cv2.cvtColor(..., COLOR_RGB2GRAY) on uint8: fixed-point 0.299 / 0.587 / 0.114 with rounding
    r, g, b = rgb[..., 0], rgb[..., 1], rgb[..., 2]
    return ((r * 4899 + g * 9617 + b * 1868 + 8192) >> 14).astype(np.float64)


def ssim(x, y, win=7, data_range=255.0):
    # skimage.metrics.structural_similarity defaults: uniform 7x7 window, sample covariance, K1 = 0.01, K2 = 0.03, borders cropped
    npx = win * win
    cov_norm = npx / (npx - 1.0)
    ux, uy = uniform_filter(x, win), uniform_filter(y, win)
    uxx, uyy, uxy = uniform_filter(x * x, win), uniform_filter(y * y, win), uniform_filter(x * y, win)
    vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
    c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    s = ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux * ux + uy * uy + c1) * (vx + vy + c2))
    pad = (win - 1) // 2
    return float(s[pad:-pad, pad:-pad].mean())


def psnr(x, y):
    mse = float(((x - y) ** 2).mean())
    return float("inf") if mse == 0 else 10.0 * np.log10(255.0 ** 2 / mse)


def frame(n, nx, ny, ns, spl, precision):
    W = rt.World(n, nx, ny, precision=precision).upload(); O = rt.Octree(W, spl).upload()
    st = rt.alloc_rand_state(nx, ny); fb = rt.alloc_fb(nx, ny, precision=precision)
    rt.render_init(nx, ny, st); rt.render(fb, nx, ny, ns, W, st, O); torch.cuda.synchronize()
    return gray8(ppm_levels(fb.float().cpu().numpy().reshape(ny, nx, 3)))


if __name__ == "__main__":
    for n, spl, ns in ((8000, 30, 10), (10000, 32, 64)):           # the reference's default configuration (main.cu:22-24, :348-350) and C3/C4
        a = frame(n, 1200, 800, ns, spl, rt.FP32); b = frame(n, 1200, 800, ns, spl, rt.FP16)
        print("N=%d SPL=%d 1200x800x%d: fp16 vs fp32  SSIM %.4f  PSNR %.2f dB   (reference, its own fp16 vs fp32 build: SSIM 0.4317, PSNR 12.4 dB)"
              % (n, spl, ns, ssim(a, b), psnr(a, b)), flush=True)
    c = frame(10000, 1200, 800, 64, 32, rt.FP32)
    print("determinism: fp32 vs fp32 SSIM %.4f PSNR %s" % (ssim(a, c), psnr(a, c)))
