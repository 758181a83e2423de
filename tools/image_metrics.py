"""Image metrics as the reference's evaluations.ipynb computes them (cell 11, `calculate_metrics` :1021 / `compare_experiment_images`
:1027: 8-bit PPM values, cv2 RGB->gray, skimage.metrics.structural_similarity with its defaults, cv2.PSNR) — restated with
numpy / scipy (neither cv2 nor skimage is in the image).  No GPU needed: shared by tools/quality.py, tools/run_experiment.py and
the CPU tests."""
import numpy as np
from scipy.ndimage import uniform_filter


def ppm_levels(fb):
    """int(255.99 * c) per channel as output_to_stream writes it (main.cu:321-333), top row first"""
    a = np.nan_to_num(np.asarray(fb, np.float64), nan=0.0, posinf=1.0, neginf=0.0)
    return np.clip((255.99 * a).astype(np.int64), 0, 255)[::-1]


def gray8(rgb):
    """cv2.cvtColor(..., COLOR_RGB2GRAY) on uint8: fixed-point 0.299 / 0.587 / 0.114 with rounding"""
    r, g, b = rgb[..., 0], rgb[..., 1], rgb[..., 2]
    return ((r * 4899 + g * 9617 + b * 1868 + 8192) >> 14).astype(np.float64)


def ssim(x, y, win=7, data_range=255.0):
    """skimage.metrics.structural_similarity defaults: uniform 7x7 window, sample covariance, K1 = 0.01, K2 = 0.03, borders cropped"""
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
    """cv2.PSNR on 8-bit values: 10 log10(255^2 / MSE); identical images -> inf"""
    mse = float(((x - y) ** 2).mean())
    return float("inf") if mse == 0 else float(10.0 * np.log10(255.0 ** 2 / mse))


def compare_frames(fb_a, fb_b):
    """the notebook's per-pair comparison for two float framebuffers (H, W, 3): greyscale SSIM and PSNR of their PPM levels"""
    a, b = gray8(ppm_levels(fb_a)), gray8(ppm_levels(fb_b))
    return {"ssim": round(ssim(a, b), 6), "psnr_db": (None if not np.isfinite(psnr(a, b)) else round(psnr(a, b), 3)), "identical": bool(np.array_equal(a, b))}
