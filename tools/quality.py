#!/usr/bin/env python3
"""fp16-vs-fp32 image metrics the way the reference's evaluations.ipynb computes them (cell "calculate_metrics": 8-bit PPM values,
cv2 RGB->gray, skimage structural_similarity with its defaults, cv2.PSNR) — restated with numpy/scipy, for the frames this build
renders on MI355X.  The reference reports SSIM 0.4317 / PSNR 12.4 dB for its fp16 build against its fp32 build.  GPU box only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dd2360-raytracing_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch
import rt_amd as rt
from image_metrics import ppm_levels, gray8, ssim, psnr


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
