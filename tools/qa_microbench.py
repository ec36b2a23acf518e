#!/usr/bin/env python3
"""dev: time the evaluation kernels (eld_quality_assess variants via eld_debug_kernel_mask: 0 = strip kernel, two columns per lane; 32 = one column;
64 = the round-2 tile kernels; eld_illuminance_correct) on SonyA7S2 frames, HIP events, 1 and 8 frames per launch."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import eld_amd
lib = eld_amd.load_library()
from eld_amd.metrics import illuminance_correct, quality_assess_frames

def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

out = {}
for B in (1, 8):
    g = torch.Generator(device='cuda').manual_seed(B)
    ref = torch.rand(B, 4, 1424, 2128, device='cuda', generator=g)
    est = (ref + 0.05 * torch.randn(B, 4, 1424, 2128, device='cuda', generator=g)).contiguous()
    npx = ref.numel()
    for mask, name in ((0, 'strip2'), (32, 'strip1'), (64, 'tiles')):
        prev = lib.eld_debug_kernel_mask(mask)
        try:
            ms = timed(lambda: quality_assess_frames(est, ref))
            q = quality_assess_frames(est, ref)[0].tolist()
        finally:
            lib.eld_debug_kernel_mask(prev)
        out['qa_%s_B%d' % (name, B)] = {'ms_per_frame': round(ms / B, 4), 'GBps': round(8.0 * npx / (ms * 1e-3) / 1e9, 1), 'frac_hbm': round(8.0 * npx / (ms * 1e-3) / 8e12, 4), 'psnr_ssim': q}
    ms = timed(lambda: illuminance_correct(est, ref))
    out['illum_B%d' % B] = {'ms_per_frame': round(ms / B, 4), 'GBps': round(16.0 * npx / (ms * 1e-3) / 1e9, 1), 'frac_hbm': round(16.0 * npx / (ms * 1e-3) / 8e12, 4)}
print(json.dumps(out, indent=1))
