"""Sampler-only timing (HIP events on the launch stream) at BASELINE.json config-2 size; dev tool."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from eld_amd import _lib as L
if os.environ.get('ELD_DEV_LIB'):          # tools/build_dev.sh: library with the ablation switches (ELD_NOISE_DBG) compiled in
    L.LIB_PATH = os.environ['ELD_DEV_LIB']
from eld_amd.noise import NoiseParams, sample_noise, model_flags

def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    H, W = 1424, 2128
    g = torch.Generator(device='cuda').manual_seed(0)
    y = (torch.rand(N, 4, H, W, device='cuda', generator=g) ** 2.2 * 65535).floor() / 65535
    out = torch.empty_like(y)
    for model, p in [('PGRU', NoiseParams(2.288, 6.451, 15583, 208.98, tl_lambda=-0.14285714, tl_scale=3.3, row_scale=0.9)),
                     ('PGRU', NoiseParams(0.1, 0.7, 15583, 100.0, tl_lambda=-0.14285714, tl_scale=3.3, row_scale=0.9)),
                     ('GRU', NoiseParams(2.288, 6.451, 15583, 208.98, tl_lambda=-0.14285714, tl_scale=3.3, row_scale=0.9)),
                     ('Pg', NoiseParams(2.288, 6.451, 15583, 208.98)), ('pg', NoiseParams(2.288, 6.451, 15583, 208.98)),
                     ('g', NoiseParams(2.288, 6.451, 15583, 208.98)), ('', NoiseParams(2.288, 6.451, 15583, 208.98))]:
        fl = model_flags(model)
        for _ in range(3):
            sample_noise(y, [p] * N, fl, 2018, list(range(N)), out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            sample_noise(y, [p] * N, fl, 2018, list(range(N)), out=out)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        byt = 8.0 * y.numel()
        print('model %-5s K=%.3g N=%d  %.3f ms  %.1f GB/s (%.1f%% of 8 TB/s)  %.0f MPix/s' % (
            model, p[0], N, ms, byt / ms / 1e6, byt / ms / 1e6 / 80, y.numel() / ms / 1e3))

main()
