"""CPU test: eld_amd.metrics (torch) against the NumPy/scipy oracle restatement of the skimage calls in util/index.py:76-81."""
import numpy as np
import torch

from eld_amd.metrics import quality_assess
from oracle import metrics_ref as M


def test_psnr_ssim_match_oracle():
    g = torch.Generator().manual_seed(3)
    for shape, noise in (((4, 40, 56), 8.0), ((3, 31, 29), 40.0)):
        y = torch.rand(*shape, generator=g) * 255
        x = torch.clamp(y + noise * torch.randn(*shape, generator=g), 0, 255)
        r = quality_assess(x, y)
        assert abs(r['PSNR'] - M.psnr(y.numpy(), x.numpy())) < 1e-9
        assert abs(r['SSIM'] - M.ssim(y.numpy(), x.numpy())) < 1e-10
