"""CPU oracle for the evaluation metrics -- TEST INFRASTRUCTURE ONLY.

The reference calls scikit-image (util/index.py:2,79-80: peak_signal_noise_ratio, structural_similarity(data_range=255,
multichannel=True)).  scikit-image is a third-party dependency that is ABSENT here and unpinned by the reference
(no requirements file), so parity for SSIM is "unpinned": this file restates the published algorithm (Wang et al. 2004) with
skimage.metrics.structural_similarity's documented defaults -- win_size 7, uniform filter, K1=0.01, K2=0.03,
use_sample_covariance=True, borders of (win_size-1)//2 cropped before the mean, channels averaged -- on NumPy +
scipy.ndimage.uniform_filter (which is what skimage itself filters with)."""
import numpy as np
from scipy.ndimage import uniform_filter


def psnr(true, test, data_range=255.0):
    err = np.mean((np.asarray(true, np.float64) - np.asarray(test, np.float64)) ** 2)
    return 10 * np.log10(data_range ** 2 / err)


def ssim_channel(x, y, data_range=255.0, win_size=7, K1=0.01, K2=0.03):
    x, y = np.asarray(x, np.float64), np.asarray(y, np.float64)
    NP = win_size ** 2
    cov_norm = NP / (NP - 1.0)
    f = lambda t: uniform_filter(t, size=win_size)
    ux, uy = f(x), f(y)
    uxx, uyy, uxy = f(x * x), f(y * y), f(x * y)
    vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
    C1, C2 = (K1 * data_range) ** 2, (K2 * data_range) ** 2
    S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux ** 2 + uy ** 2 + C1) * (vx + vy + C2))
    pad = (win_size - 1) // 2
    return S[pad:-pad, pad:-pad].mean()


def ssim(x, y, data_range=255.0):
    """x, y: (C,H,W); multichannel=True -> mean of the per-channel SSIMs."""
    return float(np.mean([ssim_channel(x[c], y[c], data_range) for c in range(x.shape[0])]))
