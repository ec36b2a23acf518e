"""CPU oracle for the evaluation metrics -- TEST INFRASTRUCTURE ONLY.

tensor2im and illuminance_correct are pinned by tests/golden/eval.npz, which oracle/gen_golden_eval.py mints by running the
reference's models/ELD_model.py (tensor2im :23-38, IlluminanceCorrect :138-169); forward_chop's restatement is
oracle/unet_ref.py::forward_chop, pinned by the same file.  PSNR / SSIM stay "parity unpinned":

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


def tensor2im(x):
    """models/ELD_model.py:23-38 for one (C,H,W) image in [0,1] units: x*255 clipped to [0,255], float32, not rounded
    (the transpose to HWC does not change the metrics).  PINNED by tests/golden/eval.npz (t2i_*: the reference's own output)."""
    return np.clip(np.asarray(x, np.float32) * np.float32(255.0), 0, 255)


def illuminance_correct(predict, source):
    """models/ELD_model.py:138-169 per image: p = clip(predict,0,1); alpha = <p,s>/<p,p> over source != 1 (float32 ratio of the
    two dot products, accumulated here in float64 and rounded once: the reference's float32 torch.dot differs from that by its
    summation-order rounding, a few ulp of alpha); out = alpha * p.  predict: (N,C,H,W); source: (N or 1,C,H,W).
    PINNED by tests/golden/eval.npz (minted by the reference's IlluminanceCorrect, oracle/gen_golden_eval.py): equal to the
    last bit on its three cases, tests/test_oracle_golden.py::test_eval_oracle_vs_reference_fixture."""
    predict, source = np.asarray(predict, np.float32), np.asarray(source, np.float32)
    out = np.empty_like(predict)
    for i in range(predict.shape[0]):
        p = np.clip(predict[i], 0, 1)
        s = source[i] if source.shape[0] != 1 else source[0]
        m = s != 1
        num = np.dot(p[m].astype(np.float64), s[m].astype(np.float64))
        den = np.dot(p[m].astype(np.float64), p[m].astype(np.float64))
        out[i] = np.float32(num) / np.float32(den) * p
    return out
