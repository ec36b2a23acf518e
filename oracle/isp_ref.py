"""CPU oracle for the raw -> sRGB ISP -- TEST INFRASTRUCTURE ONLY (tests/, smoke, bench cpu_baseline).

Restates util/process.py:15-68 of the reference (`process`: white-balance gains, clamp, RGBG -> RGB binning, 3x3 colour
correction, clamp, gamma compression or camera-response interpolation, 8-bit quantisation by truncation) in NumPy float32,
operation by operation.  Pinned by tests/golden/isp.npz, minted by running the reference function itself
(oracle/gen_golden.py).  Floating point up to the quantiser: a last-bit difference in powf becomes a +-1 code difference
on isolated pixels after the truncating quantiser -- the tests count them.  The CRF branch calls torchinterp1d (third-party, absent here and unpinned by the reference): its published
algorithm (piecewise-linear, end segments extended) is restated; parity for that branch is unpinned."""
import numpy as np


def apply_gains(bayer, wbs):                                   # process.py:15-19
    return bayer * wbs.reshape(wbs.shape[0], wbs.shape[1], 1, 1)


def binning(bayer):                                            # process.py:42-49: R, mean(G1, G2), B of RGBG
    g = (bayer[:, 1] + bayer[:, 3]) / np.float32(2)            # torch.mean over 2 elements
    return np.stack([bayer[:, 0], g, bayer[:, 2]], axis=1)


def apply_ccms(images, ccms):                                  # process.py:22-31: out[c] = sum_j images[j] * ccm[c][j], j ascending
    out = np.empty_like(images)
    for c in range(3):
        acc = images[:, 0] * ccms[:, c, 0].reshape(-1, 1, 1)
        acc = acc + images[:, 1] * ccms[:, c, 1].reshape(-1, 1, 1)
        acc = acc + images[:, 2] * ccms[:, c, 2].reshape(-1, 1, 1)
        out[:, c] = acc
    return out


def quantise(x):                                               # clamp((x*255).int(), 0, 255).float() / 255  (truncation)
    q = np.clip((x * np.float32(255)).astype(np.int32), 0, 255)
    return q.astype(np.float32) / np.float32(255)


_T22 = None


def gamma_compression(images, gamma=2.2):                      # process.py:34-39
    """clamp((clamp(x, 1e-8) ** (1/gamma) * 255).int(), 0, 255) / 255 as torch evaluates it on float32 tensors.  For gamma = 2.2
    (the only value the reference uses) that composition is a monotone step function of x with 255 thresholds, minted
    exhaustively from torch itself by oracle/gen_gamma_table.py (tests/golden/gamma22_thresholds.npy): exact codes.  Other
    gammas: the power in float64, rounded once (torch's vectorised powf is within 1 ulp of that)."""
    global _T22
    x = np.maximum(images, np.float32(1e-8)).astype(np.float32)
    if np.float32(gamma) == np.float32(2.2):
        if _T22 is None:
            import os
            _T22 = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'gamma22_thresholds.npy'))
        q = np.searchsorted(_T22[1:], x.view(np.uint32), side='right')
        return q.astype(np.float32) / np.float32(255)
    outs = np.power(x.astype(np.float64), np.float64(np.float32(1.0 / gamma))).astype(np.float32)
    return quantise(outs)


def camera_response_function(images, E, fs):                   # process.py:71-83 with torchinterp1d's piecewise-linear rule
    E, fs = np.asarray(E, np.float32), np.asarray(fs, np.float32)
    ind = np.clip(np.searchsorted(E, images, side='left') - 1, 0, len(E) - 2)
    slope = (fs[ind + 1] - fs[ind]) / (E[ind + 1] - E[ind])
    return quantise(fs[ind] + slope * (images - E[ind]))


def process(bayer, wbs, cam2rgbs, gamma=2.2, CRF=None):        # process.py:52-68
    bayer = np.asarray(bayer, np.float32)
    wbs, cam2rgbs = np.asarray(wbs, np.float32), np.asarray(cam2rgbs, np.float32)
    x = np.clip(apply_gains(bayer, wbs), 0.0, 1.0).astype(np.float32)
    img = np.clip(apply_ccms(binning(x), cam2rgbs), 0.0, 1.0).astype(np.float32)
    if CRF is None:
        return gamma_compression(img, gamma)
    return camera_response_function(img, CRF[0], CRF[1])
