"""Evaluation metrics on device: what the reference's eval path computes with scikit-image on the host
(util/index.py:76-81 `quality_assess`: PSNR and multichannel SSIM with data_range=255 on the x255-clipped float
images of `tensor2im`, models/ELD_model.py:23-38).  Everything here runs the HIP kernels of csrc/eval.hip (one pass over the
frames, double accumulation, no host round trip); host arrays handed to `quality_assess` are moved to the GPU -- there is no second
implementation in the product.
scikit-image is absent from this stack, so SSIM restates skimage.metrics.structural_similarity's defaults
(7x7 uniform window, K1=0.01, K2=0.03, sample covariance, border of 3 cropped, mean over channels) -- the oracle
statement is oracle/metrics_ref.py (NumPy + scipy.ndimage)."""
import torch


def quality_assess(X, Y, data_range=255.0):
    """util/index.py:76-81 for one image pair already on the [0, data_range] scale (what tensor2im returns), (C,H,W) or (H,W,C)
    as skimage takes it.  Host tensors / ndarrays are moved to the current GPU (the reference's callers hold host arrays; the
    arithmetic still runs in csrc/eval.hip -- there is no CPU implementation in the product).  The kernel's x255 + clip stage
    (tensor2im) is switched off for this entry point by scale = 1: the values are used as given, as util/index.py does."""
    import numpy as np
    dev = torch.device('cuda', torch.cuda.current_device())
    X = torch.as_tensor(np.ascontiguousarray(X) if isinstance(X, np.ndarray) else X).to(dev)
    Y = torch.as_tensor(np.ascontiguousarray(Y) if isinstance(Y, np.ndarray) else Y).to(dev)
    if X.dim() == 3 and X.shape[-1] <= 4 < X.shape[0]:        # HWC (tensor2im's layout) -> CHW
        X, Y = X.permute(2, 0, 1), Y.permute(2, 0, 1)
    q = quality_assess_frames(X.float()[None], Y.float()[None], data_range, scale=1.0)[0].tolist()
    return {'PSNR': q[0], 'SSIM': q[1]}


def quality_assess_frames(est, ref, data_range=255.0, scale=None):
    """util/index.py:76-81 per image, fused with tensor2im (ELD_model.py:23-38): est/ref are CUDA (N,C,H,W) float32 tensors in
    [0,1] units; returns a CUDA float64 tensor (N,2) = [PSNR, SSIM] per image (csrc/eval.hip eld_quality_assess).
    scale=1.0: est/ref are already images on the [0, data_range] scale (eld_quality_assess_images: no x255 stage)."""
    from . import _lib as L
    assert est.is_cuda and ref.is_cuda and est.shape == ref.shape and est.dim() == 4
    est, ref = est.contiguous().float(), ref.contiguous().float()
    N, C, H, W = est.shape
    lib = L.lib()
    ws = torch.empty(lib.eld_quality_assess_workspace_bytes(N, C, H, W), dtype=torch.uint8, device=est.device)
    out = torch.empty((N, 2), dtype=torch.float64, device=est.device)
    if scale is not None and float(scale) != 1.0:
        raise ValueError('scale must be None ([0,1] inputs, tensor2im fused) or 1.0 (images on the data_range scale)')
    fn = lib.eld_quality_assess if scale is None else lib.eld_quality_assess_images
    L.check(fn(L.dptr(est), L.dptr(ref), L.dptr(out), L.dptr(ws), ws.numel(), N, C, H, W, float(data_range), L.cur_stream()), 'eld_quality_assess')
    return out


def illuminance_correct(predict, source):
    """models/ELD_model.py:138-169 on the device (csrc/eval.hip): one least-squares gain per image over source != 1."""
    from . import _lib as L
    assert predict.is_cuda and source.is_cuda and predict.dim() == 4
    predict, source = predict.contiguous().float(), source.contiguous().float()
    N = predict.shape[0]
    assert source.shape[0] in (1, N) and source.shape[1:] == predict.shape[1:]
    lib = L.lib()
    ws = torch.empty(lib.eld_illuminance_correct_workspace_bytes(N), dtype=torch.uint8, device=predict.device)
    out = torch.empty_like(predict)
    L.check(lib.eld_illuminance_correct(L.dptr(predict), L.dptr(source), L.dptr(out), L.dptr(ws), ws.numel(), N, source.shape[0],
                                        predict[0].numel(), L.cur_stream()), 'eld_illuminance_correct')
    return out
