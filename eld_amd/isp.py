"""Raw -> sRGB ISP on the device: the reference's util/process.py:52-68 `process` (gains, binning, CCM, gamma / camera
response, 8-bit quantisation) as one HBM-bound HIP kernel (csrc/eval.hip eld_isp_process).  Used by the sRGB training /
evaluation stages (train_syn.py:55-58, models/ELD_model.py:230-233)."""
import numpy as np

from . import _lib as L


def process(bayer_images, wbs, cam2rgbs, gamma=2.2, CRF=None):
    """Same signature and semantics as util/process.py:52-68.  bayer_images: CUDA (N,4,H,W) float32 RGBG; wbs (N,4);
    cam2rgbs (N,3,3); CRF: None or (E, fs) 1-D tensors/arrays (ascending E).  Returns CUDA (N,3,H,W) float32."""
    import torch
    assert bayer_images.is_cuda and bayer_images.dim() == 4 and bayer_images.shape[1] == 4
    dev = bayer_images.device
    x = bayer_images.contiguous().float()
    N, _, H, W = x.shape
    wbs = torch.as_tensor(wbs, dtype=torch.float32, device=dev).reshape(N, 4).contiguous()
    ccm = torch.as_tensor(cam2rgbs, dtype=torch.float32, device=dev).reshape(N, 9).contiguous()
    out = torch.empty((N, 3, H, W), dtype=torch.float32, device=dev)
    E = fs = None
    n = 0
    if CRF is not None:
        E = torch.as_tensor(np.asarray(CRF[0].cpu() if hasattr(CRF[0], 'cpu') else CRF[0]), dtype=torch.float32, device=dev).contiguous()
        fs = torch.as_tensor(np.asarray(CRF[1].cpu() if hasattr(CRF[1], 'cpu') else CRF[1]), dtype=torch.float32, device=dev).contiguous()
        n = int(E.numel())
        assert fs.numel() == n and n >= 2
    L.check(L.lib().eld_isp_process(L.dptr(x), L.dptr(wbs), L.dptr(ccm), L.dptr(out), N, H, W, float(gamma), L.dptr(E), L.dptr(fs), n,
                                    L.cur_stream()), 'eld_isp_process')
    return out
