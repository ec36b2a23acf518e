#!/usr/bin/env python3
"""Mint tests/golden/eval.npz by RUNNING THE REFERENCE's evaluation helpers themselves (models/ELD_model.py):
    tensor2im            :23-38    x255, clip to [0,255], float32, HWC, not rounded
    IlluminanceCorrect   :138-169  per-image least-squares gain over source != 1 (float32 torch.dot)
    ELDModel.forward_chop :434-467      four overlapping quadrants, shave >= 10 rounded up to a multiple of 16
TEST INFRASTRUCTURE ONLY.      python oracle/gen_golden_eval.py [--ref /root/reference]

`import models.ELD_model` needs the SURVEY App. D shims (absent third-party modules, torch._utils._accumulate, the `stty` call of
util/util.py:185); they are the ones oracle/gen_golden.py installs.  forward_chop is called as the unbound reference method on a
stand-in whose only attribute is `netG` = the reference's UNetSeeInDark under torch.manual_seed(2018) default init (31 MB of weights
are not committed: the fixture carries the parameter checksum `wsum` instead; the same torch build on the GPU box reproduces the
init, and tests skip with a message if it does not).

Cases
  ic_n_*      IlluminanceCorrect, batch 3 with per-image sources (a saturated-pixel mask in each)
  ic_one_*    batch 3 against ONE source frame (source.shape[0] == 1 branch, :148-150)
  ic_b1_*     batch 1 (the `else` branch, :151-152), source containing pixels exactly equal to 1 and predict values outside [0,1]
  t2i_*       tensor2im of a batch (only image 0 is returned, :31-32), values below 0 and above 1; t2i_out1 = tensor2im(batch[1:])
  chop_a_*    1x4x64x96: h_half 32 -> shave 0 -> +16 -> 48; w_half 48 -> shave 0 -> +16 -> 64      (shave < 10 branch)
  chop_b_*    1x4x44x74: h_half 22 -> shave 10 -> 32;       w_half 37 -> shave 11 -> 48            (shave >= 10 branch)
"""
import argparse
import io
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, 'tests', 'golden')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--ref', default='/root/reference')
    ref = os.path.abspath(ap.parse_args().ref)
    os.chdir(ref)
    sys.path.insert(0, ref)
    sys.path.insert(0, ROOT)
    from oracle.gen_golden import install_stubs
    install_stubs()
    for mod, names in (('skimage.metrics', ('structural_similarity', 'peak_signal_noise_ratio')), ('skvideo.measure', ('strred',)),
                       ('skvideo.utils', ('rgb2gray',))):       # names util/index.py:2-6 imports; never called here
        for n in names:
            sys.modules[mod].__dict__.setdefault(n, lambda *a, **k: float("nan"))
    _popen = os.popen
    os.popen = lambda cmd, *a, **k: io.StringIO('24 80') if 'stty' in cmd else _popen(cmd, *a, **k)     # App. D item 3
    import torch
    import models.ELD_model as ref_model                        # <- the reference module
    os.popen = _popen
    torch.set_num_threads(1)                                     # torch.dot's float32 summation order is thread-count dependent

    out = {}
    g = torch.Generator().manual_seed(41)
    ic = ref_model.IlluminanceCorrect()

    def pair(shape, one):
        pred = torch.rand(*shape, generator=g) * 1.4 - 0.2       # outside [0,1] on both sides: the clamp matters
        src = torch.rand(*(((1,) + shape[1:]) if one else shape), generator=g)
        src[src > 0.9] = 1.0                                     # saturated pixels leave the fit (source != 1)
        return pred, src
    for tag, shape, one in (('ic_n', (3, 4, 24, 40), False), ('ic_one', (3, 4, 16, 16), True), ('ic_b1', (1, 4, 37, 41), False)):
        pred, src = pair(shape, one)
        with torch.no_grad():
            res = ic(pred, src)                                  # <- the reference
        out[tag + '_pred'], out[tag + '_src'], out[tag + '_out'] = pred.numpy(), src.numpy(), res.numpy()

    x = torch.rand(2, 4, 9, 13, generator=g) * 1.3 - 0.15
    out['t2i_in'] = x.numpy()
    out['t2i_out'] = ref_model.tensor2im(x)                      # <- the reference: (9,13,4) float32 of image 0
    out['t2i_out1'] = ref_model.tensor2im(x[1:])                 # image 1 the same way (the pair feeds the fused PSNR/SSIM kernel test)
    assert out['t2i_out'].dtype == np.float32 and out['t2i_out'].shape == (9, 13, 4)

    torch.manual_seed(2018)
    net = ref_model.arch.__dict__['unet'](4, 4)                  # models/arch/__init__.py:6-7, default init (ELD_model.py:391-393)
    out['wsum'] = np.array([float(p.detach().double().sum()) for p in net.parameters()])
    holder = types.SimpleNamespace(netG=net)
    for tag, shape in (('chop_a', (1, 4, 64, 96)), ('chop_b', (1, 4, 44, 74))):
        x = torch.rand(*shape, generator=g)
        with torch.no_grad():
            y = ref_model.ELDModel.forward_chop(holder, x)   # <- the reference method, unbound
        out[tag + '_x'], out[tag + '_out'] = x.numpy(), y.numpy()
    np.savez_compressed(os.path.join(GOLD, 'eval.npz'), torch_version=torch.__version__, **out)
    print('wrote eval.npz:', {k: getattr(v, 'shape', v) for k, v in out.items()})


if __name__ == '__main__':
    main()
