"""Raw -> sRGB ISP: oracle vs the reference-minted golden (CPU), HIP kernel vs oracle and golden (GPU)."""
import os

import numpy as np
import pytest

from oracle import isp_ref as I      # checker only


@pytest.fixture(scope='module')
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, 'isp.npz'))


@pytest.mark.parametrize('case', ['a', 'b', 'c'])
def test_oracle_matches_reference_golden(gold, case):
    out = I.process(gold[case + '_bayer'], gold[case + '_wb'], gold[case + '_ccm'])
    assert np.array_equal(out, gold[case + '_out'])        # same 8-bit codes everywhere on these fixtures


def test_oracle_crf_branch_is_piecewise_linear():
    E = np.linspace(0, 1, 17, dtype=np.float32)
    fs = (E ** 0.5).astype(np.float32)
    x = np.random.RandomState(0).rand(1, 4, 8, 8).astype(np.float32)
    wb = np.ones((1, 4), np.float32); ccm = np.eye(3, dtype=np.float32)[None]
    out = I.process(x, wb, ccm, CRF=(E, fs))
    lin = I.binning(np.clip(x, 0, 1))
    ref = np.clip((np.interp(lin, E, fs).astype(np.float32) * 255).astype(np.int32), 0, 255) / 255.0
    assert np.abs(out - ref).max() <= 1.0 / 255 + 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize('case', ['a', 'b', 'c'])
def test_kernel_matches_reference_golden(eld_lib, gold, case):
    import torch
    from eld_amd.isp import process
    out = process(torch.from_numpy(gold[case + '_bayer']).cuda(), gold[case + '_wb'], gold[case + '_ccm']).cpu().numpy()
    assert np.array_equal(out, gold[case + '_out'])             # bit-exact 8-bit codes (exact step-function table of the gamma quantiser)


@pytest.mark.gpu
def test_kernel_full_frame_and_crf(eld_lib):
    import torch
    from eld_amd.isp import process
    rng = np.random.RandomState(5)
    x = (rng.rand(2, 4, 356, 532) * 1.1).astype(np.float32)
    wb = np.array([[2.0, 1.0, 1.6, 1.0], [1.8, 1.0, 1.4, 1.0]], np.float32)
    ccm = np.stack([np.eye(3) * 1.3 - 0.1, np.eye(3) * 1.6 - 0.2]).astype(np.float32)
    ref = I.process(x, wb, ccm)
    out = process(torch.from_numpy(x).cuda(), wb, ccm).cpu().numpy()
    assert np.array_equal(out, ref)                             # every code of a 2 x 356 x 532 frame
    t = torch.from_numpy(x)                                     # ... and the same codes torch itself produces (reference arithmetic)
    lin = torch.from_numpy(np.clip(I.apply_ccms(I.binning(np.clip(I.apply_gains(x, wb), 0, 1).astype(np.float32)), ccm), 0, 1).astype(np.float32))
    tq = (torch.clamp((torch.clamp(lin, min=1e-8) ** (1 / 2.2) * 255).int(), min=0, max=255).float() / 255).numpy()
    assert (tq != ref).mean() <= 1e-5                           # torch's scalar tail path may differ on the last few elements of the tensor
    out27 = process(torch.from_numpy(x).cuda(), wb, ccm, gamma=2.7).cpu().numpy()          # other gammas: double pow, isolated +-1 codes
    codes = np.rint(np.abs(out27 - I.process(x, wb, ccm, gamma=2.7)) * 255).astype(int)
    assert codes.max() <= 1 and (codes > 0).mean() <= 2e-4
    E = np.linspace(0, 1, 1024, dtype=np.float32)
    fs = (E ** 0.45).astype(np.float32)
    ref = I.process(x, wb, ccm, CRF=(E, fs))
    out = process(torch.from_numpy(x).cuda(), wb, ccm, CRF=(E, fs)).cpu().numpy()
    codes = np.rint(np.abs(out - ref) * 255).astype(int)
    assert codes.max() <= 1 and (codes > 0).mean() <= 2e-4
