"""CPU test: the U-Net oracle reproduces the reference module (golden minted from models/arch/Unet.py)."""
import os

import numpy as np
import torch

from oracle import unet_ref as U


def test_unet_oracle_matches_reference_golden(golden_dir):
    d = np.load(os.path.join(golden_dir, 'unet.npz'))
    sd = U.seeded_state_dict(4, 4, seed=2018)
    names = [str(n) for n in d['names']]
    assert names == list(sd.keys())
    wsum = np.array([float(sd[n].double().sum()) for n in names])
    if str(d['torch_version']) == torch.__version__:
        assert np.array_equal(wsum, d['wsum'])          # same seeded default init as the reference module
    x, t = torch.from_numpy(d['x']), torch.from_numpy(d['t'])
    torch.set_num_threads(max(1, torch.get_num_threads()))
    out, loss, grads = U.loss_and_grads(sd, x, t)
    assert np.max(np.abs(out.numpy() - d['out'])) < 1e-6
    assert abs(loss - float(d['loss'])) < 1e-6
    gsum = np.array([float(grads[n].double().sum()) for n in names])
    gabs = np.array([float(grads[n].double().abs().sum()) for n in names])
    assert np.allclose(gsum, d['gsum'], rtol=1e-4, atol=1e-6) and np.allclose(gabs, d['gabs'], rtol=1e-4, atol=1e-7)
    for k in ('conv1_1.weight', 'conv10_1.weight', 'upv9.bias'):
        assert np.allclose(grads[k].numpy(), d['grad_' + k.replace('.', '__')], rtol=1e-4, atol=1e-7)


def test_lrelu_tie_and_pool_tie_semantics():
    """Autograd conventions the HIP backward mirrors: lrelu'(0) = 0.6, max-pool ties -> first."""
    x = torch.zeros(3, requires_grad=True)
    U.lrelu(x).sum().backward()
    assert torch.allclose(x.grad, torch.full((3,), 0.6))
    p = torch.ones(1, 1, 2, 2, requires_grad=True)
    torch.nn.functional.max_pool2d(p, 2).sum().backward()
    assert p.grad.reshape(-1).tolist() == [1.0, 0.0, 0.0, 0.0]
