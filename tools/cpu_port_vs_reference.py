#!/usr/bin/env python3
"""Same-container timing of the oracle PORT (what bench.py's cpu_baseline times on the GPU box, where /root/reference does not exist)
against the REFERENCE ITSELF (noise.py:149-170, models/arch/Unet.py + nn.L1Loss + torch.optim.Adam), on identical inputs.
TEST / MEASUREMENT INFRASTRUCTURE (imports oracle/ and /root/reference).   python tools/cpu_port_vs_reference.py [--ref /root/reference]"""
import argparse
import importlib.util
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def best(fn, n=3):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return min(ts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--ref', default='/root/reference')
    ap.add_argument('--out', default=os.path.join(ROOT, 'profiles', 'r04_cpu_port_vs_reference.md'))
    a = ap.parse_args()
    ref = os.path.abspath(a.ref)
    from oracle import noise_ref as O
    from oracle import unet_ref as U
    os.chdir(ref); sys.path.insert(0, ref)
    import noise as ref_noise
    spec = importlib.util.spec_from_file_location('ref_unet', os.path.join(ref, 'models', 'arch', 'Unet.py'))
    ref_unet = importlib.util.module_from_spec(spec); spec.loader.exec_module(ref_unet)
    rows = []
    rs = np.random.RandomState(2018)
    for shape in ((4, 512, 512), (4, 1424, 2128)):
        y = (np.floor(65535.0 * rs.uniform(size=shape) ** 2.2) / 65535.0).astype(np.float32)
        params = (np.float32(2.288), np.float32(6.451), 15583, np.float32(208.98))
        for model in ('Pg', 'pg', 'g'):
            nm = ref_noise.NoiseModel(model=model, include=4)
            np.random.seed(1); zr = nm(y, params=params)
            np.random.seed(1); zo, _ = O.noise_numpy_rng(y, model, params)
            assert np.array_equal(zr, zo), (model, shape)                 # same bits: the port IS the reference's arithmetic
            t_ref = best(lambda: nm(y, params=params))
            t_port = best(lambda: O.noise_numpy_rng(y, model, params))
            rows.append(('sampler %s %dx%dx%d' % ((model,) + shape), t_ref, t_port, y.size))
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    for hw in ((512, 512),):
        torch.manual_seed(2018)
        net = ref_unet.UNetSeeInDark(4, 4)
        opt_r = torch.optim.Adam(net.parameters(), lr=1e-4, betas=(0.9, 0.999))
        sd = U.seeded_state_dict(4, 4, seed=2018)
        pr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        opt_p = torch.optim.Adam(list(pr.values()), lr=1e-4)
        g = torch.Generator().manual_seed(3)
        x = torch.rand(1, 4, *hw, generator=g); t = torch.rand(1, 4, *hw, generator=g)

        def step_ref():
            out = net(x); loss = torch.nn.L1Loss()(out, t); opt_r.zero_grad(); loss.backward(); opt_r.step(); return loss.item()

        def step_port():
            loss = torch.nn.functional.l1_loss(U.unet_forward(pr, x), t); opt_p.zero_grad(); loss.backward(); opt_p.step(); return loss.item()
        l_r, l_p = step_ref(), step_port()
        assert abs(l_r - l_p) < 1e-6, (l_r, l_p)
        rows.append(('U-Net train step (fwd, L1, bwd, Adam, loss.item()) 1x4x%dx%d, %d threads' % (hw + (torch.get_num_threads(),)), best(step_ref, 2), best(step_port, 2), 4 * hw[0] * hw[1]))
    lines = ['# r04: the oracle PORT timed against the REFERENCE ITSELF in the build container (%d host cores)' % (os.cpu_count() or 1), '',
             '`python tools/cpu_port_vs_reference.py`: `bench.py`\'s `cpu_baseline` is `kind: "port"` because `/root/reference` does not exist on the GPU box; this table',
             'shows, where both can run, that the port costs what the reference costs (identical inputs, identical NumPy seed: the sampler outputs are asserted bit-equal,',
             'the first training loss equal to 1e-6).  Minimum of 3 (sampler) / 2 (U-Net) runs.', '',
             '| workload | reference (s) | port (s) | port / reference | reference raw MPix/s |', '|---|---|---|---|---|']
    for name, tr, tp, n in rows:
        lines.append('| %s | %.3f | %.3f | %.2f | %.2f |' % (name, tr, tp, tp / tr, n / tr / 1e6))
    open(a.out, 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(lines))


if __name__ == '__main__':
    main()
