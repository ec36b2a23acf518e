#!/usr/bin/env python3
"""Training-trajectory check: the same seeded run (on-device noise synthesis -> U-Net -> L1 -> Adam) under the fp32 product
schemes and bf16; prints the loss every 25 steps and the maximum relative deviation from the fp32-MFMA trajectory."""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import eld_amd
from eld_amd.model import ELDModel
from eld_amd.noise import NoiseModel

def run(scheme, precision, steps=150, B=4, H=256, W=256):
    eld_amd.set_fp32_products(scheme)
    torch.manual_seed(2018); np.random.seed(2018)
    opt = types.SimpleNamespace(precision=precision, gpu_ids=[0], isTrain=True, checkpoints_dir='/tmp/eld_curves', name='c', netG='unet', channels=4,
                                stage_in='raw', stage_out='raw', lr=1e-4, beta1=0.9, wd=0.0, loss='l1', resume=False, no_log=True, chop=False, model='eld_model')
    m = ELDModel(); m.initialize(opt)
    nm = NoiseModel(model='PGRU', include=4)
    m.set_noise_model(nm)
    g = torch.Generator(device='cuda').manual_seed(1)
    losses = []
    for i in range(steps):
        clean = (torch.floor(65535.0 * torch.rand(B, 4, H, W, device='cuda', generator=g) ** 2.2) / 65535.0).contiguous()
        m.set_input({'target': clean, 'params': [nm._sample_params() for _ in range(B)], 'sample_ids': list(range(i * B, (i + 1) * B))}, 'train')
        m.optimize_parameters()
        losses.append(m.get_current_errors()['Pixel'])
    return np.array(losses)

ref = run('mfma', 'fp32')
print('step      ' + ' '.join('%9d' % s for s in range(0, len(ref), 25)))
print('%-9s ' % 'fp32 mfma' + ' '.join('%9.6f' % ref[s] for s in range(0, len(ref), 25)))
for name, scheme, prec in (('bf16x3', 'bf16x3', 'fp32'), ('fp16x2', 'fp16x2', 'fp32'), ('bf16', 'bf16x3', 'bf16')):
    l = run(scheme, prec)
    print('%-9s ' % name + ' '.join('%9.6f' % l[s] for s in range(0, len(l), 25)) + '   max rel dev %.2e' % float(np.max(np.abs(l - ref) / ref)))
eld_amd.set_fp32_products('bf16x3')
