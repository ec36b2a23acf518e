"""GPU: where a workgroup runs is a speed choice only.  Round 6 remaps work-item ids per XCD (conv.h xcd_block, env ELD_XCD) and numbers tiles in bands
(band_tile, env ELD_TILE_BAND).  Both switches are read once per process, so each setting runs in its own interpreter and prints checksums:
  * ELD_XCD=0 against the default: the SAME work items on other workgroups -- the output and every gradient must agree bit for bit, both precisions;
  * ELD_TILE_BAND=1 against the default: the convolutions are the same bits (a tile's result does not depend on its number); the weight gradients add the
    same tiles in another order -- equal to fp32 summation-order accuracy."""
import hashlib
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import hashlib, json, sys
sys.path.insert(0, %r)
import torch
import eld_amd
from eld_amd.unet import UNetSeeInDark
eld_amd.load_library()
res = {}
for prec in ('fp32', 'bf16'):
    torch.manual_seed(5)
    net = UNetSeeInDark(4, 4).cuda()
    shape = (2, 4, 528, 1072)                      # big enough for the tiled kernels of every level (codes, specialised waves, 8-wave weight gradients)
    g = torch.Generator(device='cuda').manual_seed(17)
    x = torch.rand(*shape, device='cuda', generator=g) ** 2.2
    dout = torch.randn(*shape, device='cuda', generator=g) / x.numel()
    out, key, _ = net._engine_forward(x, save=True, bf16=prec == 'bf16')
    grads = net._engine_backward(dout, key, shape).clone()
    torch.cuda.synchronize()
    res[prec] = {'out': hashlib.sha256(out.float().cpu().numpy().tobytes()).hexdigest(), 'grads': hashlib.sha256(grads.cpu().numpy().tobytes()).hexdigest(),
                 'gnorm': float(grads.double().norm()), 'gsum': float(grads.double().sum())}
    torch.save(grads.cpu(), sys.argv[1] + '_' + prec + '.pt')
print('RESULT ' + json.dumps(res))
''' % ROOT


def _run(tmp_path, tag, env):
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, '-c', CHILD, str(tmp_path / tag)], env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith('RESULT ')][-1]
    return json.loads(line[7:])


def test_xcd_aware_ids_change_no_bit(eld_lib, tmp_path):
    a = _run(tmp_path, 'default', {})
    b = _run(tmp_path, 'noxcd', {'ELD_XCD': '0'})
    for prec in ('fp32', 'bf16'):
        assert a[prec]['out'] == b[prec]['out'], prec
        assert a[prec]['grads'] == b[prec]['grads'], prec


def test_banded_tile_order_changes_only_the_summation_order(eld_lib, tmp_path):
    a = _run(tmp_path, 'band4', {})
    b = _run(tmp_path, 'band1', {'ELD_TILE_BAND': '1'})
    for prec, tol in (('fp32', 2e-6), ('bf16', 2e-6)):
        assert a[prec]['out'] == b[prec]['out'], prec                     # convolutions: a tile's result does not depend on its number
        ga, gb = torch.load(tmp_path / ('band4_%s.pt' % prec)), torch.load(tmp_path / ('band1_%s.pt' % prec))
        rel = float((ga.double() - gb.double()).norm() / ga.double().norm())
        assert rel < tol, (prec, rel)                                       # weight gradients: the same partial products, added in another order
