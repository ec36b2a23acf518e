"""Run the reference's UNMODIFIED entry script (train_syn.py) with the MI355X plugins installed.

    cd <scratch dir>; python -m eld_amd.launch --ref /path/to/ELD [--plugins noise,arch,model] -- \
        --name run1 --include 4 --noise PGRU --gpu_ids 0 --nThreads 0 --no-log

What it does (nothing in the reference tree is edited or written to):
  1. installs the shims the 2024 reference needs on a 2026 software stack: stub modules for
     rawpy / exifread / tensorboardX / torchinterp1d / skimage / skvideo / cv2 / colour that are only touched when real
     SID data is opened, `torch._utils._accumulate`, a TTY-less `stty size`, and -- when no LMDB is present -- an
     in-memory synthetic stand-in for the `lmdb` module so `LMDBDataset` (dataset/lmdb_dataset.py) works;
  2. installs the plugins at the reference's three registries:
       noise : sys.modules['noise']            = eld_amd.noise        (resolved by `import noise`, train_syn.py:10)
       arch  : models.arch.__dict__['unet']    = eld_amd.unet.unet     (looked up at models/ELD_model.py:391)
       model : models.__dict__['eld_model']    = eld_amd.model.eld_model  (engine.py:26; the fully fused step)
  3. runs `<ref>/train_syn.py` with runpy from a scratch CWD that symlinks the data tables the script opens relatively.
With `--plugins none` the reference runs on its own code (used by the CPU test of the harness itself).
"""
import argparse
import io
import itertools
import os
import pickle
import runpy
import sys
import types

import numpy as np


def install_shims(synthetic_lmdb=True, patches=16, patch_hw=(512, 512)):
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m
    for name in ('rawpy', 'exifread', 'cv2', 'colour', 'torchinterp1d', 'skvideo', 'skvideo.measure', 'skvideo.utils',
                 'tensorboardX', 'skimage', 'skimage.metrics', 'torchvision', 'torchvision.transforms'):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                mod(name)
    sys.modules['torchinterp1d'].__dict__.setdefault('Interp1d', object)
    sys.modules['tensorboardX'].__dict__.setdefault('SummaryWriter', lambda *a, **k: types.SimpleNamespace(add_scalar=lambda *a, **k: None))
    sm = sys.modules['skimage.metrics']
    sm.__dict__.setdefault('peak_signal_noise_ratio', lambda a, b, data_range=255: float(10 * np.log10(data_range ** 2 / np.mean((np.asarray(a, np.float64) - b) ** 2))))
    sm.__dict__.setdefault('structural_similarity', lambda *a, **k: float('nan'))
    sys.modules['skvideo.measure'].__dict__.setdefault('strred', None)
    sys.modules['skvideo.utils'].__dict__.setdefault('rgb2gray', None)
    import torch._utils
    if not hasattr(torch._utils, '_accumulate'):
        torch._utils._accumulate = itertools.accumulate
    _popen = os.popen
    os.popen = lambda cmd, *a, **k: io.StringIO('24 80') if 'stty' in cmd else _popen(cmd, *a, **k)      # util/util.py:185
    if synthetic_lmdb and 'lmdb' not in sys.modules:
        try:
            import lmdb  # noqa: F401
        except Exception:
            rng = np.random.default_rng(2018)
            h, w = patch_hw
            recs = [np.floor(65535.0 * rng.uniform(size=(4, h, w)) ** 2.2).astype(np.uint16).tobytes() for _ in range(patches)]

            class _Env:
                def begin(self, write=False):
                    return self

                def __enter__(self):
                    return self

                def __exit__(self, *a):
                    return False

                def stat(self):
                    return {'entries': len(recs)}

                def get(self, key):
                    return recs[int(key.decode('ascii')) % len(recs)]

            def _open(path, **kw):
                os.makedirs(path, exist_ok=True)
                meta = os.path.join(path, 'meta_info.pkl')
                if not os.path.exists(meta):
                    with open(meta, 'wb') as f:
                        pickle.dump({'shape': (4, h, w), 'dtype': np.uint16}, f)
                return _Env()
            mod('lmdb', open=_open)


def install_plugins(which):
    if 'noise' in which:
        import eld_amd.noise as plug
        sys.modules['noise'] = plug
    if 'arch' in which or 'model' in which:
        import models                                        # the reference package (on sys.path)
        if 'arch' in which:
            import eld_amd.unet as u
            models.arch.__dict__['unet'] = u.unet
        if 'model' in which:
            import eld_amd.model as m
            models.__dict__['eld_model'] = m.eld_model


def prepare_cwd(ref, cwd):
    os.makedirs(cwd, exist_ok=True)
    for name in ('camera_params', 'dataset', 'SID_Sony_15_paired.txt', 'SID_Sony_paired.txt'):
        dst = os.path.join(cwd, name)
        if not os.path.lexists(dst) and name != 'dataset':
            os.symlink(os.path.join(ref, name), dst)
    os.makedirs(os.path.join(cwd, 'dataset'), exist_ok=True)       # train_syn.py:25-27 reads ./dataset/*.txt
    for f in os.listdir(os.path.join(ref, 'dataset')):
        if f.endswith('.txt') and not os.path.lexists(os.path.join(cwd, 'dataset', f)):
            os.symlink(os.path.join(ref, 'dataset', f), os.path.join(cwd, 'dataset', f))
    os.makedirs(os.path.join(cwd, 'checkpoints'), exist_ok=True)
    os.makedirs(os.path.join(cwd, 'data', 'Train'), exist_ok=True)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--ref', required=True, help='path of the (unmodified) reference checkout')
    ap.add_argument('--plugins', default='noise,arch', help='comma list of noise,arch,model or "none"')
    ap.add_argument('--script', default='train_syn.py')
    ap.add_argument('--cwd', default=None, help='scratch working directory (default: $TMPDIR/eld_amd_run)')
    ap.add_argument('rest', nargs=argparse.REMAINDER)
    args = ap.parse_args(argv)
    ref = os.path.abspath(args.ref)
    cwd = args.cwd or os.path.join(os.environ.get('TMPDIR', '/tmp'), 'eld_amd_run')
    prepare_cwd(ref, cwd)
    os.chdir(cwd)
    sys.path.insert(0, ref)
    install_shims()
    which = [] if args.plugins == 'none' else [p.strip() for p in args.plugins.split(',') if p.strip()]
    install_plugins(which)
    rest = args.rest[1:] if args.rest and args.rest[0] == '--' else args.rest
    sys.argv = [os.path.join(ref, args.script)] + rest
    runpy.run_path(sys.argv[0], run_name='__main__')


if __name__ == '__main__':
    main()
