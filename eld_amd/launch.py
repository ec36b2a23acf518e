"""Run the reference's UNMODIFIED entry script (train_syn.py) with the MI355X plugins installed.

    cd <scratch dir>; python -m eld_amd.launch --ref /path/to/ELD [--plugins noise,arch,model] -- \
        --name run1 --include 4 --noise PGRU --gpu_ids 0 --nThreads 0 --no-log

What it does (nothing in the reference tree is edited or written to):
  1. installs the shims the 2024 reference needs on a 2026 software stack: stub modules for
     rawpy / exifread / tensorboardX / torchinterp1d / skimage / skvideo / cv2 / colour that are only touched when real
     SID data is opened, `torch._utils._accumulate`, a TTY-less `stty size`, and -- when no LMDB is present -- an
     in-memory synthetic stand-in for the `lmdb` module so `LMDBDataset` (dataset/lmdb_dataset.py) works;
  2. installs the plugins at the reference's registries:
       noise : sys.modules['noise']            = eld_amd.noise        (resolved by `import noise`, train_syn.py:10)
       arch  : models.arch.__dict__['unet']    = eld_amd.unet.unet     (looked up at models/ELD_model.py:391)
       model : models.__dict__['eld_model']    = eld_amd.model.eld_model  (engine.py:26; the fully fused step).  The model
               attaches the NoiseModel instance the script built (train_syn.py:38) by itself.
       data  : dataset.lmdb_dataset.LMDBDataset, dataset.sid_dataset.{SynDataset, ELDTrainDataset, worker_init_fn}
               = eld_amd.data's deferred-synthesis classes (needs `model`): DataLoader workers (default --nThreads 8) draw the
               per-sample parameters and augmentation bits, the pixels are synthesised on the device in set_input.
               The shipped train_syn.py reads OFFLINE pre-synthesised noise (train_syn.py:66-70; the on-the-fly SynDataset
               lines 61-64 are commented out, although scripts/train.sh:2-4 says the released models were trained on-the-fly):
               with --online-noise (default when the offline database directory does not exist) opening
               `SID_Sony_syn_Raw_<camera>.db` yields SynDataset(LMDBDataset(SID_Sony_Raw.db), noise_maker=<that NoiseModel>),
               i.e. exactly the commented lines, without editing the script.
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


def install_shims(synthetic_lmdb=True, patches=16, patch_hw=(512, 512), entries=None):
    """entries: what the in-memory LMDB stand-in reports as its record count (default: `patches`); record i is patch i % patches, so an
    epoch can have the reference's length (1288 samples, train_syn.py:40) over a handful of synthetic patches."""
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
                    return {'entries': int(entries) if entries else len(recs)}

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


def install_plugins(which, online_noise=None, num_burst=1):
    if 'data' in which and 'model' not in which:
        raise SystemExit("--plugins data needs model: deferred samples are synthesised by eld_amd.model.ELDModel.set_input")
    if 'noise' in which:
        import eld_amd.noise as plug
        sys.modules['noise'] = plug
    if 'data' in which:
        import dataset.lmdb_dataset as ref_lmdb               # the reference modules (on sys.path); train_syn.py resolves the
        import dataset.sid_dataset as ref_sid                 # classes through these module objects at call time
        import eld_amd.data as D
        import eld_amd.noise as N

        class _LMDBDataset(D.LMDBDataset):
            """Redirects the offline-noise database of train_syn.py:66-70 to on-the-fly synthesis (train_syn.py:61-64)."""
            def __new__(cls, db_path, size=None, repeat=1, **kw):
                base = os.path.basename(os.path.normpath(db_path))
                offline = base.startswith('SID_Sony_syn_Raw_') and base.endswith('.db')
                want = online_noise if online_noise is not None else not os.path.exists(os.path.join(db_path, 'meta_info.pkl'))
                if offline and want:
                    nm = N.NoiseModel.last_instance
                    if nm is None:
                        raise RuntimeError('--online-noise: no NoiseModel has been constructed yet (needs the noise plugin)')
                    clean = D.LMDBDataset(os.path.join(os.path.dirname(os.path.normpath(db_path)), 'SID_Sony_Raw.db'), size=size, repeat=repeat)
                    print('[i] eld_amd: %s -> on-device synthesis from SID_Sony_Raw.db with noise model %r' % (base, nm.model))
                    return D.SynDataset(clean, noise_maker=nm, num_burst=num_burst, size=size, repeat=repeat)
                return super().__new__(cls)
        ref_lmdb.LMDBDataset = _LMDBDataset
        ref_sid.SynDataset = D.SynDataset
        ref_sid.ISPDataset = D.ISPDataset                     # --stage_in srgb (train_syn.py:55-58)
        ref_sid.ELDTrainDataset = D.ELDTrainDataset
        ref_sid.worker_init_fn = D.worker_init_fn
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
    ap.add_argument('--plugins', default='noise,arch', help='comma list of noise,arch,model,data or "none"')
    ap.add_argument('--online-noise', dest='online_noise', action='store_true', default=None, help='data plugin: synthesise the input on the fly even if an offline-noise LMDB exists')
    ap.add_argument('--offline-noise', dest='online_noise', action='store_false', help='data plugin: read the offline-noise LMDB (train_syn.py as shipped)')
    ap.add_argument('--num-burst', type=int, default=1, help='data plugin: burst frames per sample (sid_dataset.py:267-273)')
    ap.add_argument('--lmdb-entries', type=int, default=0, help='synthetic LMDB stand-in: reported record count (0 = the 16 synthetic patches); 1288 = an epoch of the reference\'s length')
    ap.add_argument('--script', default='train_syn.py')
    ap.add_argument('--cwd', default=None, help='scratch working directory (default: $TMPDIR/eld_amd_run)')
    ap.add_argument('--stop-after-epochs', type=int, default=0, help='harness: leave the script cleanly after this many Engine.train calls (train_syn.py:100 loops to epoch 200)')
    ap.add_argument('--max-iters-per-epoch', type=int, default=0, help='harness: truncate every epoch to this many batches (an epoch of train_syn.py is always 1288 samples)')
    ap.add_argument('rest', nargs=argparse.REMAINDER)
    args = ap.parse_args(argv)
    ref = os.path.abspath(args.ref)
    cwd = args.cwd or os.path.join(os.environ.get('TMPDIR', '/tmp'), 'eld_amd_run')
    prepare_cwd(ref, cwd)
    os.chdir(cwd)
    sys.path.insert(0, ref)
    install_shims(entries=args.lmdb_entries or None)
    which = [] if args.plugins == 'none' else [p.strip() for p in args.plugins.split(',') if p.strip()]
    install_plugins(which, online_noise=args.online_noise, num_burst=args.num_burst)
    if args.stop_after_epochs > 0 or args.max_iters_per_epoch > 0:
        import engine as ref_engine                        # the reference's engine.py: train_syn.py does `from engine import Engine`
        _train, left = ref_engine.Engine.train, [args.stop_after_epochs]

        class _Head(object):                                # the first K batches of a loader
            def __init__(self, loader, k):
                self.loader, self.k = loader, k

            def __len__(self):
                return min(self.k, len(self.loader))

            def __iter__(self):
                return itertools.islice(iter(self.loader), self.k)

        def train(self, loader, *a, **k):
            if args.max_iters_per_epoch > 0:
                loader = _Head(loader, args.max_iters_per_epoch)
            r = _train(self, loader, *a, **k)
            left[0] -= 1
            if args.stop_after_epochs > 0 and left[0] <= 0:
                print('[i] eld_amd.launch: stopping after %d epoch(s)' % args.stop_after_epochs)
                raise SystemExit(0)
            return r
        ref_engine.Engine.train = train
    rest = args.rest[1:] if args.rest and args.rest[0] == '--' else args.rest
    sys.argv = [os.path.join(ref, args.script)] + rest
    runpy.run_path(sys.argv[0], run_name='__main__')


if __name__ == '__main__':
    main()
