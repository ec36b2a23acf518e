#!/usr/bin/env python3
"""Mint golden vectors by RUNNING THE REFERENCE ITSELF (python, importable in the build
container from /root/reference).  Outputs small fixtures under tests/golden/ that travel to
the GPU box, where /root/reference does not exist.  TEST INFRASTRUCTURE ONLY.

    python oracle/gen_golden.py [--ref /root/reference]

The reference has no tests / golden vectors of its own, so these are the pins:
  noise_*.npz       NoiseModelBase.__call__ (noise.py:149-170) on seeded inputs, with the NumPy
                    draws it consumed (replayed in the reference's own order) stored beside the
                    output so they can be re-injected into the oracle and the HIP kernel.
  sample_params.npz NoiseModel._sample_params (noise.py:201-225) under fixed seeds.
  rawpacker.npz     RawPacker.pack_raw_bayer / unpack_raw_bayer (noise.py:10-20,66-81).
  lmdb_decode.npz   LMDBDataset.__getitem__ uint16 decode (dataset/lmdb_dataset.py:28-41) for
                    all 65,536 codes, through an in-memory stand-in for the `lmdb` module.
  augment.npz       ELDTrainDataset.__getitem__ flips/transpose (dataset/sid_dataset.py:332-363).
  unet.npz          UNetSeeInDark (models/arch/Unet.py:6-104) seeded forward / L1 backward.
  camera_params.json  restatement of camera_params/release/*.npy (pickled dicts -> JSON) so the
                    plugin has the calibrated tables where the reference tree is absent.

Scalar params are passed to the reference as np.float32 so that it evaluates in float32, which is
what it does under the NumPy-1.x casting rules it was written for.
"""
import argparse
import io
import itertools
import json
import os
import pickle
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, 'tests', 'golden')
CAMERAS = ['CanonEOS5D4', 'CanonEOS70D', 'CanonEOS700D', 'NikonD850', 'SonyA7S2']


def synth_clean(rng, shape):
    """LMDB-like clean patch: uint16 grid, dark-heavy."""
    u16 = np.floor(65535.0 * rng.uniform(0, 1, size=shape) ** 2.2).astype(np.uint16)
    return (u16 / 65535).astype(np.float32)


def install_stubs():
    """modules the reference imports that are absent here."""
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m
    for name in ('rawpy', 'exifread', 'cv2', 'colour', 'torchinterp1d', 'skvideo', 'skvideo.measure',
                 'skvideo.utils', 'tensorboardX', 'skimage', 'skimage.metrics', 'torchvision',
                 'torchvision.transforms', 'scipy.io'):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                mod(name)
    sys.modules['torchinterp1d'].Interp1d = object
    sys.modules['tensorboardX'].SummaryWriter = object
    import torch._utils
    if not hasattr(torch._utils, '_accumulate'):
        torch._utils._accumulate = itertools.accumulate


class FakeLMDB:
    """In-memory stand-in for the `lmdb` module (open/begin/stat/get) -- App. D item 5."""
    def __init__(self, records):
        self.records = records

    def open(self, path, **kw):
        return self

    def begin(self, write=False):
        return self

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def stat(self):
        return {'entries': len(self.records)}

    def get(self, key):
        return self.records[int(key.decode('ascii'))]


def mint_isp():
    """tests/golden/isp.npz: util/process.py:52-68 `process` (gamma branch) of the reference on seeded RGBG batches."""
    import torch
    import util.process as ref_process
    rng = np.random.RandomState(77)
    cases = {}
    for name, shape in (('a', (2, 4, 24, 40)), ('b', (1, 4, 7, 129)), ('c', (3, 4, 16, 16))):
        bayer = (rng.rand(*shape) ** 2.0 * 1.2 - 0.05).astype(np.float32)          # some values outside [0,1]
        wb = np.stack([np.array([rng.uniform(1.5, 2.5), 1.0, rng.uniform(1.2, 2.0), 1.0], np.float32) for _ in range(shape[0])])
        ccm = np.stack([(np.eye(3) + rng.uniform(-0.3, 0.3, (3, 3))).astype(np.float32) for _ in range(shape[0])])
        ccm /= ccm.sum(axis=2, keepdims=True)
        out = ref_process.process(torch.from_numpy(bayer), torch.from_numpy(wb), torch.from_numpy(ccm), gamma=2.2, CRF=None).numpy()
        cases.update({name + '_bayer': bayer, name + '_wb': wb, name + '_ccm': ccm.astype(np.float32), name + '_out': out})
    np.savez_compressed(os.path.join(GOLD, 'isp.npz'), torch_version=torch.__version__, **cases)
    print('isp golden written')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--ref', default='/root/reference')
    ap.add_argument('--only', default='', help="'isp': mint only tests/golden/isp.npz")
    args = ap.parse_args()
    ref = os.path.abspath(args.ref)
    os.makedirs(GOLD, exist_ok=True)
    os.chdir(ref)                       # noise.py:187 loads camera_params relative to CWD
    sys.path.insert(0, ref)
    sys.path.insert(0, ROOT)
    if args.only == 'isp':
        install_stubs()
        return mint_isp()
    import noise as ref_noise           # the reference module itself
    from oracle import noise_ref as O

    # ---------------- camera params -> JSON -------------------------------------------------
    cams = {}
    for cam in CAMERAS:
        d = np.load(os.path.join('camera_params', 'release', cam + '_params.npy'), allow_pickle=True).item()
        out = {}
        for k, v in d.items():
            if isinstance(v, dict):
                out[k] = {kk: {kkk: float(vvv) for kkk, vvv in vv.items()} for kk, vv in v.items()}
            elif hasattr(v, 'tolist'):
                out[k] = np.asarray(v, dtype=np.float64).tolist()
            else:
                out[k] = float(v)
        cams[cam] = out
    with open(os.path.join(ROOT, 'eld_amd', 'camera_params.json'), 'w') as f:
        json.dump(cams, f, indent=1, sort_keys=True)

    # ---------------- noise sampler ---------------------------------------------------------
    param_sets = {
        'sony_mid': (np.float32(2.288), np.float32(6.451), 15583, np.float32(208.98)),     # SURVEY.md sec. 6 values
        'bright': (np.float32(0.1), np.float32(0.7), 15583, np.float32(100.0)),           # lambda up to 1558
        'dark': (np.float32(29.0), np.float32(40.0), 15583, np.float32(300.0)),           # lambda << 1
    }
    shapes = {'s': (4, 12, 20), 'ragged': (4, 5, 7), 'empty': (4, 0, 8)}
    seed = 2018
    case_id = 0
    for model in ('g', 'Pg', 'pg', 'P', 'p'):
        nm = ref_noise.NoiseModel(model=model, include=4)
        for pname, params in param_sets.items():
            for sname, shape in shapes.items():
                case_id += 1
                rng = np.random.default_rng(1000 + case_id)
                y = synth_clean(rng, shape)
                np.random.seed(seed + case_id)
                z_ref = nm(y, params=params)                      # <- the reference
                assert z_ref.dtype == np.float32, z_ref.dtype
                np.random.seed(seed + case_id)                    # replay its draws
                z_orc, v = O.noise_numpy_rng(y, model, params)
                assert np.array_equal(z_orc, z_ref), (model, pname, sname)
                np.savez_compressed(
                    os.path.join(GOLD, 'noise_%s_%s_%s.npz' % (model, pname, sname)),
                    y=y, z=z_ref, params=np.array([float(p) for p in params], np.float64), model=model,
                    np_seed=seed + case_id,
                    **{k: np.asarray(a) for k, a in v.items()})

    # default-params path (params=None): 5 host draws first, then the per-pixel blocks
    nm = ref_noise.NoiseModel(model='Pg', include=4)
    y = synth_clean(np.random.default_rng(7), (4, 8, 16))
    np.random.seed(99)
    z_ref = nm(y)                                                  # float64 under NumPy 2 (F7)
    np.random.seed(99)
    params = nm._sample_params()
    np.savez_compressed(os.path.join(GOLD, 'noise_default_params.npz'), y=y, z=np.asarray(z_ref, np.float64),
                        params=np.array(params, np.float64), np_seed=99, model='Pg')

    # ---------------- _sample_params -------------------------------------------------------
    recs = []
    for inc in (None, 4, 1):
        nm = ref_noise.NoiseModel(model='g', include=inc)
        for s in (0, 1, 2018):
            np.random.seed(s)
            for _ in range(3):
                recs.append([-1 if inc is None else inc, s] + [float(x) for x in nm._sample_params()])
    np.savez_compressed(os.path.join(GOLD, 'sample_params.npz'), recs=np.array(recs, np.float64))

    # ---------------- RawPacker ---------------------------------------------------------------
    rp = ref_noise.RawPacker('bayer')
    mosaic = np.random.default_rng(3).integers(0, 16383, size=(10, 14)).astype(np.float32)
    packed = rp.pack_raw(mosaic)
    np.savez_compressed(os.path.join(GOLD, 'rawpacker.npz'), mosaic=mosaic, packed=packed, unpacked=rp.unpack_raw(packed))

    # ---------------- LMDB decode + augmentation (need the App. D shims) -------------------------
    install_stubs()
    codes = np.arange(65536, dtype=np.uint16).reshape(4, 128, 128)
    tmp = '/tmp/_eld_gen_golden_db'
    os.makedirs(tmp, exist_ok=True)
    with open(os.path.join(tmp, 'meta_info.pkl'), 'wb') as f:
        pickle.dump({'shape': codes.shape, 'dtype': np.uint16}, f)
    sys.modules['lmdb'] = FakeLMDB([codes.tobytes()])
    import dataset.lmdb_dataset as ref_lmdb
    ds = ref_lmdb.LMDBDataset(tmp)
    dec = ds[0]                                                     # <- the reference decode
    assert dec.dtype == np.float32
    np.savez_compressed(os.path.join(GOLD, 'lmdb_decode.npz'), decoded=dec.reshape(-1))

    try:
        os.environ.setdefault('COLUMNS', '80')
        _popen = os.popen
        os.popen = lambda cmd, *a, **k: io.StringIO('24 80') if 'stty' in cmd else _popen(cmd, *a, **k)   # App. D item 3
        import dataset.sid_dataset as ref_sid
        os.popen = _popen

        class ListDS:
            def __init__(self, arrs):
                self.arrs = arrs

            def __getitem__(self, i):
                return self.arrs[i]

            def __len__(self):
                return len(self.arrs)
        rng = np.random.default_rng(11)
        tgt = [rng.uniform(0, 1, (4, 6, 6)).astype(np.float32) for _ in range(8)]
        inp = [rng.uniform(-0.2, 1.2, (4, 6, 6)).astype(np.float32) for _ in range(8)]
        tds = ref_sid.ELDTrainDataset(target_dataset=ListDS(tgt), input_datasets=[ListDS(inp)])
        outs_i, outs_t, bits = [], [], []
        for i in range(8):
            np.random.seed(500 + i)
            b = [int(np.random.randint(2, size=1)[0]) for _ in range(3)]
            np.random.seed(500 + i)
            d = tds[i]                                              # <- the reference augmentation + clip
            outs_i.append(d['input']); outs_t.append(d['target']); bits.append(b)
        np.savez_compressed(os.path.join(GOLD, 'augment.npz'), inp=np.stack(inp), tgt=np.stack(tgt),
                            out_inp=np.stack(outs_i), out_tgt=np.stack(outs_t), bits=np.array(bits))
    except Exception as ex:                                         # pragma: no cover
        print('augment fixture skipped:', repr(ex))

    # ---------------- U-Net -----------------------------------------------------------------------
    import importlib.util
    import torch
    spec = importlib.util.spec_from_file_location('ref_unet', os.path.join(ref, 'models', 'arch', 'Unet.py'))
    ref_unet = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_unet)
    torch.manual_seed(2018)
    net = ref_unet.UNetSeeInDark(4, 4)                               # default init, models/ELD_model.py:391-393
    g = torch.Generator().manual_seed(5)
    x = torch.rand(2, 4, 32, 48, generator=g)
    t = torch.rand(2, 4, 32, 48, generator=g)
    out = net(x)
    loss = torch.nn.L1Loss()(out, t)                                 # models/losses.py:32
    loss.backward()
    names, gsum, gabs, wsum = [], [], [], []
    for n_, p_ in net.named_parameters():
        names.append(n_)
        gsum.append(float(p_.grad.double().sum()))
        gabs.append(float(p_.grad.double().abs().sum()))
        wsum.append(float(p_.detach().double().sum()))
    keep = {n_: p_.grad.numpy().copy() for n_, p_ in net.named_parameters()
            if n_ in ('conv1_1.weight', 'conv1_1.bias', 'conv10_1.weight', 'conv10_1.bias', 'upv9.bias', 'conv9_2.bias')}
    np.savez_compressed(os.path.join(GOLD, 'unet.npz'), x=x.numpy(), t=t.numpy(), out=out.detach().numpy(),
                        loss=float(loss), names=np.array(names), gsum=np.array(gsum), gabs=np.array(gabs),
                        wsum=np.array(wsum), torch_version=torch.__version__,
                        **{'grad_' + k.replace('.', '__'): v for k, v in keep.items()})
    mint_isp()
    print('golden vectors written to', GOLD)


if __name__ == '__main__':
    main()
