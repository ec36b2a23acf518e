"""Noise-model plugin: drop-in for the reference `noise` module (noise.py), on the MI355X.

Same constructor, `__call__(y, params=None)` and `_sample_params()` as the reference
(noise.py:149-170, 175-225); the per-pixel synthesis runs in the fused HIP sampler
(eld_amd/csrc/noise.hip) through the C ABI (include/eld_amd.h: eld_noise_forward).

Differences a user can observe, all documented in DESIGN.md:
  * the per-pixel variates come from Philox4x32-10 counters, not NumPy's MT19937 stream
    (SURVEY.md F8): same distribution, different numbers.  `_sample_params()` still draws
    from np.random in the reference's order, so the per-image parameters are reproducible
    with np.random.seed exactly as before;
  * besides ndarray (C,H,W) in / ndarray out, `__call__` accepts a CUDA tensor (C,H,W) or
    (N,C,H,W) and then returns a CUDA tensor without touching the host;
  * model letters G / R / U / B select the withheld ELD terms (Tukey-lambda read, row,
    quantisation, colour bias) -- the reference ignores unknown letters (noise.py:158-166).
There is no CPU path: calling the plugin without a GPU / built library raises.
"""
import json
import os
from os.path import join

import numpy as np

from . import _lib as L

_HERE = os.path.dirname(os.path.abspath(__file__))
ALL_CAMERAS = ['CanonEOS5D4', 'CanonEOS70D', 'CanonEOS700D', 'NikonD850', 'SonyA7S2']   # noise.py:179


def model_flags(model):
    """Letter-containment parse, as noise.py:158-166 ('P' shadows 'p')."""
    f = 0
    if 'P' in model:
        f |= L.SHOT_POISSON
    elif 'p' in model:
        f |= L.SHOT_GAUSS
    if 'g' in model:
        f |= L.READ_GAUSS
    if 'G' in model:
        f |= L.READ_TL
    if 'R' in model:
        f |= L.ROW
    if 'U' in model:
        f |= L.QUANT
    if 'B' in model:
        f |= L.CBIAS
    return f


class NoiseParams(tuple):
    """(K, g_scale, saturation_level, ratio) exactly as noise.py:225 returns it; the withheld-model
    terms ride along as attributes so burst call sites (sid_dataset.py:269-272) keep working."""
    def __new__(cls, K, g_scale, saturation_level, ratio, tl_lambda=0.0, tl_scale=0.0, row_scale=0.0,
                q_step=1.0, color_bias=(0.0, 0.0, 0.0, 0.0)):
        self = super().__new__(cls, (K, g_scale, saturation_level, ratio))
        self.tl_lambda, self.tl_scale, self.row_scale, self.q_step = tl_lambda, tl_scale, row_scale, q_step
        self.color_bias = tuple(color_bias)
        return self

    @classmethod
    def coerce(cls, p):
        if isinstance(p, cls):
            return p
        if isinstance(p, dict):
            return cls(p['K'], p['g_scale'], p.get('saturation', 16383 - 800), p['ratio'], p.get('tl_lambda', 0.0),
                       p.get('tl_scale', 0.0), p.get('row_scale', 0.0), p.get('q_step', 1.0),
                       p.get('color_bias', (0.0,) * 4))
        K, g, s, r = p
        return cls(K, g, s, r)

    def record(self, sample_id):
        rec = np.zeros((), dtype=L.NOISE_PARAMS_DTYPE)
        rec['K'], rec['g_scale'], rec['saturation'], rec['ratio'] = self[0], self[1], self[2], self[3]
        rec['tl_lambda'], rec['tl_scale'], rec['row_scale'], rec['q_step'] = \
            self.tl_lambda, self.tl_scale, self.row_scale, self.q_step
        rec['color_bias'] = self.color_bias
        rec['sample_id_lo'], rec['sample_id_hi'] = sample_id & 0xFFFFFFFF, (sample_id >> 32) & 0xFFFFFFFF
        return rec


def load_camera_params(camera, param_dir=None):
    """Calibrated tables.  The reference np.load()s `camera_params/release/<cam>_params.npy` relative
    to the CWD (noise.py:187,195-196); that is honoured when the file is there, otherwise the JSON
    restatement shipped with this package (minted by oracle/gen_golden.py) is used."""
    if param_dir is not None:
        p = join(param_dir, camera + '_params.npy')
        if os.path.exists(p):
            return np.load(p, allow_pickle=True).item()
    with open(join(_HERE, 'camera_params.json')) as f:
        d = json.load(f)[camera]
    out = {}
    for k, v in d.items():
        if k == 'color_bias':
            out[k] = np.asarray(v, np.float32)
        elif k == 'G_shape':
            out[k] = np.asarray(v, np.float64)
        elif isinstance(v, dict):
            out[k] = {kk: {kkk: np.float64(vvv) for kkk, vvv in vv.items()} for kk, vv in v.items()}
        else:
            out[k] = np.float64(v)
    return out


def sample_noise(y, params, flags, seed, sample_ids, in_u16=False, inject=None, dump=None, out=None):
    """Device-side batched sampler call.  y: CUDA tensor (N,C,H,W) float32 (or uint16 codes viewed as
    int16 when in_u16); params: list of N NoiseParams; returns CUDA float32 tensor (N,C,H,W)."""
    import torch
    assert y.is_cuda and y.is_contiguous() and y.dim() == 4
    N, C, H, W = y.shape
    recs = np.stack([NoiseParams.coerce(p).record(int(s)) for p, s in zip(params, sample_ids)]) if N else \
        np.zeros((0,), L.NOISE_PARAMS_DTYPE)
    prm = torch.from_numpy(recs.view(np.uint8).reshape(-1).copy()).to(y.device, non_blocking=True)
    if out is None:
        out = torch.empty((N, C, H, W), dtype=torch.float32, device=y.device)
    rc = L.lib().eld_noise_forward(L.dptr(y), L.IN_U16 if in_u16 else L.IN_F32, L.dptr(out), L.dptr(prm),
                                   N, C, H, W, int(flags), int(seed) & (2 ** 64 - 1), L.dptr(inject), L.dptr(dump),
                                   L.cur_stream())
    L.check(rc, 'eld_noise_forward')
    return out


def augment(x, bits, clip=False):
    """Device-side ELDTrainDataset augmentation (sid_dataset.py:344-354): x CUDA [N,C,H,W]; bits[n] = flipH | flipW<<1 |
    transpose<<2 (the three np.random.randint(2) draws, in the reference's order)."""
    import torch
    x = x.contiguous().float()
    N, C, H, W = x.shape
    out = torch.empty_like(x)
    b = torch.as_tensor(list(bits), dtype=torch.int32, device=x.device)
    L.check(L.lib().eld_augment(L.dptr(x), L.dptr(out), L.dptr(b), N, C, H, W, L.CLIP if clip else 0, L.cur_stream()), 'eld_augment')
    return out


class RawPacker:
    """Bayer pack/unpack (noise.py:6-145) on the device.  X-Trans packing is outside the hot path
    (only the Bayer maps are in scope) and raises NotImplementedError like an
    unknown CFA does in the reference (noise.py:135,144)."""
    def __init__(self, cfa='bayer'):
        self.cfa = cfa

    def _run(self, fn, src, out_shape, h, w):
        import torch
        as_np = isinstance(src, np.ndarray)
        t = torch.from_numpy(np.ascontiguousarray(src, dtype=np.float32)).cuda() if as_np else src.contiguous().float()
        batched = t.dim() == (4 if fn == 'eld_unpack_bayer' else 3)
        if not batched:
            t = t.unsqueeze(0)
        N = t.shape[0]
        out = torch.empty((N,) + out_shape, dtype=torch.float32, device=t.device)
        L.check(getattr(L.lib(), fn)(L.dptr(t), L.dptr(out), N, h, w, L.cur_stream()), fn)
        if not batched:
            out = out[0]
        return out.cpu().numpy() if as_np else out

    def pack_raw_bayer(self, cfa_img):
        H, W = cfa_img.shape[-2:]
        return self._run('eld_pack_bayer', cfa_img, (4, H // 2, W // 2), H // 2, W // 2)

    def unpack_raw_bayer(self, img):
        h, w = img.shape[-2:]
        return self._run('eld_unpack_bayer', img, (2 * h, 2 * w), h, w)

    def pack_raw(self, cfa_img):
        if self.cfa == 'bayer':
            return self.pack_raw_bayer(cfa_img)
        raise NotImplementedError

    def unpack_raw(self, img):
        if self.cfa == 'bayer':
            return self.unpack_raw_bayer(img)
        raise NotImplementedError


class NoiseModelBase:  # same name / role as noise.py:148
    seed = int(os.environ.get('ELD_AMD_SEED', '2018'))      # Philox key (reference --seed default, base_option.py:22)
    sample_base = 0                                         # first global sample index handed out by this instance
    sample_stride = 1                                       # data-parallel: rank r uses base=r, stride=world

    def _next_ids(self, n):
        ids = [self.sample_base + self.sample_stride * (self._counter + i) for i in range(n)]
        self._counter += n
        return ids

    def __call__(self, y, params=None):
        import torch
        if not hasattr(self, '_counter'):
            self._counter = 0
        as_np = isinstance(y, np.ndarray)
        if as_np:
            wi = torch.utils.data.get_worker_info()
            if wi is not None:
                raise RuntimeError(
                    'eld_amd.noise.NoiseModel was called inside a DataLoader worker process; the HIP sampler '
                    'runs in the training process.  Use --nThreads 0 (the sampler is ~10^4x faster than the '
                    'NumPy one, workers are not needed) or let eld_amd.model.ELDModel synthesise on device.')
            t = torch.from_numpy(np.ascontiguousarray(y, dtype=np.float32)).cuda()
        else:
            t = y.contiguous()
            if t.dtype != torch.float32:
                t = t.float()
        single = t.dim() == 3
        if single:
            t = t.unsqueeze(0)
        N = t.shape[0]
        if params is None:
            plist = [self._sample_params() for _ in range(N)]
        elif isinstance(params, list):
            plist = params
        else:
            plist = [params] * N
        out = sample_noise(t, plist, model_flags(self.model), self.seed, self._next_ids(N))
        if single:
            out = out[0]
        return out.cpu().numpy() if as_np else out


class NoiseModel(NoiseModelBase):
    def __init__(self, model='g', cameras=None, include=None, exclude=None, cfa='bayer'):
        super().__init__()
        assert cfa in ['bayer', 'xtrans']                   # noise.py:177
        assert include is None or exclude is None           # noise.py:178
        self.cameras = cameras or list(ALL_CAMERAS)
        if include is not None:                             # noise.py:181-182
            self.cameras = [self.cameras[include]]
        if exclude is not None:                             # noise.py:183-185
            exclude_camera = set([self.cameras[exclude]])
            self.cameras = list(set(self.cameras) - exclude_camera)
        self.param_dir = join('camera_params', 'release')   # noise.py:187
        print('[i] NoiseModel with {}'.format(self.param_dir))        # noise.py:189-191
        print('[i] cameras: {}'.format(self.cameras))
        print('[i] using noise model {}'.format(model))
        self.camera_params = {}
        for camera in self.cameras:                         # noise.py:194-196
            self.camera_params[camera] = load_camera_params(camera, self.param_dir)
        self.model = model
        self.raw_packer = RawPacker(cfa)                    # noise.py:199
        self._counter = 0

    def _sample_params(self):
        """noise.py:201-225, same five draws from the global NumPy RandomState in the same order.
        When the model string asks for withheld terms (G/R/U/B) their parameters are drawn AFTER the
        reference's five, so the reference's tuple is unchanged for a given np.random.seed."""
        camera = np.random.choice(self.cameras)
        saturation_level = 16383 - 800
        profiles = ['Profile-1']
        camera_params = self.camera_params[camera]
        profile = np.random.choice(profiles)
        prof = camera_params[profile]
        log_K = np.random.uniform(low=np.log(1e-1), high=np.log(30))
        log_g_scale = np.random.standard_normal() * prof['g_scale']['sigma'] * 1 + \
            prof['g_scale']['slope'] * log_K + prof['g_scale']['bias']
        K = np.exp(log_K)
        g_scale = np.exp(log_g_scale)
        ratio = np.random.uniform(low=100, high=300)
        if not any(ch in self.model for ch in 'GRUB'):
            return NoiseParams(K, g_scale, saturation_level, ratio)

        def reg(name):        # log sigma | log K ~ N(slope*logK + bias, sigma)   [ELD paper eq. for joint sampling]
            r = prof[name]
            return float(np.exp(np.random.standard_normal() * r['sigma'] + r['slope'] * log_K + r['bias']))
        tl_scale, row_scale = reg('G_scale'), reg('R_scale')
        i = np.random.randint(len(camera_params['G_shape']))
        return NoiseParams(K, g_scale, saturation_level, ratio, tl_lambda=float(camera_params['G_shape'][i]),
                           tl_scale=tl_scale, row_scale=row_scale, q_step=1.0,
                           color_bias=tuple(float(b) for b in np.asarray(camera_params['color_bias'])[i]))
