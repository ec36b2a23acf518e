"""Noise-model plugin: drop-in for the reference `noise` module (noise.py), on the MI355X.

Same constructor, `__call__(y, params=None)` and `_sample_params()` as the reference
(noise.py:149-170, 175-225); the per-pixel synthesis runs in the fused HIP sampler
(eld_amd/csrc/noise.hip) through the C ABI (include/eld_amd.h: eld_noise_forward).

Differences a user can observe, all documented in DESIGN.md:
  * the per-pixel variates come from Philox4x32-7 counters (csrc/philox.h), not NumPy's MT19937 stream
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


_PINNED = {}


def _upload(arr, device):
    """Small host array -> device through a PINNED staging ring, asynchronously on the current stream (a pageable H2D copy is
    a blocking call in front of every step).  A slot is reused only after the copy issued from it has completed."""
    import torch
    raw = np.ascontiguousarray(arr).view(np.uint8).reshape(-1)
    n = raw.size
    ring = _PINNED.setdefault(device, {'slots': [], 'next': 0})
    if not ring['slots']:
        ring['slots'] = [[torch.empty(4096, dtype=torch.uint8).pin_memory(), None] for _ in range(8)]
    slot = ring['slots'][ring['next'] % len(ring['slots'])]
    ring['next'] += 1
    if slot[0].numel() < n:
        slot[0] = torch.empty(max(n, 2 * slot[0].numel()), dtype=torch.uint8).pin_memory()
    if slot[1] is not None:
        slot[1].synchronize()           # the copy that last used this slot (8 uploads ago) has long finished
    slot[0][:n].copy_(torch.from_numpy(raw))
    dev = slot[0][:n].to(device, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(device))
    slot[1] = ev
    return dev                          # uint8 view of the bytes; callers pass its device pointer through the C ABI


def make_records(params, sample_ids):
    """list of parameter tuples + global sample ids -> structured EldNoiseParams records."""
    if len(params) == 0:
        return np.zeros((0,), L.NOISE_PARAMS_DTYPE)
    return np.stack([NoiseParams.coerce(p).record(int(s)) for p, s in zip(params, sample_ids)])


def set_sample_ids(recs, sample_ids):
    ids = np.asarray(sample_ids, dtype=np.uint64)
    recs['sample_id_lo'] = (ids & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    recs['sample_id_hi'] = (ids >> np.uint64(32)).astype(np.uint32)
    return recs


def is_u16_codes(t):
    """True for a tensor of LMDB uint16 codes as they travel (int16 view or uint16).  float16 / bfloat16 tensors are 2 bytes per
    element too but are VALUES, never reinterpreted as codes."""
    import torch
    return t is not None and t.dtype in (torch.int16, torch.uint16)


def sample_noise_records(y, recs, flags, seed, in_u16=False, inject=None, dump=None, out=None, burst_index=0, burst=1):
    """The C-ABI sampler call.  y: CUDA (N,C,H,W) float32, or int16/uint16 LMDB codes when in_u16; recs: N structured records
    (host); out: CUDA float32 (N, burst*C, H, W) -- this call fills channels [burst_index*C, (burst_index+1)*C) of every image
    (burst == 1: the whole tensor)."""
    import torch
    assert y.is_cuda and y.is_contiguous() and y.dim() == 4
    N, C, H, W = y.shape
    assert len(recs) == N
    if in_u16:
        assert is_u16_codes(y), 'in_u16 needs int16/uint16 codes, got %s' % (y.dtype,)
    else:
        assert y.dtype == torch.float32
    prm = _upload(np.ascontiguousarray(recs).view(np.uint8).reshape(-1), y.device)
    if out is None:
        out = torch.empty((N, burst * C, H, W), dtype=torch.float32, device=y.device)
    assert out.is_contiguous() and out.dtype == torch.float32 and tuple(out.shape) == (N, burst * C, H, W)
    chw = C * H * W
    optr = None if out.numel() == 0 else C_void(out.data_ptr() + 4 * burst_index * chw)
    rc = L.lib().eld_noise_forward_strided(L.dptr(y), L.IN_U16 if in_u16 else L.IN_F32, chw, optr, burst * chw, L.dptr(prm),
                                           N, C, H, W, int(flags), int(seed) & (2 ** 64 - 1), L.dptr(inject), L.dptr(dump), L.cur_stream())
    L.check(rc, 'eld_noise_forward_strided')
    return out


def C_void(addr):
    import ctypes
    return ctypes.c_void_p(addr)


def sample_noise(y, params, flags, seed, sample_ids, in_u16=False, inject=None, dump=None, out=None):
    """Device-side batched sampler call.  y: CUDA tensor (N,C,H,W) float32 (or uint16 codes viewed as
    int16 when in_u16); params: list of N NoiseParams; returns CUDA float32 tensor (N,C,H,W)."""
    return sample_noise_records(y, make_records(params, sample_ids), flags, seed, in_u16=in_u16, inject=inject, dump=dump, out=out)


def decode_augment_u16(codes, bits=None):
    """CUDA int16/uint16 LMDB codes (N,C,H,W) -> float32 clip(u16/65535) (lmdb_dataset.py:38-39), optionally through the
    ELDTrainDataset index maps (sid_dataset.py:344-352); one HIP pass."""
    import torch
    assert codes.is_cuda and is_u16_codes(codes) and codes.dim() == 4, 'uint16 codes (int16 view) expected, got %s' % (codes.dtype,)
    codes = codes.contiguous()
    N, C, H, W = codes.shape
    out = torch.empty((N, C, H, W), dtype=torch.float32, device=codes.device)
    b, flags = None, 0
    if bits is not None:
        bits = [int(v) for v in bits]
        if H != W:
            if any(v & 4 for v in bits):
                raise ValueError('augment: transpose bit set on a batch of non-square %dx%d images' % (H, W))
            flags |= L.AUG_NOTRANSPOSE
        b = _upload(np.asarray(bits, dtype=np.int32), codes.device)
    L.check(L.lib().eld_augment_u16(L.dptr(codes), L.dptr(out), L.dptr(b), N, C, H, W, flags, L.cur_stream()), 'eld_augment_u16')
    return out


def augment(x, bits, clip=False):
    """Device-side ELDTrainDataset augmentation (sid_dataset.py:344-354): x CUDA [N,C,H,W]; bits[n] = flipH | flipW<<1 |
    transpose<<2 (the three np.random.randint(2) draws, in the reference's order)."""
    import torch
    x = x.contiguous().float()
    N, C, H, W = x.shape
    out = torch.empty_like(x)
    bits = [int(v) for v in bits]
    flags = L.CLIP if clip else 0
    if H != W:                          # the reference transposes single (C,H,W) samples; a batched tensor cannot hold both shapes
        if any(v & 4 for v in bits):
            raise ValueError('augment: transpose bit set on a batch of non-square %dx%d images' % (H, W))
        flags |= L.AUG_NOTRANSPOSE
    b = _upload(np.asarray(bits, dtype=np.int32), x.device)
    L.check(L.lib().eld_augment(L.dptr(x), L.dptr(out), L.dptr(b), N, C, H, W, flags, L.cur_stream()), 'eld_augment')
    return out


class RawPacker:
    """Bayer and X-Trans pack/unpack (noise.py:6-145) on the device: same method names, same `cfa` switch, NotImplementedError for
    an unknown CFA (noise.py:135,144).  ndarray in -> ndarray out (float32, like the reference); CUDA tensor in -> CUDA tensor
    out, optionally batched on a leading axis."""
    def __init__(self, cfa='bayer'):
        self.cfa = cfa

    def _run(self, fn, src, packed_in, out_shape, a, b):
        import torch
        as_np = isinstance(src, np.ndarray)
        t = torch.from_numpy(np.ascontiguousarray(src, dtype=np.float32)).cuda() if as_np else src.contiguous().float()
        batched = t.dim() == (4 if packed_in else 3)
        if not batched:
            t = t.unsqueeze(0)
        N = t.shape[0]
        out = torch.empty((N,) + out_shape, dtype=torch.float32, device=t.device)
        L.check(getattr(L.lib(), fn)(L.dptr(t), L.dptr(out), N, a, b, L.cur_stream()), fn)
        if not batched:
            out = out[0]
        return out.cpu().numpy() if as_np else out

    def pack_raw_bayer(self, cfa_img):                     # noise.py:10-20
        H, W = cfa_img.shape[-2:]
        return self._run('eld_pack_bayer', cfa_img, False, (4, H // 2, W // 2), H // 2, W // 2)

    def unpack_raw_bayer(self, img):                       # noise.py:66-81
        h, w = img.shape[-2:]
        return self._run('eld_unpack_bayer', img, True, (2 * h, 2 * w), h, w)

    def pack_raw_xtrans(self, cfa_img):                    # noise.py:22-64 (sides truncated to whole 6x6 cells, :25-26)
        H, W = cfa_img.shape[-2:]
        return self._run('eld_pack_xtrans', cfa_img, False, (9, 2 * (H // 6), 2 * (W // 6)), H, W)

    def unpack_raw_xtrans(self, img):                      # noise.py:83-127
        h, w = img.shape[-2:]
        return self._run('eld_unpack_xtrans', img, True, (3 * h, 3 * w), h, w)

    def pack_raw(self, cfa_img):                           # noise.py:129-136
        if self.cfa == 'bayer':
            return self.pack_raw_bayer(cfa_img)
        elif self.cfa == 'xtrans':
            return self.pack_raw_xtrans(cfa_img)
        raise NotImplementedError

    def unpack_raw(self, img):                             # noise.py:138-145
        if self.cfa == 'bayer':
            return self.unpack_raw_bayer(img)
        elif self.cfa == 'xtrans':
            return self.unpack_raw_xtrans(img)
        raise NotImplementedError


def pack_raw_bayer(raw_image_visible, raw_pattern, black_level_per_channel, white_point=16383):
    """dataset/sid_dataset.py:172-196 on the device: uint16 sensor mosaic (2h,2w) or (N,2h,2w) [ndarray or CUDA int16/uint16
    tensor] -> packed, black-level-normalised float32 (4,h,w) / (N,4,h,w) in R, G1, B, G2 order.  Takes what the reference
    reads off a rawpy object: raw.raw_image_visible, raw.raw_pattern, raw.black_level_per_channel."""
    import ctypes
    import torch
    as_np = isinstance(raw_image_visible, np.ndarray)
    t = torch.from_numpy(np.ascontiguousarray(raw_image_visible, dtype=np.uint16).view(np.int16)).cuda() if as_np else raw_image_visible.contiguous()
    assert t.is_cuda and is_u16_codes(t)
    single = t.dim() == 2
    if single:
        t = t.unsqueeze(0)
    N, H2, W2 = t.shape
    if H2 % 2 or W2 % 2:
        raise ValueError('mosaic sides must be even, got %dx%d' % (H2, W2))
    out = torch.empty((N, 4, H2 // 2, W2 // 2), dtype=torch.float32, device=t.device)
    pat = (ctypes.c_int * 4)(*[int(v) for v in np.asarray(raw_pattern).reshape(-1)])
    blk = (ctypes.c_float * 4)(*[float(np.float32(v)) for v in black_level_per_channel])
    L.check(L.lib().eld_pack_raw_bayer_u16(L.dptr(t), L.dptr(out), N, H2 // 2, W2 // 2, pat, blk, float(white_point), L.cur_stream()), 'eld_pack_raw_bayer_u16')
    if single:
        out = out[0]
    return out.cpu().numpy() if as_np else out


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
    last_instance = None

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
        # noise.py:209-210 reads the calibrated Kmin/Kmax and then samples log K from the hard-coded [0.1, 30] (:215); the
        # calibrated range is the paper's.  Default = the reference's behaviour; ELD_AMD_CALIBRATED_K=1 or this attribute opts in.
        self.use_calibrated_K = os.environ.get('ELD_AMD_CALIBRATED_K', '0') == '1'
        NoiseModel.last_instance = self                     # eld_amd.model attaches the most recent instance (train_syn.py:38 -> :89)

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
        if self.use_calibrated_K:
            log_K = np.random.uniform(low=np.log(camera_params['Kmin']), high=np.log(camera_params['Kmax']))
        else:
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
