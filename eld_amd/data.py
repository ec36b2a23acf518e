"""Dataset-side plugins: drop-ins for the three dataset classes on the training path of train_syn.py, re-cut so that the
per-pixel work happens on the MI355X and the DataLoader workers (train_syn.py:78-80, default --nThreads 8, forked) do what
they are good at -- I/O and a handful of NumPy scalar draws:

    LMDBDataset       dataset/lmdb_dataset.py:7-47    returns the stored uint16 codes UNDECODED (2 B/pixel over PCIe instead
                                                      of 4; the decode clip(u16/65535) is fused into the HIP sampler / augment)
    SynDataset        dataset/sid_dataset.py:248-284  draws the per-sample noise parameters exactly where the reference does
                                                      (noise_maker._sample_params(), same NumPy draws in the same order, once per
                                                      burst) and DEFERS the per-pixel synthesis: returns a `Deferred` sample
    ELDTrainDataset   dataset/sid_dataset.py:322-367  pairs input and target (i % N, i // N), draws the three augmentation bits
                                                      exactly as the reference does (np.random.randint(2, size=1)[0], three
                                                      times) and returns them instead of flipping host arrays

What reaches the model plugin (eld_amd.model.ELDModel.set_input) after torch's default collate is
    {'target': int16 view of the uint16 codes (B,C,H,W) | float32, 'params': uint8 (B,64) EldNoiseParams records,
     'aug': int64 (B,), 'burst': int64 (B,)}
and the model runs decode -> sampler (+clip, burst concat) -> augmentation of input and target on the device.
Samples whose input dataset is an ordinary array dataset (offline-noise LMDBs, train_syn.py:66-70) take the reference's host
path below unchanged (index maps and a clip -- the same arithmetic as sid_dataset.py:344-356).
"""
import pickle
import weakref
from os.path import join

import numpy as np
import torch.utils.data as tdata

from . import _lib as L
from .noise import NoiseParams


class LMDBDataset(tdata.Dataset):
    """dataset/lmdb_dataset.py:7-47 with `decode=False` by default: uint16 records stay uint16 (tagged by dtype), everything
    else (float32 sRGB databases) is returned as stored, like the reference."""

    def __init__(self, db_path, size=None, repeat=1, decode=False):
        import lmdb
        self.db_path = db_path
        self.env = lmdb.open(db_path, max_readers=1, readonly=True, lock=False, readahead=False, meminit=False)
        with self.env.begin(write=False) as txn:
            length = txn.stat()['entries']
        self.length = size or length
        self.repeat = repeat
        with open(join(db_path, 'meta_info.pkl'), 'rb') as f:
            self.meta = pickle.load(f)
        self.shape = self.meta['shape']
        self.dtype = self.meta['dtype']
        self.decode = decode

    def __getitem__(self, index):
        index = index % self.length
        with self.env.begin(write=False) as txn:
            raw_data = txn.get('{:08}'.format(index).encode('ascii'))
        x = np.frombuffer(raw_data, self.dtype).reshape(*self.shape)
        if self.dtype == np.uint16 and self.decode:          # the reference's host decode (lmdb_dataset.py:38-39)
            x = np.clip(x / 65535, 0, 1).astype(np.float32)
        return x

    def __len__(self):
        return int(self.length * self.repeat)

    def __repr__(self):
        return self.__class__.__name__ + ' (' + self.db_path + ')'


class Deferred(object):
    """A noisy sample whose pixels do not exist yet: the clean data, the parameter record of its noise and the burst count."""
    __slots__ = ('clean', 'params', 'burst', 'isp', 'source', 'index')

    def __init__(self, clean, params, burst, isp=None, source=None, index=None):
        self.clean, self.params, self.burst, self.isp = clean, params, burst, isp      # isp: (wb[4], ccm[3,3]) -> raw2rgb after the noise
        self.source, self.index = source, index      # where `clean` came from: dataset object and the index it was read at


class SynDataset(tdata.Dataset):
    """dataset/sid_dataset.py:248-284.  Same constructor.  The RNG draws the reference makes per sample before touching pixels
    -- `_sample_params()` once per call, or once per BURST when num_burst > 1 (:267-272) -- happen here, in the worker, from the
    worker's own NumPy stream (worker_init_fn, :17-18); the pixels are synthesised later, batched, on the device."""

    def __init__(self, dataset, size=None, flag=None, noise_maker=None, repeat=1, cfa='bayer', num_burst=1):
        super(SynDataset, self).__init__()
        self.size = size
        self.dataset = dataset
        self.flag = flag
        self.repeat = repeat
        self.noise_maker = noise_maker
        self.cfa = cfa
        self.num_burst = num_burst
        SynDataset._last = weakref.ref(self)

    _last = None      # weak reference to the instance the entry script built (alive as long as its DataLoader is): ELDModel.initialize reads
                      # num_burst off it -- a burst input has num_burst * channels planes

    @classmethod
    def last(cls):
        return cls._last() if cls._last is not None else None

    def __getitem__(self, i):
        i = i % self.size if self.size is not None else i % len(self.dataset)
        data = self.dataset[i]
        params = NoiseParams.coerce(self.noise_maker._sample_params())
        return Deferred(data, params, max(1, int(self.num_burst)), source=self.dataset, index=i)

    def __len__(self):
        size = self.size or len(self.dataset)
        return int(size * self.repeat)


class ISPDataset(tdata.Dataset):
    """dataset/sid_dataset.py:287-319 (the --stage_in srgb input, train_syn.py:55-58): noise on the raw patch, clip, raw -> sRGB
    with the patch's white balance and colour matrix (meta_info[i] = (wb, ccm), util/process.py:107-112), clip.  Deferred like
    SynDataset: the worker draws the parameters, the device runs sampler + ISP (eld_amd.isp.process)."""

    def __init__(self, dataset, noise_maker=None, cfa='bayer', meta_info=None, CRF=None):
        super(ISPDataset, self).__init__()
        self.dataset = dataset
        self.noise_maker = noise_maker
        self.cfa = cfa
        self.meta_info = dataset.meta if meta_info is None else meta_info
        self.CRF = CRF                                       # (E, fs) of process.load_CRF or None; the device ISP reads it off ISPDataset.last()
        ISPDataset._last = weakref.ref(self)

    _last = None      # weak reference to the instance the entry script built (train_syn.py:55-58), picked up by ELDModel for its CRF tables

    @classmethod
    def last(cls):
        return cls._last() if cls._last is not None else None

    def __getitem__(self, i):
        data = self.dataset[i]
        wb, ccm = self.meta_info[i]
        params = NoiseParams.coerce(self.noise_maker._sample_params()) if self.noise_maker is not None else None
        return Deferred(data, params, 1, isp=(np.asarray(wb, np.float32).reshape(4), np.asarray(ccm, np.float32).reshape(3, 3)))

    def __len__(self):
        return len(self.dataset)


def _as_wire(x):
    """uint16 codes travel as an int16 view (torch's default collate stacks int16; uint16 support is partial)."""
    x = np.ascontiguousarray(x)
    return x.view(np.int16) if x.dtype == np.uint16 else x


class ELDTrainDataset(tdata.Dataset):
    """dataset/sid_dataset.py:322-367.  Same constructor, same pairing (i % N, i // N), same three augmentation draws."""

    def __init__(self, target_dataset, input_datasets, size=None, flag=None, augment=True, cfa='bayer'):
        super(ELDTrainDataset, self).__init__()
        self.size = size
        self.target_dataset = target_dataset
        self.input_datasets = input_datasets
        self.flag = flag
        self.augment = augment
        self.cfa = cfa
        # SynDataset applies the noise to ITS OWN dataset[i] (sid_dataset.py:265-275), independently of target_dataset.  When both
        # are the same database read at the same index for EVERY i (train_syn.py:61-64: SynDataset(LMDBDataset(SID_Sony_Raw.db)) beside
        # the same target database, no `size` wrap-around) the clean patch IS the target: it is read once and travels once.  Decided once per
        # input dataset, so that every sample of a run carries the same keys (default_collate takes them from the first sample of a batch).
        self._shared = [self._shares_target(d) for d in input_datasets]
        # Data parallel (the reference is single-device): every rank's DataLoader walks ITS shard of the pairs -- item i of rank r is
        # pair i * world + r, ceil(total / world) items on every rank (the tail wraps: the gradient exchange averages per-rank means, so local batch
        # sizes must agree) -- so that world x batchSize DIFFERENT pairs make up a global batch under the unmodified train_syn.py.  Taken from the
        # torchrun environment at construction (the workers are forked later and inherit it).
        from . import dist as D
        self.world, self.rank = (D.world_size(), D.rank()) if D.world_size() > 1 else D.env_world()[:2]

    def _shares_target(self, d):
        if not isinstance(d, SynDataset):
            return False
        src, tgt = d.dataset, self.target_dataset
        if src is not tgt and not (getattr(src, 'db_path', None) is not None and getattr(src, 'db_path', None) == getattr(tgt, 'db_path', object())
                                   and getattr(src, 'length', None) == getattr(tgt, 'length', object())):
            return False
        try:                                                 # SynDataset reads dataset[i % (size or len(dataset))]: the index must never wrap below len(target),
            n_in = len(self.input_datasets)                  # and this dataset's own `size` must not run the pair index i // N past the target database
            return (d.size is None or d.size >= len(tgt)) and len(src) >= len(tgt) and (self.size is None or self.size <= len(tgt) * n_in)
        except TypeError:
            return False

    def _total(self):
        return self.size or len(self.target_dataset) * len(self.input_datasets)

    def __getitem__(self, i):
        N = len(self.input_datasets)
        if self.world > 1:
            from .dist import shard_item
            i = shard_item(i, self._total(), self.rank, self.world)
        inp = self.input_datasets[i % N][i // N]
        same = self._shared[i % N] and isinstance(inp, Deferred) and inp.isp is None and inp.index == i // N
        if self._shared[i % N] and isinstance(inp, Deferred) and inp.isp is None and not same:
            # the clean patch was declared to BE the target for this input dataset (no 'clean' key travels): a sample whose SynDataset index differs
            # from its pair index would be degraded from the wrong patch without a sign of trouble
            raise RuntimeError('ELDTrainDataset: sample %d pairs target %d with SynDataset patch %d although both read the same database '
                             '(index wrap-around): give the SynDataset its own dataset object' % (i, i // N, inp.index))
        target = inp.clean if same else self.target_dataset[i // N]
        bits = 0
        if self.augment:                                     # sid_dataset.py:344-352: flip H, flip W, transpose
            for b in (1, 2, 4):
                if np.random.randint(2, size=1)[0] == 1:
                    bits |= b
        if isinstance(inp, Deferred):
            if inp.isp is not None:                          # sRGB input stage: the raw patch to degrade travels beside the sRGB target
                dic = {'clean': _as_wire(inp.clean), 'target': _as_wire(target), 'wb': inp.isp[0], 'ccm': inp.isp[1], 'aug': bits, 'burst': 1}
                if inp.params is not None:
                    dic['params'] = inp.params.record(0).reshape(1).view(np.uint8).copy()
            else:
                dic = {'target': _as_wire(target), 'params': inp.params.record(0).reshape(1).view(np.uint8).copy(),
                       'aug': bits, 'burst': inp.burst}
                if not self._shared[i % N]:                  # (per dataset, not per sample: a batch never mixes samples with and without the key)
                    dic['clean'] = _as_wire(inp.clean)
        else:                                                # pre-synthesised input (offline-noise LMDB): the reference's host path
            if getattr(inp, 'dtype', None) == np.uint16:
                inp = np.clip(inp / 65535, 0, 1).astype(np.float32)
            if getattr(target, 'dtype', None) == np.uint16:
                target = np.clip(target / 65535, 0, 1).astype(np.float32)
            if bits & 1:
                target, inp = np.flip(target, axis=1), np.flip(inp, axis=1)
            if bits & 2:
                target, inp = np.flip(target, axis=2), np.flip(inp, axis=2)
            if bits & 4:
                target, inp = np.transpose(target, (0, 2, 1)), np.transpose(inp, (0, 2, 1))
            inp = np.maximum(np.minimum(inp, 1.0), 0)
            dic = {'input': np.ascontiguousarray(inp), 'target': np.ascontiguousarray(target)}
        if self.flag is not None:
            dic.update(self.flag)
        return dic

    def __len__(self):
        return (self._total() + self.world - 1) // self.world


def worker_init_fn(worker_id):                               # dataset/sid_dataset.py:17-18
    np.random.seed(np.random.get_state()[1][0] + worker_id)


def records_from_batch(params_u8):
    """(B,64) uint8 tensor/array of a collated batch -> structured NumPy records (a copy the caller may edit)."""
    a = np.ascontiguousarray(np.asarray(params_u8, dtype=np.uint8)).reshape(-1, 64)
    return a.view(L.NOISE_PARAMS_DTYPE).reshape(-1).copy()
