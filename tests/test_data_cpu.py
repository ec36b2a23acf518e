"""CPU tests of the dataset-side plugins (eld_amd/data.py) -- host logic only, no GPU, no HIP calls:
deferred-sample batches through torch's DataLoader with forked workers (train_syn.py:78-80), the draw order of the reference
(dataset/sid_dataset.py:259-280, 332-363; noise.py:201-225), and the host path for pre-synthesised inputs against the
reference-minted augmentation golden."""
import os
import subprocess
import sys

import numpy as np
import pytest

torch = pytest.importorskip('torch')

REF = '/root/reference'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class ArrayDB(object):
    """Stand-in for dataset/lmdb_dataset.py's LMDBDataset over in-memory uint16 records (no lmdb module needed)."""

    def __init__(self, n=6, shape=(4, 16, 16), seed=0, dtype=np.uint16):
        rng = np.random.default_rng(seed)
        if dtype == np.uint16:
            self.items = [np.floor(65535.0 * rng.uniform(size=shape) ** 2.2).astype(np.uint16) for _ in range(n)]
        else:
            self.items = [rng.uniform(-0.2, 1.2, size=shape).astype(np.float32) for _ in range(n)]

    def __getitem__(self, i):
        return self.items[i % len(self.items)]

    def __len__(self):
        return len(self.items)


def quiet_model(model='PGRU'):
    import contextlib
    import io
    from eld_amd.noise import NoiseModel
    with contextlib.redirect_stdout(io.StringIO()):
        return NoiseModel(model=model, include=4)


def test_deferred_sample_follows_the_reference_draw_order():
    """np.random.seed(s); ds[i]  ==  np.random.seed(s); _sample_params() then three randint(2) draws."""
    from eld_amd import _lib as L
    from eld_amd.data import ELDTrainDataset, SynDataset, records_from_batch
    nm = quiet_model('PGRU')
    clean = ArrayDB()
    ds = ELDTrainDataset(target_dataset=clean, input_datasets=[SynDataset(clean, noise_maker=nm)])
    assert len(ds) == 6
    for i in (0, 3, 5):
        np.random.seed(100 + i)
        d = ds[i]
        np.random.seed(100 + i)
        p = nm._sample_params()
        bits = 0
        for b in (1, 2, 4):
            if np.random.randint(2, size=1)[0] == 1:
                bits |= b
        rec = records_from_batch(d['params'][None])[0]
        assert d['aug'] == bits and d['burst'] == 1
        assert rec['K'] == np.float32(p[0]) and rec['g_scale'] == np.float32(p[1]) and rec['saturation'] == 15583 and rec['ratio'] == np.float32(p[3])
        assert rec['tl_lambda'] == np.float32(p.tl_lambda) and rec['tl_scale'] == np.float32(p.tl_scale) and rec['row_scale'] == np.float32(p.row_scale)
        assert d['target'].dtype == np.int16 and np.array_equal(d['target'].view(np.uint16), clean[i])      # codes travel undecoded
        assert d['params'].dtype == np.uint8 and d['params'].shape == (64,) and L.NOISE_PARAMS_DTYPE.itemsize == 64


def test_burst_draws_parameters_once_per_sample():
    from eld_amd.data import ELDTrainDataset, SynDataset
    nm = quiet_model('Pg')
    clean = ArrayDB()
    ds = ELDTrainDataset(clean, [SynDataset(clean, noise_maker=nm, num_burst=3)], augment=False)
    np.random.seed(5)
    d = ds[2]
    np.random.seed(5)
    p = nm._sample_params()                       # sid_dataset.py:269: one draw for the whole burst
    assert d['burst'] == 3 and d['aug'] == 0
    after = np.random.uniform()
    np.random.seed(5)
    ds[2]
    assert np.random.uniform() == after           # nothing else was drawn
    assert np.frombuffer(d['params'].tobytes(), np.float32)[0] == np.float32(p[0])


@pytest.mark.parametrize('workers', [0, 2])
def test_dataloader_batches(workers):
    """The batch that reaches ELDModel.set_input after default collate; forked workers never touch HIP."""
    from eld_amd.data import ELDTrainDataset, SynDataset, worker_init_fn
    nm = quiet_model('PGRU')
    clean = ArrayDB(n=8)
    ds = ELDTrainDataset(clean, [SynDataset(clean, noise_maker=nm, num_burst=2)])
    np.random.seed(3)
    loader = torch.utils.data.DataLoader(ds, batch_size=4, shuffle=False, num_workers=workers, worker_init_fn=worker_init_fn)
    batches = list(loader)
    assert len(batches) == 2
    b = batches[0]
    assert set(b) == {'target', 'params', 'aug', 'burst'}
    assert b['target'].dtype == torch.int16 and tuple(b['target'].shape) == (4, 4, 16, 16)
    assert b['params'].dtype == torch.uint8 and tuple(b['params'].shape) == (4, 64)
    assert tuple(b['aug'].shape) == (4,) and int(b['aug'].max()) <= 7 and b['burst'].tolist() == [2, 2, 2, 2]
    assert np.array_equal(b['target'][1].numpy().view(np.uint16), clean[1])


def test_presynthesised_inputs_take_the_reference_host_path(golden_dir):
    """Offline-noise input datasets (train_syn.py:66-70): flips / transpose / clip on the host, against the golden minted by
    running the reference's ELDTrainDataset (oracle/gen_golden.py)."""
    from eld_amd.data import ELDTrainDataset
    g = np.load(os.path.join(golden_dir, 'augment.npz'))

    class Fixed(object):
        def __init__(self, a):
            self.a = a

        def __getitem__(self, i):
            return self.a[i]

        def __len__(self):
            return len(self.a)
    ds = ELDTrainDataset(Fixed(g['tgt']), [Fixed(g['inp'])])
    real_randint = np.random.randint
    try:
        for i in range(len(g['inp'])):
            draws = iter(g['bits'][i])
            np.random.randint = lambda *a, **k: np.array([next(draws)])
            d = ds[i]
            assert set(d) == {'input', 'target'}
            assert np.array_equal(d['input'], g['out_inp'][i]) and np.array_equal(d['target'], g['out_tgt'][i])
            assert d['input'].flags['C_CONTIGUOUS'] and d['target'].flags['C_CONTIGUOUS']
    finally:
        np.random.randint = real_randint


def test_noise_applies_to_the_syn_datasets_own_sample():
    """sid_dataset.py:265-275: SynDataset degrades ITS OWN dataset[i], whatever the target dataset is.  Same database at the same
    index -> the patch is read once and travels once (no 'clean' key); a different database, size or repeat -> the clean patch
    travels beside the target."""
    from eld_amd.data import ELDTrainDataset, SynDataset
    nm = quiet_model('Pg')
    tgt, other = ArrayDB(n=6, seed=0), ArrayDB(n=6, seed=1)
    reads = []

    class Counting(ArrayDB):
        def __getitem__(self, i):
            reads.append(i)
            return ArrayDB.__getitem__(self, i)
    same_db = Counting(n=6, seed=0)
    d = ELDTrainDataset(same_db, [SynDataset(same_db, noise_maker=nm)], augment=False)[4]
    assert 'clean' not in d and np.array_equal(d['target'].view(np.uint16), same_db.items[4]) and reads == [4]      # one read, one copy
    d = ELDTrainDataset(tgt, [SynDataset(other, noise_maker=nm)], augment=False)[4]
    assert np.array_equal(d['clean'].view(np.uint16), other[4]) and np.array_equal(d['target'].view(np.uint16), tgt[4])
    d = ELDTrainDataset(tgt, [SynDataset(tgt, noise_maker=nm, size=3)], augment=False)[4]          # input index 4 % 3 = 1, target index 4
    assert np.array_equal(d['clean'].view(np.uint16), tgt[1]) and np.array_equal(d['target'].view(np.uint16), tgt[4])
    srgb_target = ArrayDB(n=6, shape=(3, 16, 16), dtype=np.float32)                               # --stage_out srgb beside a raw input
    d = ELDTrainDataset(srgb_target, [SynDataset(tgt, noise_maker=nm)], augment=False)[2]
    assert d['clean'].shape == (4, 16, 16) and d['target'].shape == (3, 16, 16)


def test_u16_codes_are_recognised_by_dtype_not_by_size():
    from eld_amd.noise import is_u16_codes
    assert is_u16_codes(torch.zeros(2, dtype=torch.int16)) and is_u16_codes(torch.zeros(2, dtype=torch.uint16))
    for dt in (torch.float16, torch.bfloat16, torch.float32, torch.int32):
        assert not is_u16_codes(torch.zeros(2, dtype=dt))
    assert not is_u16_codes(None)


def test_plugins_leave_burst_and_crf_for_the_model():
    """ELDModel.initialize reads the burst count off the SynDataset the script built (k * channels input planes) and the CRF tables
    off ISPDataset(CRF=...) (train_syn.py:42-58)."""
    from eld_amd import data as D
    from eld_amd.model import _plugin_num_burst
    nm = quiet_model('Pg')
    keep = D.SynDataset(ArrayDB(), noise_maker=nm, num_burst=3)
    assert _plugin_num_burst() == 3
    del keep                                       # weak reference: a dataset that no longer exists says nothing about the next model
    assert _plugin_num_burst() == 1
    keep = D.SynDataset(ArrayDB(), noise_maker=nm)
    assert _plugin_num_burst() == 1
    tables = (np.linspace(0, 1, 8), np.linspace(0, 1, 8) ** 0.5)
    db = ArrayDB()
    db.meta = [(np.ones(4), np.eye(3))] * 6
    isp = D.ISPDataset(db, noise_maker=nm, CRF=tables)
    assert D.ISPDataset.last() is isp and D.ISPDataset.last().CRF is tables
    del isp
    assert D.ISPDataset.last() is None


def test_calibrated_K_option():
    """noise.py:209-210 reads Kmin/Kmax, :215 ignores them; the option samples log K from the calibrated range."""
    nm = quiet_model('Pg')
    cp = nm.camera_params['SonyA7S2']
    np.random.seed(0)
    ks = [nm._sample_params()[0] for _ in range(200)]
    assert min(ks) >= 0.1 and max(ks) <= 30 and max(ks) > float(cp['Kmax'])
    nm.use_calibrated_K = True
    np.random.seed(0)
    ks = [nm._sample_params()[0] for _ in range(200)]
    assert min(ks) >= float(cp['Kmin']) and max(ks) <= float(cp['Kmax'])


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, 'train_syn.py')), reason='reference checkout not present')
def test_unmodified_train_syn_reaches_the_hip_model_with_all_plugins(tmp_path):
    """All four plugins installed, the UNMODIFIED train_syn.py runs through noise model, datasets (offline-noise LMDB redirected
    to on-device synthesis) and DataLoader construction and arrives in eld_amd.model.ELDModel.initialize -- which on this
    GPU-less box must refuse loudly (no CPU fallback)."""
    r = subprocess.run([sys.executable, '-m', 'eld_amd.launch', '--ref', REF, '--plugins', 'noise,arch,model,data', '--cwd', str(tmp_path),
                        '--stop-after-epochs', '1', '--max-iters-per-epoch', '1', '--',
                        '--name', 't', '--include', '4', '--noise', 'PGRU', '--gpu_ids', '0' if torch.cuda.is_available() else '-1', '--nThreads', '2', '--no-log', '--max_dataset_size', '2'],
                       capture_output=True, text=True, timeout=300, env=dict(os.environ, PYTHONPATH=ROOT))
    out = r.stdout + r.stderr
    if torch.cuda.is_available():
        assert r.returncode == 0, out[-3000:]
    else:
        assert 'on-device synthesis from SID_Sony_Raw.db' in out, out[-3000:]
        assert 'there is no CPU fallback' in out, out[-3000:]


def test_size_wraparound_keeps_the_clean_patch_with_the_sample():
    """ELDTrainDataset(size > len(target) * N) runs the pair index past the database, where SynDataset's own index wraps
    (sid_dataset.py:259-264) while the target's does not (ArrayDB wraps like the in-memory LMDB stand-in): the clean patch is then NOT the
    target, so the sample must carry its own 'clean' key (the decision is per input dataset: every sample of the run carries it)."""
    from eld_amd.data import ELDTrainDataset, SynDataset
    nm = quiet_model('Pg')
    clean = ArrayDB(n=4)
    ds = ELDTrainDataset(clean, [SynDataset(clean, noise_maker=nm)], size=6, augment=False)
    assert ds._shared == [False]
    for i in range(6):
        d = ds[i]
        assert 'clean' in d and np.array_equal(d['clean'].view(np.uint16), clean[i % 4])
    ds2 = ELDTrainDataset(clean, [SynDataset(clean, noise_maker=nm)], size=4, augment=False)
    assert ds2._shared == [True] and 'clean' not in ds2[3]
