"""GPU tests of the drop-in path: the call sequence of the reference's entry script (train_syn.py:38-113) over the eld_amd
plugins, with the input synthesised ON DEVICE from deferred samples -- NoiseModel -> LMDBDataset -> SynDataset ->
ELDTrainDataset -> DataLoader(workers) -> Engine -> set_learning_rate -> engine.train x 2 epochs -- checked per iteration
against the oracle (sampler arithmetic replayed from the dumped Philox variates, oracle augmentation, torch-CPU U-Net step +
torch.optim.Adam).  Runs without /root/reference (absent on the GPU box): LMDB records come from the in-memory `lmdb` stand-in
of eld_amd.launch.install_shims."""
import contextlib
import io
import os
import sys
import types

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')

from oracle import noise_ref as O      # noqa: E402  (checker only)
from oracle import unet_ref as U       # noqa: E402


def make_opt(tmp, **kw):
    d = dict(gpu_ids=[0], isTrain=True, checkpoints_dir=str(tmp), name='t', netG='unet', channels=4, stage_in='raw', stage_out='raw',
             lr=1e-4, beta1=0.9, wd=0.0, loss='l1', resume=False, chop=False, no_log=True, save_epoch_freq=100, model='eld_model', seed=2018)
    d.update(kw)
    return types.SimpleNamespace(**d)


@pytest.fixture()
def lmdb_stub(tmp_path):
    """The synthetic in-memory lmdb module (4 records of 4x512x512 uint16 codes), installed for the duration of one test."""
    import eld_amd.launch as Lm
    saved = sys.modules.pop('lmdb', None)
    cwd = os.getcwd()
    os.chdir(tmp_path)
    Lm.install_shims(synthetic_lmdb=True, patches=4, patch_hw=(512, 512))
    try:
        yield
    finally:
        os.chdir(cwd)
        sys.modules.pop('lmdb', None)
        if saved is not None:
            sys.modules['lmdb'] = saved


def oracle_params(rec):
    return O.Params(K=float(rec['K']), g_scale=float(rec['g_scale']), saturation=float(rec['saturation']), ratio=float(rec['ratio']),
                    tl_lambda=float(rec['tl_lambda']), tl_scale=float(rec['tl_scale']), row_scale=float(rec['row_scale']), q_step=float(rec['q_step']),
                    color_bias=tuple(float(v) for v in rec['color_bias']))


def oracle_batch(batch, model_letters, seed, sample_ids, burst=1):
    """What the model's set_input must produce for a deferred batch: replay on the CPU with the variates the HIP sampler used."""
    from eld_amd import _lib as L
    from eld_amd.data import records_from_batch
    from eld_amd.noise import model_flags, sample_noise_records, set_sample_ids
    codes = batch['target'].numpy().view(np.uint16)
    clean = O.lmdb_decode_u16(codes)
    recs = records_from_batch(batch['params'].numpy())
    flags = model_flags(model_letters) | L.CLIP
    N = codes.shape[0]
    frames = []
    for k in range(burst):
        dump = torch.zeros(L.NPLANES, clean.size, device='cuda')
        z = sample_noise_records(batch['target'].cuda(), set_sample_ids(recs.copy(), sample_ids[k::burst]), flags, seed, in_u16=True, dump=dump)
        dv = dump.cpu().numpy()
        ref = np.stack([O.noise_arith(clean[i], oracle_params(recs[i]), flags,
                                      **{name: dv[j].reshape(clean.shape)[i] for name, j in L.PLANE.items()}) for i in range(N)])
        assert np.array_equal(z.cpu().numpy(), ref)          # production variates, reference arithmetic: same bits
        frames.append(ref)
    noisy = np.concatenate(frames, axis=1)
    inp, tgt = [], []
    for i in range(N):
        b = int(batch['aug'][i])
        inp.append(np.clip(O.augment(noisy[i], b & 1, b & 2, b & 4), 0, 1))
        tgt.append(O.augment(clean[i], b & 1, b & 2, b & 4))
    return np.stack(inp).astype(np.float32), np.stack(tgt).astype(np.float32)


def test_train_syn_call_sequence_two_epochs_vs_oracle(eld_lib, lmdb_stub, tmp_path):
    import eld_amd.noise as noise                            # `import noise`                                (train_syn.py:10)
    from eld_amd import data as datasets
    from eld_amd.engine import Engine
    np.random.seed(2018)
    torch.manual_seed(2018)
    with contextlib.redirect_stdout(io.StringIO()):
        noise_model = noise.NoiseModel(model='PGRU', include=4)                                           # train_syn.py:38
    target_data = datasets.LMDBDataset('data/Train/SID_Sony_Raw.db')                                      # train_syn.py:48-53
    input_data = datasets.SynDataset(datasets.LMDBDataset('data/Train/SID_Sony_Raw.db'), noise_maker=noise_model, num_burst=1)   # :61-64
    train_dataset = datasets.ELDTrainDataset(target_dataset=target_data, input_datasets=[input_data])     # train_syn.py:73
    assert len(train_dataset) == 4
    loader = torch.utils.data.DataLoader(train_dataset, batch_size=2, shuffle=True, num_workers=2, pin_memory=True,
                                         worker_init_fn=datasets.worker_init_fn)                           # train_syn.py:78-80
    engine = Engine(make_opt(tmp_path))                                                                    # train_syn.py:89
    assert engine.model.noise_model is noise_model           # attached without anyone calling set_noise_model
    engine.model.opt.save_epoch_freq = 100                                                                 # train_syn.py:97
    engine.set_learning_rate(1e-4)                                                                         # train_syn.py:99
    sd = {k: v.detach().cpu().clone() for k, v in engine.model.netG.state_dict().items()}
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    opt = torch.optim.Adam(list(params.values()), lr=1e-4, betas=(0.9, 0.999))
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    seed = engine.model.seed
    assert seed == 2018
    it = 0
    while engine.epoch < 2:                                                                                # train_syn.py:100
        np.random.seed(77 + engine.epoch)                    # (the script reseeds from OS entropy, :101; fixed here)
        if engine.epoch == 1:
            engine.set_learning_rate(5e-5)                                                                 # train_syn.py:102-105
            for g in opt.param_groups:
                g['lr'] = 5e-5
        batches = list(loader)                               # fork workers, collate: the batches engine.train will see
        assert len(batches) == 2 and set(batches[0]) == {'target', 'params', 'aug', 'burst'}
        losses = []
        orig = engine.model.get_current_errors
        engine.model.get_current_errors = lambda: (losses.append(orig()['Pixel']) or {'Pixel': losses[-1]})
        engine.train(batches)                                                                              # train_syn.py:107
        engine.model.get_current_errors = orig
        for b, loss in zip(batches, losses):
            ids = [2 * it, 2 * it + 1]                       # global sample ids: the model counts synthesised frames
            x, t = oracle_batch(b, 'PGRU', seed, ids)
            opt.zero_grad()
            lref = torch.nn.functional.l1_loss(U.unet_forward(params, torch.from_numpy(x)), torch.from_numpy(t))
            lref.backward()
            opt.step()
            assert abs(loss - float(lref)) <= 2e-6 * (it + 1) + 1e-6, (it, loss, float(lref))
            it += 1
    assert engine.epoch == 2 and engine.iterations == 4
    # the inputs the model built for the LAST batch equal the oracle's, bit for bit
    assert np.array_equal(engine.model.input.cpu().numpy(), x) and np.array_equal(engine.model.target.cpu().numpy(), t)
    got = engine.model.netG.state_dict()
    for k, v in params.items():
        assert float(((got[k].cpu() - sd[k]) - (v.detach() - sd[k])).abs().max()) < 4e-5, k


def test_burst_synthesis_on_device(eld_lib, lmdb_stub, tmp_path):
    """num_burst frames share ONE parameter draw and differ in their variates; concatenated on the channel axis
    (sid_dataset.py:267-273); the 8-channel input trains a 8 -> 4 U-Net."""
    import eld_amd.noise as noise
    from eld_amd import data as datasets
    from eld_amd.engine import Engine
    np.random.seed(1)
    torch.manual_seed(1)
    with contextlib.redirect_stdout(io.StringIO()):
        nm = noise.NoiseModel(model='Pg', include=4)
    clean = datasets.LMDBDataset('data/Train/SID_Sony_Raw.db')
    ds = datasets.ELDTrainDataset(clean, [datasets.SynDataset(clean, noise_maker=nm, num_burst=2)])
    batch = next(iter(torch.utils.data.DataLoader(ds, batch_size=2, shuffle=False, num_workers=0)))
    engine = Engine(make_opt(tmp_path))             # the burst count comes from the SynDataset the script built: 2 x 4 input planes
    m = engine.model
    m.set_input(batch, 'train')
    assert tuple(m.input.shape) == (2, 8, 512, 512) and tuple(m.target.shape) == (2, 4, 512, 512)
    x, t = oracle_batch(batch, 'Pg', m.seed, [0, 1, 2, 3], burst=2)
    assert np.array_equal(m.input.cpu().numpy(), x) and np.array_equal(m.target.cpu().numpy(), t)
    assert not np.array_equal(x[:, :4], x[:, 4:])            # independent noise per burst frame
    m.optimize_parameters()
    loss = m.get_current_errors()['Pixel']
    sd = {k: v.detach().cpu() for k, v in m.netG.state_dict().items()}
    assert sd['conv1_1.weight'].shape == (32, 8, 3, 3)
    assert np.isfinite(loss) and 0 < loss < 1


def test_model_test_and_engine_test(eld_lib, tmp_path):
    """ELDModel.test / Engine.test (ELD_model.py:309-350, engine.py:101-107): forward on data['input'], no target."""
    from eld_amd.engine import Engine
    torch.manual_seed(0)
    engine = Engine(make_opt(tmp_path, isTrain=False))
    x = torch.rand(1, 4, 64, 96)
    outs = engine.test([{'input': x, 'fn': ['scene_0001.ARW']}], savedir=str(tmp_path / 'out'))
    assert len(outs) == 1 and tuple(outs[0].shape) == (1, 4, 64, 96)
    with torch.no_grad():
        ref = U.unet_forward({k: v.cpu() for k, v in engine.model.netG.state_dict().items()}, x)
    assert float((outs[0].cpu() - ref).abs().max()) <= 1e-5
    saved = np.load(tmp_path / 'out' / 'scene_0001' / 't.npy')
    assert np.array_equal(saved, outs[0][0].cpu().numpy())
    assert engine.test([{'input': x, 'fn': ['scene_0001.ARW']}], savedir=str(tmp_path / 'out')) == [None]   # already there: skipped (ELD_model.py:321-324)
    with pytest.raises(KeyError):
        engine.model.set_input({'target': x}, 'test')


def test_eight_forked_workers_feed_the_device_path(eld_lib, lmdb_stub, tmp_path):
    """The reference default --nThreads 8 (base_option.py:26): eight forked workers, HIP stays in the training process."""
    import eld_amd.noise as noise
    from eld_amd import data as datasets
    from eld_amd.engine import Engine
    with contextlib.redirect_stdout(io.StringIO()):
        nm = noise.NoiseModel(model='PGRU', include=4)
    clean = datasets.LMDBDataset('data/Train/SID_Sony_Raw.db', repeat=4)
    ds = datasets.ELDTrainDataset(clean, [datasets.SynDataset(clean, noise_maker=nm, repeat=4)])
    engine = Engine(make_opt(tmp_path))
    torch.zeros(1, device='cuda')                            # the parent holds a live HIP context when the workers fork
    loader = torch.utils.data.DataLoader(ds, batch_size=4, shuffle=True, num_workers=8, pin_memory=True, worker_init_fn=datasets.worker_init_fn)
    meters = engine.train(loader)
    assert engine.iterations == 4 and 0 < meters['Pixel'] < 1


def test_checkpoint_keeps_the_noise_stream_position(eld_lib, tmp_path):
    from eld_amd.engine import Engine
    import eld_amd.noise as noise
    with contextlib.redirect_stdout(io.StringIO()):
        noise.NoiseModel(model='Pg', include=4)
    e1 = Engine(make_opt(tmp_path, no_log=False, seed=99))
    t = torch.rand(2, 4, 32, 48)
    e1.model.set_input({'target': t}, 'train')
    e1.model.optimize_parameters()
    assert e1.model._sample_counter == 2 and e1.model.seed == 99
    e1.model.save(label='latest')
    e2 = Engine(make_opt(tmp_path, resume=True, resume_epoch=None, seed=5))
    assert e2.model._sample_counter == 2 and e2.model.seed == 99
    sd = torch.load(os.path.join(str(tmp_path), 't', 'model_latest.pt'), map_location='cpu')
    assert {'netG', 'opt_g', 'epoch', 'iterations'} <= set(sd)      # the reference's keys (ELD_model.py:516-523) are all there


def test_srgb_input_stage_on_device(eld_lib, lmdb_stub, tmp_path):
    """--stage_in srgb / --stage_out srgb (train_syn.py:48-58, sid_dataset.py:287-319): noise on the raw patch -> clip -> raw2rgb
    (util/process.py:52-68) -> clip, all on the device, equals the oracle chain bit for bit; a 3 -> 3 U-Net trains on it."""
    import eld_amd.noise as noise
    from eld_amd import _lib as L
    from eld_amd import data as datasets
    from eld_amd.data import records_from_batch
    from eld_amd.engine import Engine
    from eld_amd.noise import model_flags, sample_noise_records, set_sample_ids
    from oracle import isp_ref as I
    np.random.seed(4)
    torch.manual_seed(4)
    with contextlib.redirect_stdout(io.StringIO()):
        nm = noise.NoiseModel(model='Pg', include=4)
    raw_db = datasets.LMDBDataset('data/Train/SID_Sony_Raw.db')
    rng = np.random.default_rng(0)
    meta = [(np.array([2.0 + 0.1 * i, 1.0, 1.5, 1.0], np.float32), (np.eye(3) * (1.3 + 0.05 * i) - 0.1).astype(np.float32)) for i in range(4)]
    srgb_targets = [rng.uniform(0, 1, (3, 512, 512)).astype(np.float32) for _ in range(4)]

    class ListDS(object):
        def __getitem__(self, i):
            return srgb_targets[i % 4]

        def __len__(self):
            return 4
    input_data = datasets.ISPDataset(raw_db, noise_maker=nm, meta_info=meta)                              # train_syn.py:55-58
    ds = datasets.ELDTrainDataset(target_dataset=ListDS(), input_datasets=[input_data])
    batch = next(iter(torch.utils.data.DataLoader(ds, batch_size=2, shuffle=False, num_workers=0)))
    assert set(batch) == {'clean', 'target', 'wb', 'ccm', 'aug', 'burst', 'params'}
    engine = Engine(make_opt(tmp_path, stage_in='srgb', stage_out='srgb'))
    m = engine.model
    m.set_input(batch, 'train')
    assert tuple(m.input.shape) == (2, 3, 512, 512) and tuple(m.target.shape) == (2, 3, 512, 512)
    # oracle chain
    codes = batch['clean'].numpy().view(np.uint16)
    clean = O.lmdb_decode_u16(codes)
    recs = set_sample_ids(records_from_batch(batch['params'].numpy()), [0, 1])
    flags = model_flags('Pg') | L.CLIP
    dump = torch.zeros(L.NPLANES, clean.size, device='cuda')
    sample_noise_records(batch['clean'].cuda(), recs, flags, m.seed, in_u16=True, dump=dump)
    dv = dump.cpu().numpy()
    noisy = np.stack([O.noise_arith(clean[i], oracle_params(recs[i]), flags, **{n: dv[j].reshape(clean.shape)[i] for n, j in L.PLANE.items()})
                      for i in range(2)])
    rgb = I.process(noisy, batch['wb'].numpy(), batch['ccm'].numpy())
    for i in range(2):
        b = int(batch['aug'][i])
        assert np.array_equal(m.input[i].cpu().numpy(), np.clip(O.augment(rgb[i], b & 1, b & 2, b & 4), 0, 1))
        assert np.array_equal(m.target[i].cpu().numpy(), O.augment(batch['target'][i].numpy(), b & 1, b & 2, b & 4))
    m.optimize_parameters()
    assert m.netG.state_dict()['conv1_1.weight'].shape == (32, 3, 3, 3) and m.netG.state_dict()['conv10_1.weight'].shape == (3, 32, 1, 1)
    assert 0 < m.get_current_errors()['Pixel'] < 1


def test_noise_applies_to_syn_datasets_own_patch_and_crf_reaches_the_isp(eld_lib, lmdb_stub, tmp_path):
    """(a) --stage_in raw --stage_out srgb under on-the-fly noise: the target is a 3-channel sRGB image, the noise applies to the
    RAW patch SynDataset read itself (sid_dataset.py:265-275) -> the clean codes travel as 'clean', a 4 -> 3 U-Net trains.
    (b) --crf: the tables handed to ISPDataset(CRF=...) (train_syn.py:42-58) reach the device ISP (process.py:62-66), and --crf with
    no tables anywhere raises instead of rendering with gamma 2.2."""
    import eld_amd.noise as noise
    from eld_amd import _lib as L
    from eld_amd import data as datasets
    from eld_amd.data import records_from_batch
    from eld_amd.engine import Engine
    from eld_amd.noise import model_flags, sample_noise_records, set_sample_ids
    from oracle import isp_ref as I
    np.random.seed(7)
    torch.manual_seed(7)
    with contextlib.redirect_stdout(io.StringIO()):
        nm = noise.NoiseModel(model='Pg', include=4)
    raw_db = datasets.LMDBDataset('data/Train/SID_Sony_Raw.db')
    rng = np.random.default_rng(1)
    srgb_targets = [rng.uniform(0, 1, (3, 512, 512)).astype(np.float32) for _ in range(4)]

    class ListDS(object):
        def __getitem__(self, i):
            return srgb_targets[i % 4]

        def __len__(self):
            return 4
    # ---- (a)
    ds = datasets.ELDTrainDataset(target_dataset=ListDS(), input_datasets=[datasets.SynDataset(raw_db, noise_maker=nm)])
    batch = next(iter(torch.utils.data.DataLoader(ds, batch_size=2, shuffle=False, num_workers=0)))
    assert set(batch) == {'clean', 'target', 'params', 'aug', 'burst'} and batch['clean'].dtype == torch.int16
    m = Engine(make_opt(tmp_path, stage_in='raw', stage_out='srgb')).model
    m.set_input(batch, 'train')
    assert tuple(m.input.shape) == (2, 4, 512, 512) and tuple(m.target.shape) == (2, 3, 512, 512)
    x, _ = oracle_batch({'target': batch['clean'], 'params': batch['params'], 'aug': batch['aug']}, 'Pg', m.seed, [0, 1])
    assert np.array_equal(m.input.cpu().numpy(), x)                       # noise on the raw patch, not on the sRGB target
    for i in range(2):
        b = int(batch['aug'][i])
        assert np.array_equal(m.target[i].cpu().numpy(), O.augment(batch['target'][i].numpy(), b & 1, b & 2, b & 4))
    m.optimize_parameters()
    assert m.netG.state_dict()['conv1_1.weight'].shape == (32, 4, 3, 3) and m.netG.state_dict()['conv10_1.weight'].shape == (3, 32, 1, 1)
    # ---- (b)
    E = np.linspace(0, 1, 33).astype(np.float32)
    crf = (E, (E ** 0.3).astype(np.float32))
    meta = [(np.array([2.0, 1.0, 1.5, 1.0], np.float32), np.eye(3, dtype=np.float32))] * 4
    isp_ds = datasets.ISPDataset(raw_db, noise_maker=nm, meta_info=meta, CRF=crf)
    ds = datasets.ELDTrainDataset(target_dataset=ListDS(), input_datasets=[isp_ds], augment=False)
    batch = next(iter(torch.utils.data.DataLoader(ds, batch_size=1, shuffle=False, num_workers=0)))
    m = Engine(make_opt(tmp_path, stage_in='srgb', stage_out='srgb', crf=True)).model
    assert m.CRF is crf
    m.set_input(batch, 'train')
    codes = batch['clean'].numpy().view(np.uint16)
    clean = O.lmdb_decode_u16(codes)
    recs = set_sample_ids(records_from_batch(batch['params'].numpy()), [0])
    flags = model_flags('Pg') | L.CLIP
    dump = torch.zeros(L.NPLANES, clean.size, device='cuda')
    sample_noise_records(batch['clean'].cuda(), recs, flags, m.seed, in_u16=True, dump=dump)
    dv = dump.cpu().numpy()
    noisy = np.stack([O.noise_arith(clean[0], oracle_params(recs[0]), flags, **{n: dv[j].reshape(clean.shape)[0] for n, j in L.PLANE.items()})])
    want = I.process(noisy, batch['wb'].numpy(), batch['ccm'].numpy(), CRF=crf)
    gamma = I.process(noisy, batch['wb'].numpy(), batch['ccm'].numpy())
    got = m.input.cpu().numpy()
    assert float(np.mean(got != want)) < 1e-4 and float(np.mean(want != gamma)) > 0.5      # the CRF render, not gamma 2.2 (isolated quantiser flips: oracle/isp_ref.py)
    del isp_ds, ds, batch                                                 # the tables die with the dataset (weak reference)
    with pytest.raises(RuntimeError, match='crf'):
        Engine(make_opt(tmp_path, stage_in='srgb', stage_out='srgb', crf=True))
