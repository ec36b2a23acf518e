"""GPU tests of the model / engine plugins: the fused training iteration (sampler -> U-Net -> L1 -> backward ->
Adam) against the CPU oracle, checkpoint interchange with torch's Adam, the autograd drop-in path, data-parallel
semantics and the epoch driver.  Reference surface: models/ELD_model.py:172-523, engine.py:10-128."""
import os
import types

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')

from oracle import noise_ref as O      # noqa: E402  (checker only)
from oracle import unet_ref as U       # noqa: E402


def make_opt(tmp, **kw):
    d = dict(gpu_ids=[0], isTrain=True, checkpoints_dir=str(tmp), name='t', netG='unet', channels=4, stage_in='raw', stage_out='raw',
             lr=1e-4, beta1=0.9, wd=0.0, loss='l1', resume=False, chop=False, no_log=False, save_epoch_freq=2, model='eld_model')
    d.update(kw)
    return types.SimpleNamespace(**d)


def new_model(tmp, seed=2018, **kw):
    from eld_amd.model import ELDModel
    torch.manual_seed(seed)
    m = ELDModel()
    m.initialize(make_opt(tmp, **kw))
    return m


def batch(shape=(2, 4, 32, 48), seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(*shape, generator=g), torch.rand(*shape, generator=g)


def test_training_steps_match_oracle(eld_lib, tmp_path):
    """3 fused iterations == 3 iterations of (torch-CPU forward, L1, backward, torch.optim.Adam) from the same init."""
    m = new_model(tmp_path)
    sd = {k: v.detach().cpu().clone() for k, v in m.netG.state_dict().items()}
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    opt = torch.optim.Adam(list(params.values()), lr=1e-4, betas=(0.9, 0.999))
    for it in range(3):
        x, t = batch(seed=it)
        m.set_input({'input': x, 'target': t}, 'train')
        m.optimize_parameters()
        loss = m.get_current_errors()['Pixel']
        opt.zero_grad()
        lref = torch.nn.functional.l1_loss(U.unet_forward(params, x), t)
        lref.backward()
        opt.step()
        assert abs(loss - float(lref)) < 2e-6 * (it + 1) + 1e-6, (it, loss, float(lref))
    got = m.netG.state_dict()
    for k, v in params.items():
        # Adam's first steps move every weight by ~lr regardless of gradient scale: compare the UPDATE, sign-stable part
        d_ref = v.detach() - sd[k]
        d_got = got[k].cpu() - sd[k]
        assert float((d_got - d_ref).abs().max()) < 3e-5, k       # |update| <= 3e-4 after 3 steps; tiny-gradient elements may flip


def test_on_device_synthesis_matches_sampler_and_oracle(eld_lib, tmp_path):
    from eld_amd.noise import NoiseModel, NoiseParams, sample_noise, model_flags
    from eld_amd import _lib as L
    m = new_model(tmp_path)
    nm = NoiseModel(model='PGRU', include=4)
    m.set_noise_model(nm)
    t = (torch.floor(65535 * torch.rand(2, 4, 32, 48) ** 2.2) / 65535)
    p = [NoiseParams(2.288, 6.451, 15583, 208.98, tl_lambda=-0.14285714, tl_scale=3.3, row_scale=0.9)] * 2
    m.set_input({'target': t, 'params': p, 'sample_ids': [7, 8]}, 'train')
    flags = model_flags('PGRU') | L.CLIP
    dump = torch.zeros(L.NPLANES, t.numel(), device='cuda')
    z = sample_noise(t.cuda(), p, flags, nm.seed, [7, 8], dump=dump)
    assert torch.equal(m.input, z) and float(m.input.min()) >= 0 and float(m.input.max()) <= 1
    dv = dump.cpu().numpy()
    op = O.Params(K=2.288, g_scale=6.451, ratio=208.98, tl_lambda=-0.14285714, tl_scale=3.3, row_scale=0.9)
    for i in range(2):
        v = {k: dv[j].reshape(t.shape)[i] for k, j in L.PLANE.items()}
        assert np.array_equal(z[i].cpu().numpy(), O.noise_arith(t[i].numpy(), op, flags, **v))
    # default ids are global, rank-strided and advance with the step
    np.random.seed(1)
    m.set_input({'target': t}, 'train')
    a = m.input.clone()
    m.set_input({'target': t}, 'train')
    assert not torch.equal(a, m.input)
    m.optimize_parameters()
    assert np.isfinite(m.get_current_errors()['Pixel'])


def test_checkpoint_interchange_with_torch_adam(eld_lib, tmp_path):
    from eld_amd.unet import UNetSeeInDark
    m = new_model(tmp_path)
    x, t = batch()
    for _ in range(2):
        m.set_input({'input': x, 'target': t}, 'train')
        m.optimize_parameters()
    m.epoch, m.iterations = 3, 77
    m.save(label='latest')
    path = os.path.join(str(tmp_path), 't', 'model_latest.pt')
    sd = torch.load(path, map_location='cpu')
    assert set(sd) == {'netG', 'opt_g', 'epoch', 'iterations', 'eld_amd'} and sd['epoch'] == 3 and sd['iterations'] == 77      # the reference's four + the sampler position
    assert len(sd['netG']) == 46 and sd['netG']['upv6.weight'].shape == (512, 256, 2, 2)
    # (a) the reference's own way of loading it: plain module + torch.optim.Adam (ELD_model.py:492-514)
    net = UNetSeeInDark(4, 4)
    net.load_state_dict(sd['netG'])
    topt = torch.optim.Adam(net.parameters(), lr=1e-4, betas=(0.9, 0.999))
    topt.load_state_dict(sd['opt_g'])
    st = topt.state[list(net.parameters())[0]]
    assert float(st['step']) == 2.0 and st['exp_avg'].shape == (32, 4, 3, 3)
    # (b) resume in a fresh fused model and continue identically
    m2 = new_model(tmp_path, seed=1, resume=True, resume_epoch=None)
    assert m2.epoch == 3 and m2.iterations == 77 and m2.optimizer_G.step_count == 2
    assert torch.equal(m2.netG.flat_params, m.netG.flat_params) and torch.equal(m2.optimizer_G.exp_avg, m.optimizer_G.exp_avg)
    for mm in (m, m2):
        mm.set_input({'input': x, 'target': t}, 'train')
        mm.optimize_parameters()
    assert torch.equal(m2.netG.flat_params, m.netG.flat_params)
    # (c) a torch-Adam state dict loads into the fused optimizer
    m.optimizer_G.load_state_dict(topt.state_dict())
    assert m.optimizer_G.step_count == 2


def test_autograd_dropin_path_equals_fused_path(eld_lib, tmp_path):
    """arch plugin inside the reference's own loop shape: netG(x) -> nn.L1Loss -> backward -> torch.optim.Adam.step."""
    from eld_amd.unet import unet
    torch.manual_seed(5)
    net = unet(4, 4).cuda()
    m = new_model(tmp_path, seed=5)
    assert torch.equal(net.flat_params, m.netG.flat_params)
    opt = torch.optim.Adam(net.parameters(), lr=1e-4, betas=(0.9, 0.999), weight_decay=0.0)
    crit = torch.nn.L1Loss()
    for it in range(2):
        x, t = batch(seed=10 + it)
        out = net(x.cuda())
        opt.zero_grad()
        loss = crit(out, t.cuda())
        loss.backward()
        opt.step()
        m.set_input({'input': x, 'target': t}, 'train')
        m.optimize_parameters()
        assert abs(float(loss) - m.get_current_errors()['Pixel']) < 1e-6
        if it == 0:
            assert torch.equal(m.output, out.detach())          # same engine, same weights: same bits
        else:                                                   # torch Adam vs fused Adam differ by a few ulp per weight
            assert float((m.output - out.detach()).abs().max()) < 1e-5
    assert float((net.flat_params - m.netG.flat_params).abs().max()) < 1e-6


def test_data_parallel_semantics_on_one_gpu(eld_lib, tmp_path):
    """sum-all-reduce of per-rank gradients with grad_scale = 1/world == gradient of the global-batch mean loss
    (equal shard sizes): checked by linearity of the engine's backward on one device."""
    m = new_model(tmp_path)
    x, t = batch(shape=(2, 4, 32, 32))
    net = m.netG
    def grads(xb, tb):
        out, key, _ = net._engine_forward(xb.cuda(), save=True)
        dout = torch.sign(out - tb.cuda()) / out.numel()
        return net._engine_backward(dout.contiguous(), key, tuple(xb.shape)).clone()
    g_full = grads(x, t)
    g0, g1 = grads(x[:1], t[:1]), grads(x[1:], t[1:])
    ref = 0.5 * (g0 + g1)
    assert float((g_full - ref).abs().max()) <= 1e-6 + 1e-4 * float(ref.abs().max())


def test_forward_chop_and_eval_psnr(eld_lib, tmp_path):
    m = new_model(tmp_path, chop=True)
    sd = {k: v.detach().cpu() for k, v in m.netG.state_dict().items()}
    x, t = batch(shape=(1, 4, 64, 96), seed=4)
    m.set_input({'input': x, 'target': t, 'fn': ['a']}, 'eval')
    with torch.no_grad():
        out = m.forward().cpu()
    # oracle chop (ELD_model.py:434-467): h_half=32 -> h_size 48; w_half=48 -> w_size 64
    hs, ws_ = 48, 64
    tiles = [x[:, :, :hs, :ws_], x[:, :, :hs, 96 - ws_:], x[:, :, 64 - hs:, :ws_], x[:, :, 64 - hs:, 96 - ws_:]]
    with torch.no_grad():
        o = [U.unet_forward(sd, tt) for tt in tiles]
    ref = torch.empty(1, 4, 64, 96)
    ref[:, :, :32, :48] = o[0][:, :, :32, :48]
    ref[:, :, :32, 48:] = o[1][:, :, :32, ws_ - 48:]
    ref[:, :, 32:, :48] = o[2][:, :, hs - 32:, :48]
    ref[:, :, 32:, 48:] = o[3][:, :, hs - 32:, ws_ - 48:]
    assert float((out - ref).abs().max()) < 1e-5
    m.opt.chop = False
    r = m.eval({'input': x, 'target': t, 'fn': ['a']}, crop=False, correct=True)
    with torch.no_grad():
        pred = torch.clamp(U.unet_forward(sd, x), 0, 1)
    alpha = torch.dot(pred[t != 1], t[t != 1]) / torch.dot(pred[t != 1], pred[t != 1])
    a = np.clip((alpha * pred)[0].numpy() * 255.0, 0, 255)
    b = np.clip(t[0].numpy() * 255.0, 0, 255)
    psnr = 10 * np.log10(255.0 ** 2 / np.mean((a.astype(np.float64) - b) ** 2))
    assert abs(r['PSNR'] - psnr) < 1e-3
    from oracle import metrics_ref as M
    assert abs(r['SSIM'] - M.ssim(a, b)) < 1e-6 and abs(r['PSNR'] - M.psnr(b, a)) < 1e-3


def test_metrics_on_device_vs_oracle(eld_lib):
    """PSNR / SSIM as util/index.py:76-81 computes them (skimage defaults restated in oracle/metrics_ref.py)."""
    from eld_amd.metrics import quality_assess
    from oracle import metrics_ref as M
    g = torch.Generator().manual_seed(0)
    for shape, noise in (((4, 64, 80), 5.0), ((4, 37, 41), 30.0), ((3, 128, 128), 0.5)):
        y = torch.rand(*shape, generator=g) * 255
        x = torch.clamp(y + noise * torch.randn(*shape, generator=g), 0, 255)
        r = quality_assess(x.cuda(), y.cuda())
        assert abs(r['PSNR'] - M.psnr(y.numpy(), x.numpy())) < 1e-4
        assert abs(r['SSIM'] - M.ssim(y.numpy(), x.numpy())) < 1e-6
    assert abs(quality_assess(y.cuda(), y.cuda())['SSIM'] - 1.0) < 1e-12
    host = quality_assess(x, y)                   # host tensors (the reference's callers hold host arrays): moved to the GPU, same kernels
    assert host == quality_assess(x.cuda(), y.cuda())
    hwc = quality_assess(x.permute(1, 2, 0).numpy(), y.permute(1, 2, 0).numpy())      # tensor2im's HWC ndarray layout (ELD_model.py:23-38)
    assert hwc == host


def test_engine_train_loop(eld_lib, tmp_path, capsys):
    from eld_amd.engine import Engine
    torch.manual_seed(0)
    eng = Engine(make_opt(tmp_path))
    assert eng.epoch == 0 and eng.iterations == 0
    eng.set_learning_rate(5e-5)
    assert eng.model.optimizers[0].param_groups[0]['lr'] == 5e-5
    loader = [dict(zip(('input', 'target'), batch(shape=(1, 4, 32, 32), seed=s))) for s in range(3)]
    meters = eng.train(loader)
    assert eng.epoch == 1 and eng.iterations == 3 and np.isfinite(meters['Pixel'])
    first = meters['Pixel']
    for _ in range(3):
        meters = eng.train(loader)
    assert eng.epoch == 4 and eng.iterations == 12
    assert meters['Pixel'] < first                                    # it learns
    files = sorted(os.listdir(os.path.join(str(tmp_path), 't')))
    assert 'model_latest.pt' in files and any(f.startswith('model_002_') for f in files) and any(f.startswith('model_004_') for f in files)
    out = capsys.readouterr().out
    assert 'learning rate = 0.0000500' in out and 'Epoch: 0' in out


def test_prefetched_synthesis_equals_the_serial_order_bit_for_bit(eld_lib, tmp_path, monkeypatch):
    """Engine.train's one-batch lookahead (ELDModel.prefetch_input: the sampler launch of batch i+1 on the synthesis stream beside iteration i's
    U-Net kernels -- the reference overlaps synthesis and training through DataLoader workers, train_syn.py:78-80) changes no bit: the same
    host draws in the same order, the same sample ids, the same parameters after every epoch as the plain loop (ELD_AMD_PREFETCH=0)."""
    from eld_amd.engine import Engine
    from eld_amd.noise import NoiseModel
    import io
    import contextlib

    def run(prefetch):
        monkeypatch.setenv('ELD_AMD_PREFETCH', prefetch)
        np.random.seed(7)
        torch.manual_seed(0)
        with contextlib.redirect_stdout(io.StringIO()):
            nm = NoiseModel(model='PGRU', include=4)
            eng = Engine(make_opt(tmp_path / ('pf' + prefetch), no_log=True))
        eng.model.set_noise_model(nm)
        g = torch.Generator().manual_seed(3)
        # deferred batches: clean targets only, parameters drawn by the model at synthesis time (np.random order matters), ragged last batch
        loader = [{'target': torch.floor(65535.0 * torch.rand(n, 4, 48, 64, generator=g) ** 2.2) / 65535.0} for n in (2, 2, 1, 2, 1)]
        inputs, losses = [], []
        keep = eng.model.optimize_parameters

        def spy(**kw):
            inputs.append(eng.model.input.clone())
            keep(**kw)
        eng.model.optimize_parameters = spy
        for _ in range(2):
            with contextlib.redirect_stdout(io.StringIO()):
                losses.append(eng.train(loader)['Pixel'])
        torch.cuda.synchronize()
        return inputs, losses, eng.model.netG.flat_params.detach().clone(), eng.model._sample_counter
    a, b = run('1'), run('0')
    assert len(a[0]) == len(b[0]) == 10 and a[3] == b[3] == 16
    for x, y in zip(a[0], b[0]):
        assert torch.equal(x, y)
    assert a[1] == b[1]
    assert torch.equal(a[2], b[2])
    # a prefetch nobody picks up is dropped, and set_input of another batch is the plain path
    from eld_amd.model import ELDModel  # noqa: F401
    m = new_model(tmp_path / 'drop')
    with contextlib.redirect_stdout(io.StringIO()):
        m.set_noise_model(NoiseModel(model='Pg', include=4))
    d1, d2 = {'target': torch.rand(1, 4, 32, 32)}, {'input': torch.rand(1, 4, 32, 32), 'target': torch.rand(1, 4, 32, 32)}
    m.prefetch_input(d1)
    m.set_input(d2)
    assert m._prefetched is None and torch.equal(m.input.cpu(), d2['input'])
    m.optimize_parameters()
    assert np.isfinite(m.get_current_errors()['Pixel'])


def test_model_requires_gpu_and_raw_stage(eld_lib, tmp_path):
    from eld_amd.model import ELDModel
    with pytest.raises(RuntimeError):
        ELDModel().initialize(make_opt(tmp_path, gpu_ids=[]))
    with pytest.raises(NotImplementedError):                          # ELD_model.py:377-389: 'Invalid Stage'
        ELDModel().initialize(make_opt(tmp_path, stage_in='xyz'))
    m = ELDModel()
    m.initialize(make_opt(tmp_path, stage_in='srgb'))                 # wired since n4: 3-channel input net
    assert m.netG.conv1_1.weight.shape[1] == 3


def test_bf16_training_tracks_fp32(eld_lib, tmp_path):
    """BASELINE config 3 precision through the model plugin: same init, same batches -> the bf16 loss curve stays within
    1 % of the fp32 one over 6 iterations and both decrease; master weights stay fp32."""
    losses = {}
    for prec in ('fp32', 'bf16'):
        m = new_model(tmp_path, seed=3, precision=prec)
        assert m.netG.train_precision == prec and m.netG.flat_params.dtype == torch.float32
        ls = []
        for it in range(6):
            x, t = batch(shape=(2, 4, 64, 64), seed=it % 2)
            m.set_input({'input': x, 'target': t}, 'train')
            m.optimize_parameters()
            ls.append(m.get_current_errors()['Pixel'])
        losses[prec] = ls
    a, b = np.array(losses['fp32']), np.array(losses['bf16'])
    assert np.all(np.abs(a - b) / a < 0.01), (a, b)
    assert a[-1] < a[0] and b[-1] < b[0]


def test_backward_bucket_events(eld_lib, tmp_path):
    """eld_unet_backward_buckets: same gradients as the plain backward (bitwise), one event per bucket recorded in
    top-down order, bad bucket tables rejected."""
    import ctypes
    from eld_amd import _lib as L
    from eld_amd import dist as D
    m = new_model(tmp_path)
    net = m.netG
    x, t = batch(shape=(1, 4, 32, 48))
    out, key, _ = net._engine_forward(x.cuda(), save=True)
    dout = (torch.sign(out - t.cuda()) / out.numel()).contiguous()
    ref = net._engine_backward(dout, key, tuple(x.shape)).clone()
    n = ref.numel()
    bk = D.GradBuckets(n, ref.device, bucket=1 << 20)
    assert bk.n == 8 and bk.starts[0] == 0
    got = torch.zeros_like(ref)
    net._engine_backward(dout, key, tuple(x.shape), grads=got, buckets=bk)
    torch.cuda.synchronize()
    assert all(e.query() for e in bk.events)
    assert torch.equal(got, ref)
    lib = eld_lib
    ws = net._ws.bufs[key]
    bad = (ctypes.c_int64 * 2)(5, 5)
    rc = lib.eld_unet_backward_buckets(L.dptr(dout), L.dptr(net.flat_params), L.dptr(got), L.dptr(ws), ws.numel(), 1, 32, 48, 4, 4, 0,
                                       bad, bk.events_c, 2, L.cur_stream())
    assert rc == -1
    assert D.GradBuckets(n, ref.device).allreduce_sum_(got) == 1        # world size 1: nothing to exchange


def _dp_worker(rank, world, port, tmp, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0',
                      ELD_DIST_BACKEND='gloo')
    import torch as T
    from eld_amd import dist as D
    D.init()
    T.manual_seed(2018 + 7 * rank)                  # replicas start different on purpose: rank 0's weights are broadcast
    from eld_amd.model import ELDModel
    m = ELDModel()
    m.initialize(make_opt(os.path.join(tmp, 'r%d' % rank)))
    for it in range(2):
        x, t = batch(shape=(2, 4, 32, 48), seed=it)
        m.set_input({'input': x[rank::world], 'target': t[rank::world]}, 'train')
        m.optimize_parameters()
        loss = m.get_current_errors()['Pixel']
    q.put((rank, loss, m.netG.flat_params.detach().cpu().numpy().copy(), m._buckets is not None))
    T.distributed.barrier()
    T.distributed.destroy_process_group()


def test_two_rank_training_equals_global_batch(eld_lib, tmp_path):
    """Two processes (gloo over CUDA tensors, both on this GPU), one image each, bucketed overlapped all-reduce ==
    one process stepping on the 2-image batch, after 2 iterations."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(2)], key=lambda r: r[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    m = new_model(tmp_path)                          # seed 2018 == rank 0's initial weights
    for it in range(2):
        x, t = batch(shape=(2, 4, 32, 48), seed=it)
        m.set_input({'input': x, 'target': t}, 'train')
        m.optimize_parameters()
        loss = m.get_current_errors()['Pixel']
    ref = m.netG.flat_params.detach().cpu().numpy()
    assert res[0][3] and res[1][3]                   # the bucketed path ran
    assert np.array_equal(res[0][2], res[1][2])      # replicas stay identical
    assert abs(res[0][1] - loss) < 1e-6 and abs(res[1][1] - loss) < 1e-6
    assert np.abs(res[0][2] - ref).max() < 3e-5      # Adam moves every weight ~lr per step; sign-unstable tiny gradients may differ


@pytest.mark.parametrize('variant', [0, 32, 64], ids=['strip2', 'strip1', 'tiles'])
@pytest.mark.parametrize('shape', [(2, 4, 64, 80), (1, 4, 37, 41), (3, 3, 7, 129), (1, 4, 512, 512), (1, 4, 230, 300), (2, 3, 113, 257), (1, 2, 1424, 2128)])
def test_quality_assess_kernel_vs_oracle(eld_lib, shape, variant):
    """csrc/eval.hip eld_quality_assess == tensor2im + PSNR + SSIM of oracle/metrics_ref.py (float64), per image.  Round 5: the one-pass strip kernel
    (qa_fused_kernel, two window columns per lane), its one-column variant and the round-2 tile kernels (eld_debug_kernel_mask bits 5 / 6), on shapes
    that cross its 106-row chunks and 128- / 64-column strips, leave a strip almost empty (W = 129, 257) or have fewer rows than one window needs + 1."""
    from eld_amd.metrics import quality_assess_frames
    from oracle import metrics_ref as M
    g = torch.Generator().manual_seed(shape[2] * shape[3])
    ref = torch.rand(*shape, generator=g) * 1.1 - 0.05                 # some values outside [0,1]: the clip matters
    est = ref + 0.05 * torch.randn(*shape, generator=g)
    prev = eld_lib.eld_debug_kernel_mask(variant)
    try:
        q = quality_assess_frames(est.cuda(), ref.cuda()).cpu().numpy()
        q2 = quality_assess_frames(est.cuda(), ref.cuda()).cpu().numpy()
        same = quality_assess_frames(ref.cuda(), ref.cuda()).cpu().numpy()
    finally:
        eld_lib.eld_debug_kernel_mask(prev)
    assert np.array_equal(q, q2)                                       # fixed reduction order: run-to-run the same bits
    for n in range(shape[0]):
        a, b = M.tensor2im(est[n].numpy()), M.tensor2im(ref[n].numpy())
        assert abs(q[n, 0] - M.psnr(b, a)) < 1e-9
        assert abs(q[n, 1] - M.ssim(b, a)) < 1e-9
    assert np.all(np.isinf(same[:, 0])) and np.all(np.abs(same[:, 1] - 1.0) < 1e-12)


@pytest.mark.parametrize('shape,one_source', [((2, 4, 32, 48), False), ((3, 4, 16, 16), True), ((1, 4, 512, 512), False)])
def test_illuminance_correct_kernel_vs_oracle(eld_lib, shape, one_source):
    from eld_amd.model import illuminance_correct
    from oracle import metrics_ref as M
    g = torch.Generator().manual_seed(3)
    pred = torch.rand(*shape, generator=g) * 1.4 - 0.2
    src = torch.rand(*((1,) + shape[1:] if one_source else shape), generator=g)
    src[src > 0.9] = 1.0                                               # saturated pixels are excluded from the fit
    got = illuminance_correct(pred.cuda(), src.cuda()).cpu().numpy()
    ref = M.illuminance_correct(pred.numpy(), src.numpy())
    assert np.abs(got - ref).max() <= 2e-7 * max(1.0, float(np.abs(ref).max()))


def test_eval_kernels_vs_reference_minted_fixture(eld_lib, tmp_path, golden_dir):
    """SURVEY 8(f) n2 pinned: csrc/eval.hip and ELDModel.forward_chop against tests/golden/eval.npz, which oracle/gen_golden_eval.py
    minted by running the reference's IlluminanceCorrect (ELD_model.py:138-169), tensor2im (:23-38) and forward_chop (:434-467)."""
    from eld_amd.metrics import quality_assess_frames
    from eld_amd.model import illuminance_correct
    from oracle import metrics_ref as M
    d = np.load(os.path.join(golden_dir, 'eval.npz'))
    for tag in ('ic_n', 'ic_one', 'ic_b1'):          # own sources / one shared source frame / batch 1 with saturated pixels
        got = illuminance_correct(torch.from_numpy(d[tag + '_pred']).cuda(), torch.from_numpy(d[tag + '_src']).cuda()).cpu().numpy()
        ref = d[tag + '_out']
        assert np.abs(got - ref).max() <= 2e-7 * float(np.abs(ref).max()), tag        # the gain: double accumulation vs float32 torch.dot
    # tensor2im fused in the quality kernel: PSNR / SSIM of ([0,1]-unit frames) == the metrics of the reference's tensor2im images
    x = torch.from_numpy(d['t2i_in']).cuda()
    q = quality_assess_frames(x[:1], x[1:2]).cpu().numpy()[0]
    a, b = np.transpose(d['t2i_out'], (2, 0, 1)), np.transpose(d['t2i_out1'], (2, 0, 1))
    assert abs(q[0] - M.psnr(b, a)) < 1e-9 and abs(q[1] - M.ssim(b, a)) < 1e-9
    # forward_chop: the reference's seeded default init (same torch build) -> the reference's own chopped output
    m = new_model(tmp_path, chop=True)
    wsum = np.array([float(p.detach().double().cpu().sum()) for p in m.netG.parameters()])
    if not np.array_equal(wsum, d['wsum']):
        pytest.skip('seeded default init differs from the minting torch build')
    for tag in ('chop_a', 'chop_b'):                 # shave < 10 -> +16 (64x96) and shave >= 10 (44x74) branches of :441-442
        xx = torch.from_numpy(d[tag + '_x'])
        m.set_input({'input': xx, 'target': xx, 'fn': ['a']}, 'eval')
        with torch.no_grad():
            out = m.forward().cpu().numpy()
        assert out.shape == d[tag + '_out'].shape
        assert np.abs(out - d[tag + '_out']).max() < 1e-5, tag


def test_l2_loss_training_step(eld_lib, tmp_path):
    """--loss l2 (train_options / models/losses.py:34): one fused step == torch-CPU forward, MSELoss, backward, Adam."""
    m = new_model(tmp_path, loss='l2')
    sd = {k: v.detach().cpu().clone() for k, v in m.netG.state_dict().items()}
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    opt = torch.optim.Adam(list(params.values()), lr=1e-4, betas=(0.9, 0.999))
    x, t = batch(seed=4)
    m.set_input({'input': x, 'target': t}, 'train')
    m.optimize_parameters()
    loss = m.get_current_errors()['Pixel']
    lref = torch.nn.functional.mse_loss(U.unet_forward(params, x), t)
    lref.backward()
    opt.step()
    assert abs(loss - float(lref)) < 1e-6
    got = m.netG.state_dict()
    for k, v in params.items():
        assert float(((got[k].cpu() - sd[k]) - (v.detach() - sd[k])).abs().max()) < 2e-5, k


def test_unet_step_is_hipgraph_capturable(eld_lib):
    """include/eld_amd.h promises entry points without allocation or synchronisation: a forward + backward + Adam chain recorded into a HIP graph
    (torch.cuda.CUDAGraph = hipGraph on ROCm) replays to the same bits as the eager calls, also after the inputs change in place."""
    from eld_amd import _lib as L
    from eld_amd.unet import UNetSeeInDark
    torch.manual_seed(21)
    net = UNetSeeInDark(4, 4).cuda()
    g = torch.Generator(device='cuda').manual_seed(5)
    shape = (2, 4, 64, 96)
    x = torch.rand(*shape, device='cuda', generator=g)
    dout = torch.randn(*shape, device='cuda', generator=g) / x.numel()
    nparam = net._offsets[-1]
    lib = L.lib()

    def step(xin, dg, params, m, v, grads):
        out, key, _ = net._engine_forward(xin, save=True)
        net._engine_backward(dg, key, tuple(xin.shape), grads=grads)
        L.check(lib.eld_adam_step(L.dptr(params), L.dptr(grads), L.dptr(m), L.dptr(v), nparam, 1e-4, 0.9, 0.999, 1e-8, 0.0, 1, 1.0, L.cur_stream()), 'eld_adam_step')
        return out

    p0 = net.flat_params.detach().clone()
    # eager reference (also the warm-up that sets kernel attributes and per-device caches before the capture)
    grads_e = torch.empty(nparam, device='cuda'); m_e = torch.zeros(nparam, device='cuda'); v_e = torch.zeros(nparam, device='cuda')
    out_e = step(x, dout, net.flat_params, m_e, v_e, grads_e).clone()
    p_e = net.flat_params.detach().clone()
    net.flat_params.data.copy_(p0)
    torch.cuda.synchronize()
    # capture the same chain
    grads_g = torch.empty(nparam, device='cuda'); m_g = torch.zeros(nparam, device='cuda'); v_g = torch.zeros(nparam, device='cuda')
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out_g = step(x, dout, net.flat_params, m_g, v_g, grads_g)
    net.flat_params.data.copy_(p0); m_g.zero_(); v_g.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out_g, out_e) and torch.equal(grads_g, grads_e) and torch.equal(net.flat_params.detach(), p_e)
    # second replay on new data written in place: equals a fresh eager step from the same state
    x2 = torch.rand(*shape, device='cuda', generator=g)
    p1, m1, v1 = net.flat_params.detach().clone(), m_g.clone(), v_g.clone()
    x.copy_(x2)
    graph.replay()
    torch.cuda.synchronize()
    out_r, grads_r, p_r = out_g.clone(), grads_g.clone(), net.flat_params.detach().clone()
    net.flat_params.data.copy_(p1)
    grads_e2 = torch.empty(nparam, device='cuda')
    out_e2 = step(x2, dout, net.flat_params, m1, v1, grads_e2)
    torch.cuda.synchronize()
    assert torch.equal(out_r, out_e2) and torch.equal(grads_r, grads_e2) and torch.equal(net.flat_params.detach(), p_r)


@pytest.mark.parametrize('world,batch', [(2, 1), (8, 8)], ids=['dp2', 'dp8_global_batch_64'])
def test_bench_multi_rank_line_has_the_scaling_keys(eld_lib, world, batch):
    """bench.py's N > 1 path end to end on ONE GPU (the ranks share the device over gloo; RCCL itself needs N devices:
    tests/test_dist_gpu.py): the line the driver's SCALE run parses must carry the whole-job value, per-rank step times and the
    all-reduce accounting (bytes, buckets, exposed time, ring floor), so that the first 8-GPU run yields a complete record.
    dp8: BASELINE configs[3]'s shape of the job -- 8 ranks x 8 frames = global batch 64 (small frames here: eight processes on one device)."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, ELD_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    hw = '256' if world == 2 else '128'
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world), '--master-addr', '127.0.0.1',
                        '--master-port', str(port), os.path.join(root, 'bench.py'), '--gpus', str(world), '--steps', '2', '--warmup', '1', '--batch', str(batch),
                        '--height', hw, '--width', hw, '--no-cpu-baseline'], capture_output=True, text=True, timeout=1500, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, lines                                   # rank 0 prints ONE line
    line = json.loads(lines[0])
    assert line['n_gpus'] == world and line['steps'] == 2 and line['warmup'] == 1 and line['scaling'] == 'weak' and line['higher_is_better'] is True
    assert line['unit'] == 'raw MPix/s' and line['metric'].startswith('raw megapixels/sec')
    assert line['config']['global_batch'] == world * batch and line['config']['parallelism'] == 'dp%d' % world
    assert line['config']['images_per_gpu'] == batch
    px = world * batch * 4 * int(hw) * int(hw)                      # whole-job pixels per step: all ranks
    assert abs(line['value'] - px / (line['ms_per_step'] * 1e-3) / 1e6) <= 1e-3 * line['value']
    assert len(line['per_rank_ms_per_step']) == world and all(t > 0 for t in line['per_rank_ms_per_step'])
    ar = line['allreduce']
    assert ar['bytes'] == 4 * 7760484 and ar['buckets'] == 4       # 8 MiB buckets of the 31 MB gradient buffer
    for k in ('exposed_ms', 'ring_floor_ms', 'ms_per_step_without_exchange'):
        assert isinstance(ar[k], float) and ar[k] >= 0.0, k
    assert abs(ar['ring_floor_ms'] - 2.0 * (world - 1) / world * ar['bytes'] / 153e9 * 1e3) < 1e-3
    assert 'roofline' in line and 'roofline_sampler' in line
