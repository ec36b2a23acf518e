"""GPU parity of BASELINE.json configs[4] -- the multi-camera evaluation sweep (test_ELD.py:18-52 -> ELDModel.eval, models/ELD_model.py:203-307) --
at the sensor shapes and with the camera tables the sweep actually uses, not only SonyA7S2's:

    camera         packed frame (tools/eval_sweep.py CAMERAS)      G_shape entries
    SonyA7S2       1424 x 2128                                     18
    NikonD850      2752 x 4128                                     16
    CanonEOS70D    1824 x 2736                                     16
    CanonEOS700D   1728 x 2592                                     16

  * the engine's inference forward (fp32 default scheme and bf16) against the FLOAT64 oracle (oracle/unet_ref.py = models/arch/Unet.py:48-91) at
    1 x 4 x H x W of the three non-Sony sensors: tile counts, strip offsets and 32-bit buffer ranges (a 32-channel fp32 plane of a D850 frame is
    1.45 GB) that no Sony-sized test touches.  fp32: |out - f64| <= 1e-5 (1 + max|ref|) (north_star); bf16: PSNR >= 60 dB;
  * the batched sweep (tools/eval_sweep.py --batch): N frames per launch == N single-frame launches, bit for bit, at the largest sensor
    (two images inside one buffer descriptor: the virtual-row strip of conv_x3d / conv_bfd);
  * the sampler with every camera's tables: every Tukey-lambda shape of G_shape x the sweep's ISO / ratio settings, dump-replay bit-exact against
    the oracle (E-1(b)) and the variates themselves against the NumPy statement of the Philox layout.
"""
import importlib.util
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')

from oracle import noise_ref as O     # noqa: E402  (checker only)
from oracle import unet_ref as U      # noqa: E402  (checker only)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _sweep():
    spec = importlib.util.spec_from_file_location('eval_sweep', os.path.join(ROOT, 'tools', 'eval_sweep.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _psnr255(a, b):
    mse = torch.mean((a.double() * 255 - b.double() * 255) ** 2)
    return float(10 * torch.log10(255.0 ** 2 / mse))


@pytest.fixture(scope='module')
def net_and_sd(eld_lib):
    from eld_amd.unet import UNetSeeInDark
    torch.manual_seed(2018)
    net = UNetSeeInDark(4, 4)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    return net.cuda(), sd


@pytest.mark.parametrize('camera', ['NikonD850', 'CanonEOS70D', 'CanonEOS700D'])
def test_inference_at_sensor_resolution_vs_f64_oracle(eld_lib, net_and_sd, camera):
    net, sd = net_and_sd
    H, W = _sweep().CAMERAS[camera]
    g = torch.Generator(device='cuda').manual_seed(H)
    x = (torch.floor(65535.0 * torch.rand(1, 4, H, W, device='cuda', generator=g) ** 2.2) / 65535.0).contiguous()     # dark-heavy clean raw grid
    with torch.no_grad():
        net.inference_precision = 'fp32'
        o32 = net(x).clone()
        net.inference_precision = 'bf16'
        o16 = net(x).clone()
        net.inference_precision = 'fp32'
        # the float64 oracle on the GPU's fp64 vector units through stock torch ops (checker only, as tests/test_parity_full_gpu.py: the product
        # never calls an ATen convolution); a D850 frame is 4.2 TFLOP forward -- minutes on the host, seconds here
        ref = U.unet_forward({k: v.cuda().double() for k, v in sd.items()}, x.double())
    torch.cuda.synchronize()
    rmax = float(ref.abs().max())
    err = float((o32.double() - ref).abs().max())
    assert err <= 1e-5 * (1.0 + rmax), (camera, err, rmax)
    p16 = _psnr255(o16, ref)
    assert p16 >= 60.0, (camera, p16)
    assert float((o16.double() - ref).abs().max()) < 2e-2
    del ref, o32, o16
    net.release_workspaces()
    torch.cuda.empty_cache()


@pytest.mark.parametrize('precision', ['fp32', 'bf16'])
def test_batched_sweep_equals_single_frames_at_the_largest_sensor(eld_lib, net_and_sd, precision):
    """tools/eval_sweep.py --batch N: the frames of a setting go through the chain N per launch.  Image independence at 2 x 4 x 2752 x 4128 (two D850
    frames behind one buffer descriptor, 2.9 GB per 32-channel fp32 tensor): bit-identical to the single-frame launches, for the U-Net, the
    illuminance correction and the quality kernel."""
    from eld_amd.metrics import illuminance_correct, quality_assess_frames
    net, _ = net_and_sd
    H, W = _sweep().CAMERAS['NikonD850']
    g = torch.Generator(device='cuda').manual_seed(5)
    x = (torch.floor(65535.0 * torch.rand(2, 4, H, W, device='cuda', generator=g) ** 2.2) / 65535.0).contiguous()
    net.inference_precision = precision
    try:
        with torch.no_grad():
            ob = net(x).clone()
            cb = illuminance_correct(ob, x)
            qb = quality_assess_frames(cb, x)
            for i in range(2):
                xi = x[i:i + 1].contiguous()
                oi = net(xi)
                assert torch.equal(oi[0], ob[i]), (precision, i)
                ci = illuminance_correct(oi, xi)
                assert torch.equal(ci[0], cb[i])
                qi = quality_assess_frames(ci, xi)
                assert torch.equal(qi[0], qb[i]), (qi, qb)
    finally:
        net.inference_precision = 'fp32'
        net.release_workspaces()
    torch.cuda.empty_cache()


@pytest.mark.parametrize('camera', ['SonyA7S2', 'NikonD850', 'CanonEOS70D', 'CanonEOS700D'])
def test_sampler_with_every_cameras_tables(eld_lib, camera):
    """The sweep's sampler settings per camera (tools/eval_sweep.py params_for: K tied to the ISO inside the camera's [Kmin, Kmax], scales from the
    camera's regressions) with EVERY Tukey-lambda shape of the camera's G_shape table (18 entries for SonyA7S2, 16 for the others):
    one image per shape in one launch.  E-1(b): the kernel's dumped Philox variates replayed through the oracle give the kernel's output bit for bit;
    and the Tukey-lambda variates themselves agree with the NumPy statement of the same Philox layout for every shape."""
    from eld_amd import _lib as L
    from eld_amd.noise import NoiseParams, load_camera_params, model_flags, sample_noise
    es = _sweep()
    tables = load_camera_params(camera)
    shapes = [float(v) for v in tables['G_shape']]
    assert len(shapes) == (18 if camera == 'SonyA7S2' else 16)
    rng = np.random.RandomState(len(camera))
    settings = [(iso, ratio) for iso in es.ISOS for ratio in es.RATIOS]
    plist = []
    for i, lam in enumerate(shapes):
        iso, ratio = settings[i % len(settings)]
        p = es.params_for(tables, iso, ratio, rng)
        plist.append(NoiseParams(p[0], p[1], p[2], p[3], tl_lambda=lam, tl_scale=p.tl_scale, row_scale=p.row_scale, q_step=p.q_step))
    N, Hh, Ww = len(plist), 24, 40
    u16 = np.floor(65535.0 * np.random.default_rng(3).uniform(0, 1, size=(N, 4, Hh, Ww)) ** 2.2).astype(np.uint16)
    y = (u16 / 65535).astype(np.float32)
    flags = model_flags('PGRU') | L.CLIP
    ids = [500 + 7 * i for i in range(N)]
    numel = y.size
    dmp = torch.full((L.NPLANES, numel), float('nan'), dtype=torch.float32, device='cuda')
    yt = torch.from_numpy(y).cuda()
    z = sample_noise(yt, plist, flags, 2018, ids, dump=dmp)
    z_plain = sample_noise(yt, plist, flags, 2018, ids)
    torch.cuda.synchronize()
    assert torch.equal(z, z_plain)                                   # debug and production instantiations agree
    z = z.cpu().numpy()
    d = dmp.cpu().numpy()
    v = {k: d[i].reshape(y.shape) for k, i in L.PLANE.items()}
    oflags = O.model_flags('PGRU') | O.CLIP
    for i, p in enumerate(plist):
        op = O.Params(K=p[0], g_scale=p[1], saturation=p[2], ratio=p[3], tl_lambda=p.tl_lambda, tl_scale=p.tl_scale, row_scale=p.row_scale,
                      q_step=p.q_step, color_bias=p.color_bias)
        zi = O.noise_arith(y[i], op, oflags, **{k: a[i] for k, a in v.items()})
        assert np.array_equal(z[i], zi), (camera, i, p.tl_lambda)
        o = O.philox_variates(y.shape[1:], op, oflags, 2018, ids[i], y=y[i])
        assert np.array_equal(v['u_q'][i], o['u_q'])
        tl_err = np.abs(v['t_tl'][i] - o['t_tl']) / (1.0 + np.abs(o['t_tl']))
        assert np.max(tl_err) < 5e-5, (camera, p.tl_lambda, float(np.max(tl_err)))
        assert np.max(np.abs(v['n_row'][i] - o['n_row'])) < 1e-4
        assert np.mean(v['counts'][i] != o['counts']) < 4e-3
