"""GPU parity tests of the fused noise sampler, through the C ABI (eld_noise_forward).

Parity notions:
  E-1 deterministic arithmetic, bit-exact: the reference's own NumPy draws (golden fixtures) are injected
      into the kernel; and the kernel's own Philox variates (dumped) are re-played through the oracle.
  E-2 distributional: Philox-driven variates vs theory / oracle at BASELINE.json's full size.
  E-3 self-determinism: same (seed, sample_id) => identical bits across batching, vector/scalar path.
"""
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')

from oracle import noise_ref as O          # noqa: E402  (checker only)
from oracle import philox_ref as px        # noqa: E402


@pytest.fixture(scope='module')
def dev(eld_lib):
    assert torch.cuda.is_available(), 'GPU tests need a GPU (run with -m "not gpu" elsewhere)'
    return torch.device('cuda:0')


def run(y, plist, flags, seed=2018, ids=None, inject=None, dump=False, in_u16=False):
    from eld_amd.noise import sample_noise
    from eld_amd import _lib as L
    if in_u16:
        yt = torch.from_numpy(np.ascontiguousarray(y).view(np.int16)).cuda()
    else:
        yt = torch.from_numpy(np.ascontiguousarray(y)).cuda()
    N = yt.shape[0]
    numel = int(np.prod(y.shape))
    inj = dmp = None
    if inject is not None:
        planes = np.zeros((L.NPLANES, numel), np.float32)
        for k, v in inject.items():
            planes[L.PLANE[k]] = np.asarray(v, np.float32).reshape(-1)
        inj = torch.from_numpy(planes).cuda()
    if dump:
        dmp = torch.full((L.NPLANES, max(numel, 1)), float('nan'), dtype=torch.float32, device='cuda')
    z = sample_noise(yt, plist, flags, seed, ids if ids is not None else list(range(N)), in_u16=in_u16, inject=inj, dump=dmp)
    torch.cuda.synchronize()
    z = z.cpu().numpy()
    if dump:
        d = dmp.cpu().numpy()[:, :numel]
        return z, {k: d[i].reshape(y.shape) for k, i in L.PLANE.items()}
    return z


def P(K=2.288, g=6.451, sat=15583, ratio=208.98, **kw):
    from eld_amd.noise import NoiseParams
    return NoiseParams(K, g, sat, ratio, **kw)


def OP(p):
    return O.Params(K=p[0], g_scale=p[1], saturation=p[2], ratio=p[3], tl_lambda=p.tl_lambda, tl_scale=p.tl_scale,
                    row_scale=p.row_scale, q_step=p.q_step, color_bias=p.color_bias)


def synth(rng, shape):
    u16 = np.floor(65535.0 * rng.uniform(0, 1, size=shape) ** 2.2).astype(np.uint16)
    return (u16 / 65535).astype(np.float32)


# ------------------------------------------------------------------------------------------ RNG words
def test_philox_words_bit_exact(dev, eld_lib):
    from eld_amd import _lib as L
    n = 4099
    out = torch.empty((n, 4), dtype=torch.int32, device=dev)
    for (index0, sid, stream, it, seed) in [(0, 0, 0, 0, 0), (12345, (7 << 32) | 99, 5, 0, 2018),
                                           (2 ** 32 - 2000, 3, 7, 9, (0xDEADBEEF << 32) | 0x1234567)]:
        L.check(eld_lib.eld_philox_words(L.dptr(out), n, index0 & 0xFFFFFFFF, sid, stream, it, seed, L.cur_stream()))
        got = out.cpu().numpy().view(np.uint32)
        idx = (np.arange(n, dtype=np.uint64) + index0).astype(np.uint32)
        exp = np.stack(px.sampler_words(idx, sid, stream, seed, it=np.uint32(it)), axis=1)
        assert np.array_equal(got, exp)


# ------------------------------------------------------------------------------------------ E-1 vs the reference
def test_injected_reference_draws_bit_exact(dev, golden_dir):
    """Kernel arithmetic on the reference's own draws == the reference's output, every bit (45 cases:
    models g/Pg/pg/P/p x 3 parameter regimes x {12x20, ragged 5x7, empty})."""
    files = sorted(f for f in glob.glob(os.path.join(golden_dir, 'noise_*_*_*.npz')) if 'default' not in f)
    assert len(files) == 45
    from eld_amd.noise import model_flags
    for f in files:
        d = np.load(f)
        y = d['y'][None]
        inj = {k: d[k] for k in ('counts', 'n_shot', 'n_read') if k in d.files}
        z = run(y, [tuple(d['params'])], model_flags(str(d['model'])), inject=inj)
        assert z.shape == y.shape and np.array_equal(z[0], d['z']), f


def test_injected_batch_and_clip(dev, golden_dir):
    """Same through one batched launch (different params per image) with the caller's clip fused."""
    from eld_amd import _lib as L
    names = ['noise_Pg_sony_mid_s', 'noise_Pg_bright_s', 'noise_Pg_dark_s']
    ds = [np.load(os.path.join(golden_dir, n + '.npz')) for n in names]
    y = np.stack([d['y'] for d in ds])
    inj = {k: np.stack([d[k] for d in ds]) for k in ('counts', 'n_read')}
    z = run(y, [tuple(d['params']) for d in ds], L.SHOT_POISSON | L.READ_GAUSS | L.CLIP, inject=inj)
    for i, d in enumerate(ds):
        assert np.array_equal(z[i], np.maximum(np.minimum(d['z'], 1.0), 0))     # sid_dataset.py:277


# ------------------------------------------------------------------------------------------ E-1 on the Philox path
FULL = O.SHOT_POISSON | O.READ_TL | O.ROW | O.QUANT


@pytest.mark.parametrize('flags', [FULL, FULL | O.CBIAS | O.CLIP, O.SHOT_GAUSS | O.READ_GAUSS, O.SHOT_POISSON | O.READ_GAUSS,
                                   O.READ_GAUSS | O.READ_TL | O.ROW | O.QUANT | O.CBIAS, 0])
@pytest.mark.parametrize('shape', [(2, 4, 16, 24), (1, 4, 5, 7), (3, 4, 33, 12)])
def test_dumped_variates_replay_bit_exact(dev, flags, shape):
    """The production kernel (Philox) reports the variates it used; the oracle fed with them gives the same bits."""
    rng = np.random.default_rng(42)
    y = synth(rng, shape)
    plist = [P(K=0.4 + i, g=3.0 + i, ratio=100.0 + 90 * i, tl_lambda=[-0.14285714, 0.0, 0.114285715][i % 3], tl_scale=2.5 + i,
               row_scale=0.7 + i, color_bias=(1.5, -1.0, 0.25, 4.0)) for i in range(shape[0])]
    z, v = run(y, plist, flags, ids=[10 + i for i in range(shape[0])], dump=True)
    z_plain = run(y, plist, flags, ids=[10 + i for i in range(shape[0])])
    assert np.array_equal(z, z_plain)                  # debug and production instantiations agree
    for i in range(shape[0]):
        zi = O.noise_arith(y[i], OP(plist[i]), flags, **{k: a[i] for k, a in v.items()})
        assert np.array_equal(z[i], zi)


def test_philox_variates_match_oracle(dev):
    """Variates themselves vs the NumPy statement of the same Philox layout: uniforms bit-exact,
    transcendental-derived ones to hardware-intrinsic tolerance, Poisson counts equal except at
    accept/reject (PTRS) and CDF (fractional-rate inversion) boundaries decided by an ulp of exp / log."""
    shape = (1, 4, 40, 52)
    rng = np.random.default_rng(1)
    y = synth(rng, shape)
    flags = FULL | O.READ_GAUSS
    for p, sid in [(P(K=2.288, ratio=208.98, tl_lambda=-0.14285714, tl_scale=3.0, row_scale=0.5), 77),
                   (P(K=0.1, ratio=100.0, tl_lambda=0.0, tl_scale=1.0, row_scale=1.0), (5 << 32) | 1),
                   (P(K=0.5, ratio=150.0, tl_lambda=0.114285715, tl_scale=1.0, row_scale=1.0), 3)]:
        _, v = run(y, [p], flags, seed=99, ids=[sid], dump=True)
        o = O.philox_variates(shape[1:], OP(p), flags, 99, sid, y=y[0])
        assert np.array_equal(v['u_q'][0], o['u_q'])
        # gfx950 v_sin/v_cos/v_log vs NumPy libm: ~4e-5 absolute on N(0,1) draws, i.e. < 1e-5 of full scale
        # once multiplied by sigma*ratio/saturation (checked on z in test_gaussian_model_z_parity)
        assert np.max(np.abs(v['n_read'][0] - o['n_read'])) < 1e-4
        assert np.max(np.abs(v['n_row'][0] - o['n_row'])) < 1e-4
        tl_err = np.abs(v['t_tl'][0] - o['t_tl']) / (1.0 + np.abs(o['t_tl']))
        assert np.max(tl_err) < 5e-5
        mism = np.mean(v['counts'][0] != o['counts'])
        assert mism < 2e-3, mism
        assert np.max(np.abs(v['counts'][0] - o['counts'])) <= np.maximum(3, 0.2 * np.sqrt(o['counts'].max()))


def test_gaussian_model_z_parity(dev):
    """Whole-sampler parity for the reference's default model 'g' under Philox: |z_hip - z_oracle| <= 1e-5
    (BASELINE.json north_star tolerance for the floating-point part), SonyA7S2 mid-range parameters."""
    shape = (2, 4, 64, 96)
    y = synth(np.random.default_rng(5), shape)
    p = P()
    for flags in (O.READ_GAUSS, O.SHOT_GAUSS | O.READ_GAUSS, O.READ_GAUSS | O.READ_TL | O.ROW | O.QUANT):
        pp = P(tl_lambda=-0.14285714, tl_scale=3.0, row_scale=0.5)
        z = run(y, [pp, pp], flags, seed=4, ids=[8, 9])
        for i in range(2):
            zo, _ = O.noise_philox(y[i], OP(pp), flags, 4, 8 + i)
            assert np.max(np.abs(z[i] - zo)) <= 1e-5, (flags, np.max(np.abs(z[i] - zo)))


def test_division_is_numpy_division(dev):
    """The kernel forms y1 / ratio and z / saturation with a reciprocal + two FMA refinements (div_rn); the results must be NumPy's
    correctly rounded float32 quotients bit for bit: scale-only and Gaussian models (every variate injected) over random inputs,
    ratios and saturation levels, including awkward divisors (mantissa all ones, powers of two)."""
    rng = np.random.default_rng(21)
    shape = (1, 4, 128, 256)
    n = int(np.prod(shape))
    divisors = [(15583.0, 208.98), (16383.0, 100.0), (4095.0, 300.0), (float(np.float32(1.9999999)), 127.99999), (1024.0, 256.0), (65535.0, 1.0000001),
                (float(np.nextafter(np.float32(16384), np.float32(0))), float(np.nextafter(np.float32(128), np.float32(0))))]
    divisors += [(float(s_), float(r_)) for s_, r_ in zip(rng.uniform(1000, 70000, 12), rng.uniform(50, 400, 12))]
    for sat, ratio in divisors:
        y = rng.uniform(0, 1, size=shape).astype(np.float32) ** 3
        y.reshape(-1)[:8] = [0.0, 1.0, 1e-30, 1e-38, 1e-42, 0.5, 3e-7, 0.999999]
        p = P(K=1.7, g=3.3, sat=sat, ratio=ratio)
        for flags in (0, O.SHOT_GAUSS | O.READ_GAUSS):
            inj = {'n_shot': rng.standard_normal(n).astype(np.float32), 'n_read': rng.standard_normal(n).astype(np.float32)}
            z = run(y, [p], flags, inject=inj)
            zo = O.noise_arith(y[0], OP(p), flags, n_shot=inj['n_shot'].reshape(shape[1:]), n_read=inj['n_read'].reshape(shape[1:]))
            assert np.array_equal(z[0].view(np.uint32), zo.view(np.uint32)), (sat, ratio, flags, int(np.sum(z[0] != zo)))
        z = run(y, [p], 0)                                                   # production (non-debug) instantiation, scale only
        assert np.array_equal(z[0].view(np.uint32), O.noise_arith(y[0], OP(p), 0).view(np.uint32)), (sat, ratio)


# ------------------------------------------------------------------------------------------ E-3
def test_determinism_and_shard_invariance(dev):
    rng = np.random.default_rng(3)
    y = synth(rng, (5, 4, 24, 36))
    plist = [P(K=1.0 + i, ratio=120.0 + 30 * i, tl_lambda=-0.1, tl_scale=2.0, row_scale=0.6) for i in range(5)]
    ids = [1000, 1001, 1002, 1003, 1004]
    z_all = run(y, plist, FULL, ids=ids)
    assert np.array_equal(z_all, run(y, plist, FULL, ids=ids))                       # run-to-run
    for i in range(5):                                                               # any sharding of the batch
        assert np.array_equal(z_all[i], run(y[i:i + 1], [plist[i]], FULL, ids=[ids[i]])[0])
    assert not np.array_equal(z_all[0], run(y[:1], [plist[0]], FULL, ids=[2000])[0])   # other sample id -> other noise
    assert not np.array_equal(z_all[0], run(y[:1], [plist[0]], FULL, seed=1, ids=[1000])[0])


def test_vector_and_scalar_paths_agree(dev, eld_lib):
    """Force the scalar kernel on a W%4==0 image by misaligning the pointers by one float."""
    from eld_amd import _lib as L
    from eld_amd.noise import NoiseParams
    N, C, H, W = 2, 4, 8, 16
    numel = N * C * H * W
    y = synth(np.random.default_rng(9), (N, C, H, W))
    plist = [P(tl_lambda=-0.1, tl_scale=2.0, row_scale=0.6), P(K=0.3, ratio=100.0, tl_lambda=0.1, tl_scale=1.0, row_scale=0.2)]
    recs = np.stack([NoiseParams.coerce(p).record(50 + i) for i, p in enumerate(plist)])
    prm = torch.from_numpy(recs.view(np.uint8).reshape(-1).copy()).cuda()
    buf_in = torch.zeros(numel + 4, dtype=torch.float32, device=dev)
    buf_out = torch.zeros(numel + 4, dtype=torch.float32, device=dev)
    outs = []
    for off in (0, 1):
        buf_in[off:off + numel] = torch.from_numpy(y.reshape(-1)).cuda()
        rc = eld_lib.eld_noise_forward(buf_in.data_ptr() + 4 * off, 0, buf_out.data_ptr() + 4 * off, L.dptr(prm),
                                       N, C, H, W, FULL | O.READ_GAUSS, 2018, None, None, L.cur_stream())
        L.check(rc)
        torch.cuda.synchronize()
        outs.append(buf_out[off:off + numel].cpu().numpy().copy())
    assert np.array_equal(outs[0], outs[1])


# ------------------------------------------------------------------------------------------ integer paths
def test_u16_decode_all_codes_bit_exact(dev, golden_dir):
    """uint16 LMDB codes decoded in-kernel (lmdb_dataset.py:38-39) == the reference decode, all 65,536 codes."""
    dec = np.load(os.path.join(golden_dir, 'lmdb_decode.npz'))['decoded']
    codes = np.arange(65536, dtype=np.uint16).reshape(1, 4, 128, 128)
    p = P(K=1.0, g=0.0, sat=1.0, ratio=1.0)
    z = run(codes, [p], 0, in_u16=True)                      # no noise terms, S=r=1: z == y exactly
    assert np.array_equal(z.reshape(-1), dec)
    p2 = P()
    z2 = run(codes, [p2], 0, in_u16=True)
    assert np.array_equal(z2[0], O.noise_arith(dec.reshape(4, 128, 128), OP(p2), 0))


def test_bayer_pack_unpack(dev, golden_dir):
    from eld_amd.noise import RawPacker
    d = np.load(os.path.join(golden_dir, 'rawpacker.npz'))
    rp = RawPacker('bayer')
    assert np.array_equal(rp.pack_raw(d['mosaic']), d['packed'])
    assert np.array_equal(rp.unpack_raw(d['packed']), d['unpacked'])
    m = torch.rand(3, 2 * 178, 2 * 266, device=dev)
    pk = rp.pack_raw(m)
    assert pk.shape == (3, 4, 178, 266)
    assert torch.equal(rp.unpack_raw(pk), m)                   # round trip at batch
    assert torch.equal(pk[:, 2], m[:, 1::2, 1::2]) and torch.equal(pk[:, 3], m[:, 1::2, 0::2])
    with pytest.raises(NotImplementedError):
        RawPacker('foveon').pack_raw(d['mosaic'])               # unknown CFA: noise.py:135


def test_xtrans_pack_unpack(dev, golden_dir):
    """RawPacker('xtrans') (noise.py:22-64, 83-127) through eld_pack_xtrans / eld_unpack_xtrans: bit-exact against outputs minted
    from the reference (ragged sides are truncated to whole 6x6 cells; odd packed sides), and a batched round trip at sensor size."""
    from eld_amd.noise import RawPacker
    d = np.load(os.path.join(golden_dir, 'rawpacker_xtrans.npz'))
    rp = RawPacker('xtrans')
    for name in ('ragged', 'exact'):
        pk = rp.pack_raw(d[name + '_mosaic'])
        assert pk.dtype == np.float32 and np.array_equal(pk, d[name + '_packed'])
        assert np.array_equal(rp.unpack_raw(pk), d[name + '_unpacked'])
    assert np.array_equal(rp.unpack_raw(d['odd_packed']), d['odd_unpacked'])
    m = torch.rand(2, 4032, 6030, device=dev)                   # X-T2 sized mosaic (multiples of 6), batch of two
    pk = rp.pack_raw(m)
    assert pk.shape == (2, 9, 1344, 2010)
    assert torch.equal(rp.unpack_raw(pk), m)                    # the 36 cell positions are covered exactly once
    assert torch.equal(pk[:, 5], m[:, 1::3, 0::3]) and torch.equal(pk[:, 8], m[:, 2::3, 1::3])      # noise.py:60-63
    assert torch.equal(pk[:, 0, 1::2, 0::2], m[:, 3::6, 1::6]) and torch.equal(pk[:, 3, 0::2, 1::2], m[:, 2::6, 5::6])
    assert rp.pack_raw(torch.zeros(5, 5, device=dev)).shape == (9, 0, 0)      # no whole cell: empty, like the reference


def test_edge_cases(dev, eld_lib):
    assert run(np.zeros((0, 4, 8, 8), np.float32), [], FULL).shape == (0, 4, 8, 8)
    assert run(np.zeros((1, 4, 0, 8), np.float32), [P()], FULL).shape == (1, 4, 0, 8)
    z = run(np.zeros((1, 4, 3, 5), np.float32), [P(g=0.0)], O.SHOT_POISSON)     # lambda = 0 -> exactly 0
    assert np.all(z == 0)
    z = run(np.ones((1, 1, 1, 1), np.float32), [P()], O.SHOT_POISSON | O.READ_GAUSS)   # single pixel, C=1
    assert np.isfinite(z).all()


# ------------------------------------------------------------------------------------------ E-2 at full size
def test_full_size_distribution(dev):
    """One 4x1424x2128 image (BASELINE.json config 2), full model: moment identities, row structure,
    Poisson chi-square per rate, Tukey-lambda / uniform KS -- size-independent properties."""
    from scipy import stats
    shape = (1, 4, 1424, 2128)
    levels = np.array([0.0, 0.004, 0.03, 0.1, 0.35, 1.0], np.float32)
    y = np.empty(shape, np.float32)
    for i, lv in enumerate(levels):                       # vertical stripes of constant signal
        y[..., i::len(levels)] = lv
    p = P(K=2.288, g=6.451, ratio=208.98, tl_lambda=-0.14285714, tl_scale=3.3, row_scale=0.9)
    z, v = run(y, [p], FULL, ids=[123456789], dump=True)
    op = OP(p)
    assert np.array_equal(z[0], O.noise_arith(y[0], op, FULL, **{k: a[0] for k, a in v.items()}))
    lam = O.poisson_lambda(y[0], op)
    cnt = v['counts'][0]
    for i, lv in enumerate(levels):
        k = cnt[..., i::len(levels)].reshape(-1).astype(np.int64)
        l = float(lam[0, 0, i])
        n = k.size
        assert abs(k.mean() - l) < 6 * np.sqrt(max(l, 1e-12) / n) + 1e-12, (lv, k.mean(), l)
        assert abs(k.var() - l) < 7 * (l * np.sqrt(2.0 / n) + np.sqrt(max(l, 1e-12) / n)) + 1e-12
        if l > 0:
            lo, hi = int(stats.poisson.ppf(1e-5, l)), int(stats.poisson.ppf(1 - 1e-5, l))
            obs = np.bincount(np.clip(k, lo, hi) - lo, minlength=hi - lo + 1).astype(np.float64)
            pmf = stats.poisson.pmf(np.arange(lo, hi + 1), l)
            pmf[0] += stats.poisson.cdf(lo - 1, l)
            pmf[-1] += stats.poisson.sf(hi, l)
            exp = pmf * n
            keep = exp > 10
            chi2 = ((obs[keep] - exp[keep]) ** 2 / exp[keep]).sum()
            assert chi2 < stats.chi2.ppf(1 - 1e-7, keep.sum()), (lv, chi2, keep.sum())
    # E[z] = y and Var identity in ADU of the short exposure: K*lam*K + tl var + row var + q^2/12
    zz = (z[0].astype(np.float64) * op['saturation'] / op['ratio'])
    for i, lv in enumerate(levels):
        s = zz[..., i::len(levels)].reshape(-1)
        mean_exp = float(lv) * op['saturation'] / op['ratio']
        assert abs(s.mean() - mean_exp) < 0.05, (lv, s.mean(), mean_exp)
    # row noise: constant along rows, shared by channel pairs, ~N(0,1) across the 2848 sensor rows
    nr = v['n_row'][0]
    assert np.all(nr == nr[:, :, :1]) and np.array_equal(nr[0], nr[1]) and np.array_equal(nr[2], nr[3])
    rows = np.concatenate([nr[0, :, 0], nr[2, :, 0]]).astype(np.float64)
    assert stats.kstest(rows, 'norm').pvalue > 1e-4
    # Tukey-lambda and quantisation uniforms (subsample for KS)
    sub = slice(None, None, 7)
    t = v['t_tl'][0].reshape(-1)[sub].astype(np.float64)
    assert stats.kstest(t, stats.tukeylambda(-0.14285714).cdf).pvalue > 1e-4
    assert stats.kstest(v['u_q'][0].reshape(-1)[sub].astype(np.float64), 'uniform').pvalue > 1e-4


def _chi2_poisson(k, lam):
    from scipy import stats
    n = k.size
    lo, hi = int(stats.poisson.ppf(1e-5, lam)), int(stats.poisson.ppf(1 - 1e-5, lam))
    obs = np.bincount(np.clip(k, lo, hi) - lo, minlength=hi - lo + 1).astype(np.float64)
    pmf = stats.poisson.pmf(np.arange(lo, hi + 1), lam)
    pmf[0] += stats.poisson.cdf(lo - 1, lam)
    pmf[-1] += stats.poisson.sf(hi, lam)
    keep = pmf * n > 10
    chi2 = ((obs[keep] - pmf[keep] * n) ** 2 / (pmf[keep] * n)).sum()
    return chi2, stats.chi2.ppf(1 - 1e-7, keep.sum())


def test_poisson_regimes(dev):
    """Every route of the Poisson draw has the Poisson law and the oracle's counts: alias table + fractional inversion just below
    32, PTRS inline where a whole wave is above 32, PTRS through the queue where few lanes are (checkerboard of a small and a
    large rate, one pixel in eight large), and large rates (K = 0.1: lam up to 1558)."""
    shape = (1, 4, 256, 256)
    flags = O.SHOT_POISSON | O.READ_GAUSS
    sat = 15583.0
    for lam_big, lam_small, every in [(31.9, 31.9, 1), (33.0, 33.0, 1), (200.0, 200.0, 1), (1500.0, 1500.0, 1), (40.0, 2.0, 8), (50.0, 0.5, 2)]:
        # y * (S / ratio / K) = lam: choose ratio = 100, K so that y = 1 gives lam_big
        ratio = 100.0
        K = float(np.float32(sat / ratio / lam_big))
        p = P(K=K, g=1.0, sat=sat, ratio=ratio)
        y = np.full(shape, np.float32(lam_small / lam_big), np.float32)
        big = np.zeros(shape, bool)
        big.reshape(-1)[::every] = True
        y[big] = 1.0
        _, v = run(y, [p], flags, seed=7, ids=[31], dump=True)
        lam = O.poisson_lambda_fast(y[0], OP(p))
        cnt = v['counts'][0].astype(np.int64)
        for sel in ((big[0],) if every == 1 else (big[0], ~big[0])):
            l = float(lam[sel][0])
            chi2, lim = _chi2_poisson(cnt[sel], l)
            assert chi2 < lim, (lam_big, lam_small, every, l, chi2, lim)
        o = O.philox_variates(shape[1:], OP(p), flags, 7, 31, y=y[0])
        mism = np.mean(cnt != o['counts'])
        assert mism < 2e-3, (lam_big, lam_small, every, mism)


def test_gaussian_terms_distribution(dev):
    from scipy import stats
    shape = (1, 4, 512, 512)
    y = np.full(shape, 0.25, np.float32)
    _, v = run(y, [P()], O.SHOT_GAUSS | O.READ_GAUSS, ids=[5], dump=True)
    for k in ('n_shot', 'n_read'):
        n = v[k].reshape(-1).astype(np.float64)
        assert abs(n.mean()) < 5 / np.sqrt(n.size) and abs(n.var() - 1) < 6 * np.sqrt(2.0 / n.size)
        assert stats.kstest(n[::5], 'norm').pvalue > 1e-4
    assert abs(np.corrcoef(v['n_shot'].reshape(-1), v['n_read'].reshape(-1))[0, 1]) < 5e-3


def test_plugin_call_surface(dev):
    """noise.NoiseModel drop-in: ndarray in -> ndarray out, same shape/dtype, not clipped (noise.py:149-170)."""
    from eld_amd.noise import NoiseModel
    nm = NoiseModel(model='Pg', include=4)
    y = synth(np.random.default_rng(0), (4, 64, 64))
    np.random.seed(0)
    z = nm(y)
    assert isinstance(z, np.ndarray) and z.shape == y.shape and z.dtype == np.float32
    assert (z < 0).any()                                   # read noise around dark pixels goes negative: no clip
    params = nm._sample_params()
    burst = [nm(y, params=params) for _ in range(3)]       # burst call site, sid_dataset.py:267-273
    assert not np.array_equal(burst[0], burst[1])
    zt = nm(torch.from_numpy(y).cuda()[None].repeat(2, 1, 1, 1))
    assert zt.is_cuda and zt.shape == (2, 4, 64, 64)


def test_augment_matches_reference_golden(dev, golden_dir):
    """Device augmentation == ELDTrainDataset.__getitem__ output (golden minted from dataset/sid_dataset.py:332-363)."""
    from eld_amd.noise import augment
    d = np.load(os.path.join(golden_dir, 'augment.npz'))
    bits = [int(b[0]) | (int(b[1]) << 1) | (int(b[2]) << 2) for b in d['bits']]
    assert len(set(bits)) > 3                                   # several flip/transpose combinations are exercised
    oi = augment(torch.from_numpy(d['inp']).cuda(), bits, clip=True).cpu().numpy()
    ot = augment(torch.from_numpy(d['tgt']).cuda(), bits, clip=False).cpu().numpy()
    assert np.array_equal(oi, d['out_inp']) and np.array_equal(ot, d['out_tgt'])
    x = torch.rand(3, 4, 512, 512, device=dev)
    for b in range(8):                                          # involutions at the training patch size
        y = augment(x, [b] * 3)
        back = augment(y, [b] * 3) if not (b & 4) or b in (4, 7) else None
        if back is not None:
            assert torch.equal(back, x), b
        assert torch.equal(y.sum(dtype=torch.float64), x.sum(dtype=torch.float64))
    assert torch.equal(augment(x, [4, 4, 4]), x.transpose(2, 3).contiguous())
    assert torch.equal(augment(x, [1, 2, 3]), torch.stack([x[0].flip(1), x[1].flip(2), x[2].flip(1).flip(2)]))


def test_pack_raw_black_level_kernel(dev, golden_dir):
    """n3: uint16 mosaic -> per-channel black level -> /(16383 - black) -> clip -> pack, one kernel (sid_dataset.py:172-196):
    bit-exact against the reference-minted golden (four CFA patterns, odd packed widths) and against the oracle on a full
    2848 x 4256 SonyA7S2 mosaic, batched."""
    from eld_amd.noise import pack_raw_bayer
    d = np.load(os.path.join(golden_dir, 'pack_raw.npz'))
    for n in ('rggb', 'grbg', 'bggr', 'gbrg'):
        got = pack_raw_bayer(d[n + '_im'], d[n + '_pattern'], d[n + '_black'])
        assert got.dtype == np.float32 and np.array_equal(got, d[n + '_out']), n
    rng = np.random.default_rng(2)
    im = rng.integers(0, 16384, size=(2, 2848, 4256)).astype(np.uint16)
    pat, black = [[0, 1], [3, 2]], [512.0, 512.0, 512.0, 512.0]
    t = torch.from_numpy(im.view(np.int16)).to(dev)
    got = pack_raw_bayer(t, pat, black)
    assert tuple(got.shape) == (2, 4, 1424, 2128)
    for i in range(2):
        assert np.array_equal(got[i].cpu().numpy(), O.pack_raw_sid(im[i], pat, black))
    with pytest.raises(Exception):
        pack_raw_bayer(im[0], [[0, 1], [1, 2]], black)              # not a permutation of the four colour codes
