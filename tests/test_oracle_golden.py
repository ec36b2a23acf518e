"""CPU tests: the oracle (oracle/*.py) against the golden vectors minted from the reference itself
(oracle/gen_golden.py), plus the published Random123 known-answer vectors for Philox."""
import glob
import os

import numpy as np
import pytest

from oracle import noise_ref as O
from oracle import philox_ref as px


def test_philox_random123_kat():
    # kat_vectors of Random123 (philox4x32, 10 rounds)
    kats = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
            ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
            ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
             (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, exp in kats:
        got = tuple(int(x) for x in px.philox4x32_10(*ctr, *key))
        assert got == exp
    # the same file's 7-round vectors: the round count the sampler runs (px.ROUNDS, eld_amd/csrc/philox.h)
    kats7 = [((0, 0, 0, 0), (0, 0), (0x5f6fb709, 0x0d893f64, 0x4f121f81, 0x4f730a48)),
             ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x5207ddc2, 0x45165e59, 0x4d8ee751, 0x8c52f662)),
             ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
              (0x4dfccaba, 0x190a87f0, 0xc47362ba, 0xb6b5242a))]
    assert px.ROUNDS == 7
    for ctr, key, exp in kats7:
        assert tuple(int(x) for x in px.philox4x32(*ctr, *key, rounds=7)) == exp
        assert tuple(int(x) for x in px.philox4x32(*ctr, *key)) == exp


def test_u01_exact_complement():
    w = np.random.default_rng(0).integers(0, 2 ** 32, size=10000, dtype=np.uint64).astype(np.uint32)
    u, v = px.u01(w), px.u01(~w)
    assert np.all((u > 0) & (u < 1))
    assert np.array_equal((u.astype(np.float64) + v.astype(np.float64)), np.ones(w.shape))


def _cases(golden_dir):
    return sorted(f for f in glob.glob(os.path.join(golden_dir, 'noise_*_*_*.npz')) if 'default' not in f)


def test_noise_arith_matches_reference_bit_exact(golden_dir):
    """E-1: oracle arithmetic with the reference's own draws re-injected == reference output, all bits."""
    files = _cases(golden_dir)
    assert len(files) == 45
    for f in files:
        d = np.load(f)
        model = str(d['model'])
        p = O.Params.from_tuple(d['params'])
        v = {k: d[k] for k in ('counts', 'n_shot', 'n_read') if k in d.files}
        z = O.noise_arith(d['y'], p, O.model_flags(model), **v)
        assert z.dtype == np.float32 and z.shape == d['z'].shape
        assert np.array_equal(z, d['z']), f


def test_noise_numpy_rng_replay(golden_dir):
    """The oracle consumes the global NumPy stream in the reference's order (Poisson block, then randn)."""
    for f in _cases(golden_dir):
        d = np.load(f)
        np.random.seed(int(d['np_seed']))
        z, _ = O.noise_numpy_rng(d['y'], str(d['model']), tuple(d['params']))
        assert np.array_equal(z, d['z']), f


def test_noise_default_params_path(golden_dir):
    """params=None: 5 host draws (noise.py:201-225) then the per-pixel blocks; the reference evaluates this
    case in float64 under NumPy 2 so the float32 oracle agrees to float32 round-off."""
    d = np.load(os.path.join(golden_dir, 'noise_default_params.npz'))
    from eld_amd.noise import load_camera_params
    cp = {'SonyA7S2': load_camera_params('SonyA7S2')}
    np.random.seed(int(d['np_seed']))
    params = O.sample_params(cp, ['SonyA7S2'])
    assert np.allclose(params, d['params'], rtol=0, atol=0)
    z, _ = O.noise_numpy_rng(d['y'], 'Pg', tuple(np.float32(x) for x in params))
    assert np.max(np.abs(z.astype(np.float64) - d['z'])) < 5e-7


def test_sample_params(golden_dir):
    from eld_amd.noise import load_camera_params, ALL_CAMERAS
    recs = np.load(os.path.join(golden_dir, 'sample_params.npz'))['recs']
    cp = {c: load_camera_params(c) for c in ALL_CAMERAS}
    i = 0
    for inc in (None, 4, 1):
        cams = ALL_CAMERAS if inc is None else [ALL_CAMERAS[inc]]
        for s in (0, 1, 2018):
            np.random.seed(s)
            for _ in range(3):
                got = O.sample_params(cp, cams)
                assert recs[i][0] == (-1 if inc is None else inc) and recs[i][1] == s
                assert np.array_equal(np.array(got, np.float64), recs[i][2:]), (inc, s)
                i += 1


def test_rawpacker(golden_dir):
    d = np.load(os.path.join(golden_dir, 'rawpacker.npz'))
    assert np.array_equal(O.pack_raw_bayer(d['mosaic']), d['packed'])
    assert np.array_equal(O.unpack_raw_bayer(d['packed']), d['unpacked'])
    assert np.array_equal(d['unpacked'], d['mosaic'])


def test_rawpacker_xtrans(golden_dir):
    """noise.py:22-64, 83-127: the table-driven restatement vs outputs minted from the reference's RawPacker('xtrans')."""
    d = np.load(os.path.join(golden_dir, 'rawpacker_xtrans.npz'))
    for name in ('ragged', 'exact'):
        pk = O.pack_raw_xtrans(d[name + '_mosaic'])
        assert pk.dtype == np.float32 and np.array_equal(pk, d[name + '_packed'])
        assert np.array_equal(O.unpack_raw_xtrans(pk), d[name + '_unpacked'])
    assert np.array_equal(O.unpack_raw_xtrans(d['odd_packed']), d['odd_unpacked'])
    rows, cols = O.xtrans_source_index(4, 6)                    # every cell position is hit exactly once
    assert len(set(zip(rows.ravel().tolist(), cols.ravel().tolist()))) == 9 * 4 * 6 == 12 * 18
    rows = O.sensor_row_index(4, 5)
    assert rows[0].tolist() == [0, 2, 4, 6, 8] and rows[1].tolist() == rows[0].tolist()
    assert rows[2].tolist() == [1, 3, 5, 7, 9] and rows[3].tolist() == rows[2].tolist()


def test_lmdb_decode_all_codes(golden_dir):
    dec = np.load(os.path.join(golden_dir, 'lmdb_decode.npz'))['decoded']
    codes = np.arange(65536, dtype=np.uint16)
    assert np.array_equal(O.lmdb_decode_u16(codes), dec)
    # the form the kernel uses: float32 true division
    assert np.array_equal((codes.astype(np.float32) / np.float32(65535.0)).astype(np.float32), dec)


def test_augment(golden_dir):
    d = np.load(os.path.join(golden_dir, 'augment.npz'))
    for i in range(d['inp'].shape[0]):
        b = d['bits'][i]
        ti = np.clip(O.augment(d['inp'][i], b[0], b[1], b[2]), 0, 1)       # sid_dataset.py:354
        tt = O.augment(d['tgt'][i], b[0], b[1], b[2])
        assert np.array_equal(ti, d['out_inp'][i]) and np.array_equal(tt, d['out_tgt'][i])


def test_tukey_lambda_vs_scipy():
    from scipy.stats import tukeylambda
    u = np.linspace(0.001, 0.999, 999).astype(np.float32)
    for lam in (0.2, 0.114285715, 0.0, -0.142857149, -0.243):
        q = O.tukey_lambda_quantile(u, lam).astype(np.float64)
        ref = tukeylambda.ppf(u.astype(np.float64), lam)
        assert np.max(np.abs(q - ref) / (1 + np.abs(ref))) < 2e-5, lam


@pytest.mark.parametrize('lam', [0.05, 1.0, 9.5, 10.5, 40.0, 1558.0])
def test_philox_poisson_distribution(lam):
    """E-2 on the oracle's own Philox Poisson (the statement the kernel follows): moments + chi-square."""
    from scipy import stats
    shape = (4, 64, 200)
    p = O.Params(K=1.0, saturation=1.0, ratio=1.0)
    y = np.full(shape, lam, np.float32)
    v = O.philox_variates(shape, p, O.SHOT_POISSON, seed=7, sample_id=3, y=y)
    k = v['counts'].reshape(-1).astype(np.int64)
    n = k.size
    assert abs(k.mean() - lam) < 5 * np.sqrt(lam / n) + 1e-9
    assert abs(k.var() - lam) < 6 * lam * np.sqrt(2.0 / n) + 5 * np.sqrt(lam / n) + 1e-9
    lo, hi = int(stats.poisson.ppf(1e-4, lam)), int(stats.poisson.ppf(1 - 1e-4, lam))
    edges = np.arange(lo, hi + 2)
    obs = np.histogram(np.clip(k, lo, hi), bins=edges)[0].astype(np.float64)
    pmf = stats.poisson.pmf(np.arange(lo, hi + 1), lam)
    pmf[0] += stats.poisson.cdf(lo - 1, lam)
    pmf[-1] += stats.poisson.sf(hi, lam)
    exp = pmf * n
    keep = exp > 5
    chi2 = ((obs[keep] - exp[keep]) ** 2 / exp[keep]).sum() + (obs[~keep].sum() - exp[~keep].sum()) ** 2 / max(exp[~keep].sum(), 1e-9) * (exp[~keep].sum() > 5)
    dof = keep.sum()
    assert chi2 < stats.chi2.ppf(1 - 1e-6, dof), (chi2, dof)


def test_philox_normals_and_rows():
    from scipy import stats
    shape = (4, 32, 64)
    p = O.Params(K=1.0, g_scale=1.0, row_scale=1.0, saturation=1.0, ratio=1.0)
    v = O.philox_variates(shape, p, O.READ_GAUSS | O.ROW | O.QUANT, seed=11, sample_id=5)
    n = v['n_read'].reshape(-1).astype(np.float64)
    assert stats.kstest(n, 'norm').pvalue > 1e-4
    assert stats.kstest(v['u_q'].reshape(-1).astype(np.float64), 'uniform').pvalue > 1e-4
    nr = v['n_row']
    assert np.all(nr == nr[:, :, :1])                        # constant along a row
    assert np.array_equal(nr[0], nr[1]) and np.array_equal(nr[2], nr[3])     # channel pairs share the sensor row
    assert not np.array_equal(nr[0], nr[2])


def test_pack_raw_sid_matches_reference_golden(golden_dir):
    """pack_raw_bayer (dataset/sid_dataset.py:172-196): oracle restatement vs the reference's output, four CFA patterns."""
    d = np.load(os.path.join(golden_dir, 'pack_raw.npz'))
    for n in ('rggb', 'grbg', 'bggr', 'gbrg'):
        assert np.array_equal(O.pack_raw_sid(d[n + '_im'], d[n + '_pattern'], d[n + '_black']), d[n + '_out'])


# ------------------------------------------------------------------------------------------ Poisson alias tables (sampler, lam < 32)
def test_poisson_alias_header_is_generated_and_exact():
    """eld_amd/csrc/poisson_alias_table.h is what oracle/gen_poisson_alias.py generates, and the distribution each integer table
    realises equals the Poisson(n) pmf (tail merged into outcome 63) to below 2^-26 per outcome."""
    import os
    from scipy import stats
    from oracle import gen_poisson_alias as G
    assert open(G.HEADER).read() == G.header_text()
    ent, q0 = G.tables()
    assert ent.shape == (32, 64) and ent.dtype == np.uint32
    for n in range(32):
        want = stats.poisson.pmf(np.arange(64), n) if n else np.eye(1, 64)[0]
        want = want.copy()
        want[63] = stats.poisson.sf(62, n) if n else 0.0
        got = G.realised_pmf(ent[n])
        assert np.max(np.abs(got - want)) < 2.0 ** -26, n
        if n:
            assert abs(float(q0[n]) - stats.poisson.pmf(63, n) / stats.poisson.sf(62, n)) < 1e-6


def test_oracle_table_poisson_is_poisson():
    """The oracle's statement of the sampler's small-rate draw (alias table + fractional-rate inversion) has the Poisson law:
    chi-square at rates either side of the integer grid, 400k draws each from NumPy-generated words."""
    from scipy import stats
    from oracle import noise_ref as O
    rng = np.random.default_rng(11)
    n = 400_000
    for lam in (0.0, 0.37, 1.0, 2.5, 9.99, 17.2, 31.999):
        wu = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
        wv = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
        k = O._pois_table(np.full(n, lam, np.float32), wu, wv, np.arange(n, dtype=np.uint32), 1, 2).astype(np.int64)
        if lam == 0.0:
            assert not k.any()
            continue
        lo, hi = int(stats.poisson.ppf(1e-5, lam)), int(stats.poisson.ppf(1 - 1e-5, lam))
        obs = np.bincount(np.clip(k, lo, hi) - lo, minlength=hi - lo + 1).astype(np.float64)
        pmf = stats.poisson.pmf(np.arange(lo, hi + 1), lam)
        pmf[0] += stats.poisson.cdf(lo - 1, lam)
        pmf[-1] += stats.poisson.sf(hi, lam)
        keep = pmf * n > 10
        chi2 = ((obs[keep] - pmf[keep] * n) ** 2 / (pmf[keep] * n)).sum()
        assert chi2 < stats.chi2.ppf(1 - 1e-6, keep.sum()), (lam, chi2)


def test_eval_oracle_vs_reference_fixture(golden_dir):
    """the eval-path checker against outputs of the reference itself (oracle/gen_golden_eval.py ran
    models/ELD_model.py's tensor2im :23-38, IlluminanceCorrect :138-169 and forward_chop :434-467)."""
    import torch
    from oracle import metrics_ref as M
    from oracle import unet_ref as U
    d = np.load(os.path.join(golden_dir, 'eval.npz'))
    for tag in ('ic_n', 'ic_one', 'ic_b1'):                    # batch with own sources / one shared source / batch 1
        got = M.illuminance_correct(d[tag + '_pred'], d[tag + '_src'])
        ref = d[tag + '_out']
        assert got.dtype == np.float32 and got.shape == ref.shape
        assert np.abs(got - ref).max() <= 2e-7 * float(np.abs(ref).max()), tag       # float32 torch.dot vs float64 dot: <= 2 ulp of alpha
    assert (d['ic_b1_src'] == 1).any() and (d['ic_b1_pred'] < 0).any() and (d['ic_b1_pred'] > 1).any()
    t = M.tensor2im(d['t2i_in'][0])
    assert np.array_equal(np.transpose(t, (1, 2, 0)), d['t2i_out'])                  # image 0 only, HWC, exact
    assert d['t2i_out'].min() == 0.0 and d['t2i_out'].max() == 255.0                 # both clip sides exercised
    sd = U.seeded_state_dict(4, 4, seed=2018)
    wsum = np.array([float(v.double().sum()) for v in sd.values()])
    if str(d['torch_version']) != torch.__version__ or not np.array_equal(wsum, d['wsum']):
        pytest.skip('seeded default init differs from the minting torch build')
    torch.set_num_threads(1)
    with torch.no_grad():
        for tag in ('chop_a', 'chop_b'):                       # shave < 10 (+16) and shave >= 10 branches
            out = U.forward_chop(sd, torch.from_numpy(d[tag + '_x']))
            assert float(np.abs(out.numpy() - d[tag + '_out']).max()) < 1e-6, tag


def test_ssim_psnr_oracle_vs_brute_force_statement():
    """PSNR / SSIM stay "parity unpinned" (scikit-image is absent: util/index.py:76-81 calls it), so the oracle's uniform_filter formulation is checked
    against an independent brute-force statement of the same published definition (Wang et al. 2004 with skimage.metrics.structural_similarity's
    documented defaults: 7x7 uniform window, K1 = 0.01, K2 = 0.03, sample covariance NP / (NP - 1), windows fully inside the image, channel mean):
    every window summed explicitly in float64.  Plus the closed-form cases: identical images, a constant offset."""
    from oracle import metrics_ref as M
    rng = np.random.default_rng(7)
    for shape in [(3, 9, 11), (1, 7, 7), (4, 20, 13)]:
        x = np.clip(rng.uniform(-10, 265, size=shape), 0, 255).astype(np.float32)
        y = np.clip(x + rng.normal(0, 12, size=shape), 0, 255).astype(np.float32)
        C1, C2 = (0.01 * 255.0) ** 2, (0.03 * 255.0) ** 2
        per_channel = []
        for c in range(shape[0]):
            vals = []
            for i in range(shape[1] - 6):
                for j in range(shape[2] - 6):
                    a = x[c, i:i + 7, j:j + 7].astype(np.float64).ravel()
                    b = y[c, i:i + 7, j:j + 7].astype(np.float64).ravel()
                    ua, ub = a.mean(), b.mean()
                    va, vb = a.var(ddof=1), b.var(ddof=1)
                    vab = np.sum((a - ua) * (b - ub)) / 48.0
                    vals.append(((2 * ua * ub + C1) * (2 * vab + C2)) / ((ua * ua + ub * ub + C1) * (va + vb + C2)))
            per_channel.append(np.mean(vals))
        assert abs(M.ssim(x, y) - float(np.mean(per_channel))) < 1e-10
        assert abs(M.psnr(x, y) - 10 * np.log10(255.0 ** 2 / np.mean((x.astype(np.float64) - y.astype(np.float64)) ** 2))) < 1e-12
        assert abs(M.ssim(x, x) - 1.0) < 1e-12
    flat = np.full((2, 16, 16), 100.0, np.float32)
    assert abs(M.psnr(flat, flat + 5.0) - 20 * np.log10(255.0 / 5.0)) < 1e-12
