"""CPU oracle for the ELD noise sampler -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

NumPy float32 restatement of the reference's per-pixel noise synthesis and of the
host-side helpers around it.  Every function cites the reference lines it follows
(paths relative to the reference checkout).

Pinned parts (checked against the reference itself run in the build container, vectors
committed under tests/golden/ by oracle/gen_golden.py):
    * noise_arith() for models g / Pg / pg          <- noise.py:149-170
    * sample_params()                               <- noise.py:201-225
    * pack_raw_bayer() / unpack_raw_bayer()         <- noise.py:10-20, 66-81
    * pack_raw_xtrans() / unpack_raw_xtrans()       <- noise.py:22-64, 83-127
    * lmdb_decode_u16()                             <- dataset/lmdb_dataset.py:35-39
    * augment()                                     <- dataset/sid_dataset.py:344-352
PARITY UNPINNED (no reference code, test or vector exists -- README.md:41, noise.py:173):
    * Tukey-lambda read noise, row noise, quantisation noise, colour bias, and
      sample_params_full().  They follow the ELD paper (Wei et al., CVPR'20 / TPAMI'21)
      and the parameter schema of camera_params/release/*.npy.

dtype contract: the reference was written for NumPy-1.x value-based casting, i.e. the
whole chain evaluates in float32.  The oracle therefore evaluates every
op in float32 with one rounding per op (no FMA), which is what the reference computes
when its scalar params are passed as np.float32 -- that is how gen_golden.py runs it.
"""
import numpy as np

from . import philox_ref as px

# model-term flags; shared verbatim with include/eld_amd.h
SHOT_POISSON = 1    # 'P'  noise.py:158-159
SHOT_GAUSS = 2      # 'p'  noise.py:160-161
READ_GAUSS = 4      # 'g'  noise.py:165-166
READ_TL = 8         # 'G'  Tukey-lambda read noise           [unpinned]
ROW = 16            # 'R'  per-sensor-row Gaussian banding   [unpinned]
QUANT = 32          # 'U'  uniform quantisation noise        [unpinned]
CBIAS = 64          # 'B'  per-channel colour bias           [unpinned]
CLIP = 128          # fused caller-side clip to [0,1]  (dataset/sid_dataset.py:277)

F32 = np.float32


def model_flags(model):
    """Letter-containment parse of the model string, as noise.py:158-166 does
    ('P' wins over 'p'); G/R/U/B are the withheld-model letters."""
    f = 0
    if 'P' in model:
        f |= SHOT_POISSON
    elif 'p' in model:
        f |= SHOT_GAUSS
    if 'g' in model:
        f |= READ_GAUSS
    if 'G' in model:
        f |= READ_TL
    if 'R' in model:
        f |= ROW
    if 'U' in model:
        f |= QUANT
    if 'B' in model:
        f |= CBIAS
    return f


class Params(dict):
    """Per-image parameter record (mirror of EldNoiseParams in include/eld_amd.h)."""
    FIELDS = ('K', 'g_scale', 'tl_lambda', 'tl_scale', 'row_scale', 'q_step', 'saturation', 'ratio')

    def __init__(self, K=1.0, g_scale=0.0, saturation=15583.0, ratio=1.0, tl_lambda=0.0, tl_scale=0.0,
                 row_scale=0.0, q_step=1.0, color_bias=(0.0, 0.0, 0.0, 0.0)):
        super().__init__(K=float(K), g_scale=float(g_scale), tl_lambda=float(tl_lambda), tl_scale=float(tl_scale),
                         row_scale=float(row_scale), q_step=float(q_step), saturation=float(saturation),
                         ratio=float(ratio), color_bias=tuple(float(b) for b in color_bias))

    @classmethod
    def from_tuple(cls, t):
        """(K, g_scale, saturation_level, ratio) as returned by noise.py:225."""
        K, g, sat, ratio = t
        return cls(K=K, g_scale=g, saturation=sat, ratio=ratio)


# --------------------------------------------------------------------------------------
# deterministic arithmetic (parity notion E-1)
# --------------------------------------------------------------------------------------
def noise_arith(y, p, flags, counts=None, n_shot=None, n_read=None, t_tl=None, n_row=None, u_q=None):
    """z = f(y; params, variates) in strict float32, op-for-op as noise.py:155-169.

    y       float32 (C,H,W) in [0,1]
    counts  Poisson counts (any numeric dtype)      used when flags & SHOT_POISSON
    n_shot  standard normals, shape of y            used when flags & SHOT_GAUSS
    n_read  standard normals, shape of y            used when flags & READ_GAUSS
    t_tl    Tukey-lambda(lambda) variates (unit scale), shape of y     [unpinned term]
    n_row   standard normals PER ELEMENT (already broadcast along rows) [unpinned term]
    u_q     uniforms in [0,1), shape of y                                [unpinned term]
    """
    y = np.asarray(y, dtype=F32)
    S, r, K = F32(p['saturation']), F32(p['ratio']), F32(p['K'])
    y1 = (y * S).astype(F32)                       # noise.py:155
    y2 = (y1 / r).astype(F32)                      # noise.py:156
    if flags & SHOT_POISSON:                       # noise.py:158-159
        z = (np.asarray(counts).astype(F32) * K).astype(F32)
    elif flags & SHOT_GAUSS:                       # noise.py:160-161
        sd = np.sqrt(np.maximum((K * y2).astype(F32), F32(1e-10))).astype(F32)
        z = (y2 + (np.asarray(n_shot, F32) * sd).astype(F32)).astype(F32)
    else:                                          # noise.py:162-163
        z = y2
    if flags & READ_GAUSS:                         # noise.py:165-166
        g = np.maximum(F32(p['g_scale']), F32(1e-10))
        z = (z + (np.asarray(n_read, F32) * g).astype(F32)).astype(F32)
    if flags & READ_TL:                            # [unpinned] App. A-2: N_read ~ TL(lambda; scale)
        z = (z + (np.asarray(t_tl, F32) * F32(p['tl_scale'])).astype(F32)).astype(F32)
    if flags & ROW:                                # [unpinned] App. A-2: N_row ~ N(0, sigma_r) per sensor row
        z = (z + (np.asarray(n_row, F32) * F32(p['row_scale'])).astype(F32)).astype(F32)
    if flags & QUANT:                              # [unpinned] App. A-2: N_q ~ U(-q/2, q/2)
        z = (z + ((np.asarray(u_q, F32) - F32(0.5)).astype(F32) * F32(p['q_step'])).astype(F32)).astype(F32)
    if flags & CBIAS:                              # [unpinned] per-channel DC offset in ADU
        cb = np.asarray(p['color_bias'], F32).reshape(-1, 1, 1)
        z = (z + cb).astype(F32)
    z = (z * r).astype(F32)                        # noise.py:168
    z = (z / S).astype(F32)                        # noise.py:169
    if flags & CLIP:                               # dataset/sid_dataset.py:277
        z = np.maximum(np.minimum(z, F32(1.0)), F32(0.0)).astype(F32)
    return z


def poisson_lambda(y, p):
    """float32 rate handed to the Poisson draw: (y*S/r)/K  (noise.py:155-159)."""
    y = np.asarray(y, dtype=F32)
    y2 = ((y * F32(p['saturation'])).astype(F32) / F32(p['ratio'])).astype(F32)
    return (y2 / F32(p['K'])).astype(F32)


def sensor_row_index(C, H):
    """Packed (c,h) -> sensor row of the Bayer mosaic.  From the pack map noise.py:16-19:
    channels 0,1 come from even mosaic rows 2h, channels 2,3 from odd rows 2h+1."""
    assert C == 4, "row noise is defined on the 4-channel Bayer packing"
    c = np.arange(C).reshape(C, 1)
    h = np.arange(H).reshape(1, H)
    return (2 * h + (c >> 1)).astype(np.int64)      # (C,H)


# --------------------------------------------------------------------------------------
# reference-order NumPy RNG replay (pins noise_arith against the reference, bit-exact)
# --------------------------------------------------------------------------------------
def noise_numpy_rng(y, model, params):
    """Same draws, same order as NoiseModelBase.__call__ (noise.py:158-166) on the GLOBAL
    legacy NumPy RandomState: Poisson block first, then the randn block.  Returns
    (z, variates) so the variates can be re-injected into the HIP kernel."""
    flags = model_flags(model) & (SHOT_POISSON | SHOT_GAUSS | READ_GAUSS)
    p = params if isinstance(params, dict) else Params.from_tuple(params)
    v = {}
    if flags & SHOT_POISSON:
        v['counts'] = np.random.poisson(poisson_lambda(y, p))
    elif flags & SHOT_GAUSS:
        v['n_shot'] = np.random.randn(*y.shape).astype(F32)
    if flags & READ_GAUSS:
        v['n_read'] = np.random.randn(*y.shape).astype(F32)
    return noise_arith(y, p, flags, **v), v


def sample_params(camera_params, cameras):
    """Restatement of NoiseModel._sample_params (noise.py:201-225): 5 draws from the global
    RandomState in this order: choice(cameras), choice(profiles), uniform, standard_normal,
    uniform.  Kmin/Kmax are read but unused there (noise.py:209-210,214-215)."""
    camera = np.random.choice(cameras)                         # :202
    saturation_level = 16383 - 800                             # :205
    profile = np.random.choice(['Profile-1'])                  # :206,211
    cp = camera_params[camera][profile]                        # :212
    log_K = np.random.uniform(low=np.log(1e-1), high=np.log(30))           # :215
    log_g = np.random.standard_normal() * cp['g_scale']['sigma'] * 1 + \
        cp['g_scale']['slope'] * log_K + cp['g_scale']['bias']            # :217-218
    K = np.exp(log_K)                                          # :220
    g_scale = np.exp(log_g)                                    # :221
    ratio = np.random.uniform(low=100, high=300)               # :223
    return (K, g_scale, saturation_level, ratio)               # :225


def sample_params_full(camera_params, cameras, rng):
    """[PARITY UNPINNED] joint parameter sampling of the full ELD model (paper sec. 4;
    SURVEY.md App. A-2): log K ~ U(ln .1, ln 30) (kept from noise.py:215),
    log sigma_TL | log K ~ N(slope*logK + bias, sigma) with the 'G_scale' regression,
    same form with 'R_scale' for the row noise, lambda ~ uniform choice from 'G_shape',
    colour bias row chosen with the same index as lambda.  `rng` is a np.random.Generator."""
    camera = cameras[int(rng.integers(len(cameras)))]
    cp = camera_params[camera]
    prof = cp['Profile-1']
    log_K = rng.uniform(np.log(1e-1), np.log(30))
    def reg(name):
        r = prof[name]
        return float(np.exp(rng.standard_normal() * r['sigma'] + r['slope'] * log_K + r['bias']))
    g_scale, tl_scale, row_scale = reg('g_scale'), reg('G_scale'), reg('R_scale')
    i = int(rng.integers(len(cp['G_shape'])))
    ratio = rng.uniform(100, 300)
    return Params(K=float(np.exp(log_K)), g_scale=g_scale, saturation=16383 - 800, ratio=ratio,
                  tl_lambda=float(cp['G_shape'][i]), tl_scale=tl_scale, row_scale=row_scale, q_step=1.0,
                  color_bias=tuple(float(b) for b in np.asarray(cp['color_bias'])[i]))


# --------------------------------------------------------------------------------------
# variate transforms the HIP kernel implements (Philox-driven; parity notion E-2/E-3)
# --------------------------------------------------------------------------------------
def tukey_lambda_quantile(u, lam):
    """Q(u) = (u^lam - (1-u)^lam)/lam, lam=0 -> ln(u/(1-u)).  [unpinned term]
    Checked against scipy.stats.tukeylambda.ppf in tests.  `one_minus_u` is formed exactly
    (u01(~w)) by the callers that start from words; here it is 1-u in float32."""
    u = np.asarray(u, F32)
    v = (F32(1.0) - u).astype(F32)
    return _tl_from_uv(u, v, lam)


def _tl_from_uv(u, v, lam):
    lam = F32(lam)
    lu, lv = np.log2(u).astype(F32), np.log2(v).astype(F32)
    if lam == 0:
        return ((lu - lv).astype(F32) * F32(np.log(2.0))).astype(F32)
    a = np.exp2((lam * lu).astype(F32)).astype(F32)
    b = np.exp2((lam * lv).astype(F32)).astype(F32)
    return ((a - b).astype(F32) / lam).astype(F32)


def _log1pmx(x):
    """log1p(x) - x, stable in float32 (series for |x|<0.125, direct otherwise)."""
    x = np.asarray(x, F32)
    small = np.abs(x) < F32(0.125)
    xs = np.where(small, x, F32(0))
    # -x^2/2 + x^3/3 - ... up to x^9/9 (Horner)
    s = np.zeros_like(xs)
    for n in range(9, 1, -1):
        s = ((s * xs).astype(F32) + F32(((-1.0) ** (n + 1)) / n)).astype(F32)
    s = ((s * xs).astype(F32) * xs).astype(F32)
    xd = np.where(small, F32(0), x)
    d = (np.log1p(xd).astype(F32) - xd).astype(F32)
    return np.where(small, s, d).astype(F32)


def _pois_logpmf(k, lam, loglam):
    """-lam + k*log(lam) - lgamma(k+1) without catastrophic cancellation in float32:
    k*(log1p(x)-x) - 0.5*log(2*pi*k) - (1/(12k) - 1/(360k^3) + 1/(1260k^5)),  x=(lam-k)/k ; k=0 -> -lam."""
    k = np.asarray(k, F32)
    kk = np.maximum(k, F32(1))
    x = ((lam - kk) / kk).astype(F32)
    rk = (F32(1) / kk).astype(F32)
    rk2 = (rk * rk).astype(F32)
    corr = (rk * (F32(1 / 12.0) - rk2 * (F32(1 / 360.0) - rk2 * F32(1 / 1260.0)))).astype(F32)
    v = (kk * _log1pmx(x) - F32(0.5) * np.log(F32(2 * np.pi) * kk).astype(F32) - corr).astype(F32)
    return np.where(k < F32(0.5), -lam, v).astype(F32)


POIS_SMALL = 32.0   # below: alias table of Poisson(floor(lam)) + inversion at the fractional rate; at or above: PTRS (valid from 10)
POIS_KMAX = 96


def poisson_lambda_fast(y, p):
    """Rate on the Philox path.  Rounds 2-3 of the HIP sampler formed it as y * c with one per-image constant (an ulp off the reference's rate);
    since round 4 the kernel evaluates the reference's own chain ((y*S)/r)/K with correctly rounded divisions, i.e. exactly poisson_lambda
    (noise.py:155-159).  The name is kept for the callers."""
    return poisson_lambda(y, p)


def _pois_inversion(lam, u):
    """CDF inversion by sequential search with one uniform (float32), in the form the kernel uses (rates below 1 there):
    r = u - p0; while r > 0: k += 1; p *= lam * fp32(1/k); r -= p."""
    lam = np.asarray(lam, F32)
    p = np.exp(-lam).astype(F32)
    r = (u - p).astype(F32)
    k = np.zeros(lam.shape, np.int32)
    it = 0
    while (r > 0).any() and it < POIS_KMAX:
        it += 1
        active = r > 0
        k = np.where(active, k + 1, k)
        p = (p * (lam * F32(F32(1.0) / F32(it))).astype(F32)).astype(F32)
        r = np.where(active, (r - p).astype(F32), r)
    return k


def _ptrs_consts(lam):
    slam = np.sqrt(lam).astype(F32)
    b = (F32(0.931) + F32(2.53) * slam).astype(F32)
    a = (F32(-0.059) + F32(0.02483) * b).astype(F32)
    invalpha = (F32(1.1239) + F32(1.1328) / (b - F32(3.4))).astype(F32)
    vr = (F32(0.9277) - F32(3.6224) / (b - F32(2.0))).astype(F32)
    return a, b, invalpha, vr


def _ptrs_k(lam, U01, a, b):
    U = (U01 - F32(0.5)).astype(F32)
    us = (F32(0.5) - np.abs(U)).astype(F32)
    k = np.floor(((F32(2) * a / us + b).astype(F32) * U + lam + F32(0.43)).astype(F32)).astype(F32)
    return k, us


def _ptrs_slow(lam, k, us, V, a, b, invalpha):
    """Slow accept test of PTRS (Hoermann 1993; the transformed-rejection sampler NumPy's legacy
    rk_poisson_ptrs uses), float32."""
    rej = (k < 0) | ((us < F32(0.013)) & (V > us))
    lhs = (np.log(V).astype(F32) + np.log(invalpha).astype(F32)
           - np.log((a / (us * us).astype(F32) + b).astype(F32)).astype(F32)).astype(F32)
    rhs = _pois_logpmf(np.maximum(k, 0), lam, None)
    return (~rej) & (lhs <= rhs)


def _ptrs_attempt(lam, U01, V):
    a, b, invalpha, vr = _ptrs_consts(lam)
    k, us = _ptrs_k(lam, U01, a, b)
    ok = ((us >= F32(0.07)) & (V <= vr)) | _ptrs_slow(lam, k, us, V, a, b, invalpha)
    return k.astype(np.int32), ok


_ALIAS = None


def _alias_tables():
    global _ALIAS
    if _ALIAS is None:
        from . import gen_poisson_alias
        _ALIAS = gen_poisson_alias.tables()
    return _ALIAS


def _pois_table(lam, wu, wv, elem, seed, sample_id):
    """lam < 32 (eld_amd/csrc/noise.hip phase 1): X = A_n(wu) + Inv(lam - n, u01(wv)), n = floor(lam).  A_n: alias table of
    Poisson(n) (oracle/gen_poisson_alias.py) indexed by the top 6 bits of the word, accept test on the low 26; outcome 63 is
    the tail {X >= 63}, finished by inversion from 63 with word x of retry call 64 of the element."""
    ent, q0 = _alias_tables()
    lam = np.asarray(lam, F32)
    n = lam.astype(np.int32)
    d = (lam - n.astype(F32)).astype(F32)
    wu = np.asarray(wu, np.uint32)
    j = (wu >> np.uint32(26)).astype(np.int64)
    e = ent[n, j]
    k0 = np.where((wu << np.uint32(6)) < (e & np.uint32(0xFFFFFFC0)), j, (e & np.uint32(63)).astype(np.int64)).astype(np.int32)
    t = np.nonzero(k0 == 63)[0]
    if t.size:
        words = px.sampler_words(elem[t].astype(np.uint32), sample_id, px.STREAM_POIS_R, seed, it=np.uint32(64))
        u = px.u01(words[0])
        for i, idx in enumerate(t):
            q = F32(q0[n[idx]])
            r = F32(u[i] - q)
            k = 63
            while r > 0 and k < 255:
                k += 1
                q = F32(q * F32(F32(n[idx]) / F32(k)))
                r = F32(r - q)
            k0[idx] = k
    return k0 + _pois_inversion(d, px.u01(wv))


def _poisson_philox(lam, wu, wv, elem, seed, sample_id):
    """Poisson counts under the sampler's word usage (wu / wv: streams POIS_U / POIS_V, group e//4, word e%4).
    lam < 32: alias table + fractional-rate inversion (_pois_table).  Otherwise PTRS: attempt 0 takes U = u01(wu), V = u01(wv);
    attempts 2k+1, 2k+2 take words (x,y), (z,w) of the per-element retry call k (stream POIS_R, index = element, iter = k)."""
    n = lam.size
    k_out = np.zeros(n, np.int32)
    small = lam < F32(POIS_SMALL)
    k_out[small] = _pois_table(lam[small], wu[small], wv[small], elem[small], seed, sample_id)
    idx = np.nonzero(~small)[0]
    if idx.size == 0:
        return k_out
    k, ok = _ptrs_attempt(lam[idx], px.u01(wu[idx]), px.u01(wv[idx]))
    k_out[idx[ok]] = k[ok]
    rem = idx[~ok]
    call = 0
    while rem.size and call < 64:
        words = px.sampler_words(elem[rem].astype(np.uint32), sample_id, px.STREAM_POIS_R, seed, it=np.uint32(call))
        for a_ in (0, 2):
            if not rem.size:
                break
            k, ok = _ptrs_attempt(lam[rem], px.u01(words[a_]), px.u01(words[a_ + 1]))
            k_out[rem[ok]] = k[ok]
            rem = rem[~ok]
            words = tuple(x[~ok] for x in words)
        call += 1
    return k_out


def philox_variates(shape, p, flags, seed, sample_id, y=None):
    """Variates for one image under the sampler's Philox layout (oracle/philox_ref.py).
    Transcendentals here are NumPy's; the HIP kernel uses gfx950 hardware log2/exp2/sin/cos,
    so Philox-mode comparisons of the *variates* are tolerance-based, while the raw words
    are bit-exact (tests/test_noise_gpu.py)."""
    C, H, W = shape
    n = C * H * W
    e = np.arange(n, dtype=np.uint32)
    g, j = e >> np.uint32(2), (e & np.uint32(3)).astype(np.int64)
    ng = (n + 3) // 4

    def group_words(stream):
        w = np.stack(px.sampler_words(np.arange(ng, dtype=np.uint32), sample_id, stream, seed), axis=1)  # (ng,4)
        return w[g.astype(np.int64), j]

    out = {}
    if flags & READ_TL:
        w = group_words(px.STREAM_TL)
        out['t_tl'] = _tl_from_uv(px.u01(w), px.u01(~w), p['tl_lambda']).reshape(shape)
    if flags & QUANT:
        if (flags & READ_TL) and (flags & SHOT_POISSON):
            # full model: 18 leftover bits -- low 9 of the Tukey-lambda word, low 9 of the Poisson V word (u01 takes w >> 9)
            bits = ((group_words(px.STREAM_TL) & np.uint32(511)) << np.uint32(9)) | (group_words(px.STREAM_POIS_V) & np.uint32(511))
            out['u_q'] = (bits.astype(F32) * F32(2.0 ** -18)).reshape(shape)
        else:
            out['u_q'] = px.u01_closed_open(group_words(px.STREAM_QUANT)).reshape(shape)
    for flag, stream, name in ((READ_GAUSS, px.STREAM_NREAD, 'n_read'), (SHOT_GAUSS, px.STREAM_NSHOT, 'n_shot')):
        if flags & flag:
            w = np.stack(px.sampler_words(np.arange(ng, dtype=np.uint32), sample_id, stream, seed), axis=1)
            n0, n1 = px.box_muller(w[:, 0], w[:, 1])
            n2, n3 = px.box_muller(w[:, 2], w[:, 3])
            out[name] = np.stack([n0, n1, n2, n3], axis=1).reshape(-1)[:n].reshape(shape)
    if flags & ROW:
        rows = sensor_row_index(C, H)                               # (C,H)
        w = px.sampler_words(np.arange(2 * H, dtype=np.uint32), sample_id, px.STREAM_ROW, seed)
        nrm, _ = px.box_muller(w[0], w[1])
        out['row_normals'] = nrm                                   # (2H,)
        out['n_row'] = np.broadcast_to(nrm[rows][:, :, None], shape).astype(F32)
    if flags & SHOT_POISSON:
        lam = poisson_lambda_fast(y, p).reshape(-1)
        out['counts'] = _poisson_philox(lam, group_words(px.STREAM_POIS_U), group_words(px.STREAM_POIS_V), e, seed, sample_id).reshape(shape)
    return out


def noise_philox(y, p, flags, seed, sample_id):
    """Whole sampler under Philox (what the HIP kernel computes, up to transcendental ulps)."""
    v = philox_variates(y.shape, p, flags, seed, sample_id, y=y)
    v.pop('row_normals', None)
    return noise_arith(y, p, flags, **v), v


def noise_numpy_full(y, p, flags, rng=None):
    """[unpinned terms] CPU port of the FULL model the way the reference would write it: NumPy legacy
    RandomState draws (np.random.poisson as noise.py:159, scipy-free Tukey-lambda quantile of a uniform
    draw, one randn per sensor row, uniform quantisation).  Used as bench.py's cpu_baseline ("port")."""
    rs = np.random if rng is None else rng
    v = {}
    C, H, W = y.shape
    if flags & SHOT_POISSON:
        v['counts'] = rs.poisson(poisson_lambda(y, p))
    elif flags & SHOT_GAUSS:
        v['n_shot'] = rs.randn(*y.shape).astype(F32)
    if flags & READ_GAUSS:
        v['n_read'] = rs.randn(*y.shape).astype(F32)
    if flags & READ_TL:
        v['t_tl'] = tukey_lambda_quantile(rs.uniform(size=y.shape).astype(F32).clip(1e-7, 1 - 1e-7), p['tl_lambda'])
    if flags & ROW:
        nr = rs.randn(2 * H).astype(F32)
        v['n_row'] = np.broadcast_to(nr[sensor_row_index(C, H)][:, :, None], y.shape)
    if flags & QUANT:
        v['u_q'] = rs.uniform(size=y.shape).astype(F32)
    return noise_arith(y, p, flags, **v)


# --------------------------------------------------------------------------------------
# integer / indexing helpers either side of the sampler (bit-exact)
# --------------------------------------------------------------------------------------
def pack_raw_bayer(cfa_img):
    """(H,W) mosaic -> (4,H/2,W/2): R G1 B G2 quads.  noise.py:10-20."""
    m = np.asarray(cfa_img)
    return np.stack((m[0::2, 0::2], m[0::2, 1::2], m[1::2, 1::2], m[1::2, 0::2]), axis=0).astype(F32)


def unpack_raw_bayer(img4c):
    """(4,h,w) -> (2h,2w) mosaic.  noise.py:66-81."""
    _, h, w = img4c.shape
    out = np.zeros((2 * h, 2 * w), F32)
    out[0::2, 0::2], out[0::2, 1::2], out[1::2, 1::2], out[1::2, 0::2] = img4c[0], img4c[1], img4c[2], img4c[3]
    return out


# X-Trans (noise.py:22-64, 83-127): 6x6 colour cell <-> 9 planes at 1/3 resolution.  Planes 0-4 hold one cell position per
# (row parity, column parity) of the packed coordinate; planes 5-8 are the four positions (1|2, 0|1) of every 3x3 block.
# XTRANS_RC[c][pi][pj] = (row, column) inside the 6x6 cell for packed position (2a + pi, 2b + pj) of plane c < 5.
XTRANS_RC = (
    (((0, 0), (0, 4)), ((3, 1), (3, 3))),
    (((0, 2), (0, 5)), ((3, 2), (3, 5))),
    (((0, 1), (0, 3)), ((3, 0), (3, 4))),
    (((1, 2), (2, 5)), ((5, 2), (4, 5))),
    (((2, 2), (1, 5)), ((4, 2), (5, 5))),
)
XTRANS_RC3 = ((1, 0), (1, 1), (2, 0), (2, 1))      # planes 5..8: (row, column) inside the 3x3 block


def xtrans_source_index(h, w):
    """(9,h,w) arrays of the mosaic row / column each packed position reads (pack) or writes (unpack)."""
    i, j = np.meshgrid(np.arange(h), np.arange(w), indexing='ij')
    rows, cols = np.zeros((9, h, w), np.int64), np.zeros((9, h, w), np.int64)
    for c in range(5):
        t = np.array(XTRANS_RC[c])                  # [pi][pj][2]
        rows[c] = 6 * (i // 2) + t[i % 2, j % 2, 0]
        cols[c] = 6 * (j // 2) + t[i % 2, j % 2, 1]
    for c in range(5, 9):
        rows[c] = 3 * i + XTRANS_RC3[c - 5][0]
        cols[c] = 3 * j + XTRANS_RC3[c - 5][1]
    return rows, cols


def pack_raw_xtrans(cfa_img):
    """(H,W) mosaic -> (9, 2*(H//6), 2*(W//6)) float32.  noise.py:22-64 (sides truncated to multiples of 6, :25-26)."""
    m = np.asarray(cfa_img)
    rows, cols = xtrans_source_index(2 * (m.shape[0] // 6), 2 * (m.shape[1] // 6))
    return m[rows, cols].astype(F32)


def unpack_raw_xtrans(img9c):
    """(9,h,w) -> (3h,3w) mosaic.  noise.py:83-127."""
    _, h, w = img9c.shape
    rows, cols = xtrans_source_index(h, w)
    out = np.zeros((3 * h, 3 * w), F32)
    out[rows, cols] = img9c
    return out


def pack_raw_sid(im, raw_pattern, black_level, white_point=16383):
    """dataset/sid_dataset.py:172-196 pack_raw_bayer: mosaic (H,W) -> (4,H/2,W/2) in R, G1, B, G2 order (positions of colour codes
    0..3 in the 2x2 raw_pattern), (x - black) / (white_point - black) per channel in float32, clipped to [0,1]."""
    im = np.asarray(im).astype(F32)
    pat = np.asarray(raw_pattern)
    H, W = im.shape
    chans = []
    for k in range(4):
        oy, ox = np.where(pat == k)
        chans.append(im[oy[0]:H:2, ox[0]:W:2])
    out = np.stack(chans, axis=0).astype(F32)
    black = np.asarray(black_level, F32)[:, None, None]
    out = (out - black) / (F32(white_point) - black)
    return np.clip(out, 0, 1).astype(F32)


def lmdb_decode_u16(x):
    """uint16 -> float32 in [0,1]: clip(x/65535, 0, 1) evaluated in float64 then cast
    (dataset/lmdb_dataset.py:38-39).  An fp32 true division float(u16)/65535.0f gives the
    same bits for all 65,536 codes -- that is what the kernel does."""
    return np.clip(np.asarray(x, np.uint16) / 65535, 0, 1).astype(F32)


def augment(x, flip_h, flip_w, transpose):
    """ELDTrainDataset augmentation (dataset/sid_dataset.py:344-352) on a (C,H,W) array,
    applied in the reference's order: flip axis1, flip axis2, transpose(0,2,1)."""
    if flip_h:
        x = np.flip(x, axis=1)
    if flip_w:
        x = np.flip(x, axis=2)
    if transpose:
        x = np.transpose(x, (0, 2, 1))
    return np.ascontiguousarray(x)
