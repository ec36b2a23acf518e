"""Philox4x32-R counter-based RNG, NumPy restatement (oracle; test infrastructure only).

Algorithm: Salmon et al., "Parallel Random Numbers: As Easy as 1, 2, 3" (SC'11),
Philox-4x32.  The sampler runs ROUNDS = 7 (the paper's smallest Crush-resistant count for
Philox-4x32; 10 is its default with a safety margin -- eld_amd/csrc/philox.h says why 7).
Both round counts are pinned by the Random123 known-answer vectors in
``tests/test_oracle_golden.py``.  The reference (noise.py:159,161,166) uses NumPy's
MT19937 global stream, which a counter-based GPU sampler cannot reproduce
(SURVEY.md F8); this file defines the stream the HIP sampler must reproduce
bit-for-bit instead.

Counter layout used by the sampler (eld_amd/csrc/philox.h mirrors this):
    key  = (seed_lo, seed_hi)
    ctr  = (index, sample_id_lo, sample_id_hi, stream | (iter << 8))
Streams (``index`` meaning in brackets):
    0 ROW    [sensor row 0..2H-1]   words 0,1 -> one Box-Muller normal
    1 TL     [group g = elem//4]    word j   -> Tukey-lambda uniform of elem 4g+j
    2 QUANT  [group]                word j   -> quantisation uniform
    3 NREAD  [group]                words    -> 4 normals (Gaussian read noise 'g')
    4 NSHOT  [group]                words    -> 4 normals (heteroscedastic shot 'p')
    5 POIS_U [group]                word j   -> Poisson attempt-0 U (also the inversion uniform when lam < 10)
    6 POIS_V [group]                word j   -> Poisson attempt-0 V (PTRS only)
    7 POIS_R [element], iter        words (U,V),(U,V) -> attempts 1+2*iter, 2+2*iter
"""
import numpy as np

M0 = np.uint64(0xD2511F53)
M1 = np.uint64(0xCD9E8D57)
W0 = 0x9E3779B9
W1 = 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)

STREAM_ROW, STREAM_TL, STREAM_QUANT, STREAM_NREAD, STREAM_NSHOT, STREAM_POIS_U, STREAM_POIS_V, STREAM_POIS_R = range(8)


ROUNDS = 7      # == ELD_PHILOX_ROUNDS (eld_amd/csrc/philox.h)


def philox4x32(c0, c1, c2, c3, k0, k1, rounds=None):
    """Vectorised Philox4x32-`rounds` (default: the sampler's ROUNDS).  All inputs broadcastable uint32-valued; returns 4 uint32 arrays."""
    rounds = ROUNDS if rounds is None else int(rounds)
    c0, c1, c2, c3 = [np.asarray(c, dtype=np.uint64) & MASK for c in np.broadcast_arrays(c0, c1, c2, c3)]
    k0 = int(k0) & 0xFFFFFFFF
    k1 = int(k1) & 0xFFFFFFFF
    for r in range(rounds):
        p0 = M0 * c0
        p1 = M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & MASK
        hi1, lo1 = p1 >> np.uint64(32), p1 & MASK
        c0, c1, c2, c3 = (hi1 ^ c1 ^ np.uint64(k0)), lo1, (hi0 ^ c3 ^ np.uint64(k1)), lo0
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return tuple(c.astype(np.uint32) for c in (c0, c1, c2, c3))


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    return philox4x32(c0, c1, c2, c3, k0, k1, rounds=10)


def sampler_words(index, sample_id, stream, seed, it=0):
    """4 words for (index, sample_id, stream[, iter]) under the sampler's counter layout."""
    sample_id = int(sample_id)
    seed = int(seed)
    return philox4x32(index, sample_id & 0xFFFFFFFF, (sample_id >> 32) & 0xFFFFFFFF,
                         np.uint32(stream) | (np.asarray(it, dtype=np.uint32) << np.uint32(8)),
                         seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)


def u01(w):
    """uint32 word -> float32 uniform in (0,1), 23-bit resolution, exactly representable:
    (w>>9)*2^-23 + 2^-24 = (2k+1)*2^-24.  u01(~w) == 1 - u01(w) exactly."""
    return ((np.asarray(w, dtype=np.uint32) >> np.uint32(9)).astype(np.float32) * np.float32(2.0 ** -23)
            + np.float32(2.0 ** -24)).astype(np.float32)


def u01_closed_open(w):
    """uint32 word -> float32 in [0,1): (w>>8)*2^-24."""
    return ((np.asarray(w, dtype=np.uint32) >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)).astype(np.float32)


def box_muller(wa, wb):
    """Two words -> two float32 standard normals (r*cos, r*sin)."""
    u1 = u01(wa)
    u2 = u01_closed_open(wb)
    r = np.sqrt(np.float32(-2.0) * np.log(u1)).astype(np.float32)
    ang = (np.float32(2.0 * np.pi) * u2).astype(np.float32)
    return (r * np.cos(ang)).astype(np.float32), (r * np.sin(ang)).astype(np.float32)
