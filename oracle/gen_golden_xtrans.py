#!/usr/bin/env python3
"""Mint tests/golden/rawpacker_xtrans.npz by running the reference's RawPacker('xtrans') (noise.py:22-64, 83-127) itself.
TEST INFRASTRUCTURE ONLY.      python oracle/gen_golden_xtrans.py [--ref /root/reference]

Cases: a mosaic whose sides are not multiples of 6 (the reference truncates to (H//6)*6 x (W//6)*6, noise.py:25-26), an exact
multiple, and an unpack of a packed tensor with odd h and w (3h x 3w mosaic, noise.py:87-88)."""
import argparse
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(os.path.dirname(HERE), 'tests', 'golden')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--ref', default='/root/reference')
    ref = os.path.abspath(ap.parse_args().ref)
    sys.path.insert(0, ref)
    import noise as ref_noise
    rp = ref_noise.RawPacker('xtrans')
    rng = np.random.default_rng(5)
    out = {}
    for name, hw in (('ragged', (20, 27)), ('exact', (12, 18))):
        m = rng.integers(0, 16383, size=hw).astype(np.float32)
        pk = rp.pack_raw(m)                       # <- the reference
        out[name + '_mosaic'] = m
        out[name + '_packed'] = pk
        out[name + '_unpacked'] = rp.unpack_raw(pk)
    odd = rng.integers(0, 16383, size=(9, 3, 5)).astype(np.float32)
    out['odd_packed'] = odd
    out['odd_unpacked'] = rp.unpack_raw(odd)      # <- the reference
    np.savez_compressed(os.path.join(GOLD, 'rawpacker_xtrans.npz'), **out)
    print('wrote rawpacker_xtrans.npz:', {k: v.shape for k, v in out.items()})


if __name__ == '__main__':
    main()
