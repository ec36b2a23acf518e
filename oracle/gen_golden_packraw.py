#!/usr/bin/env python3
"""Mint tests/golden/pack_raw.npz by running the reference's pack_raw_bayer (dataset/sid_dataset.py:172-196) on fake `raw`
objects (rawpy is absent: the function only touches raw_image_visible, raw_pattern and black_level_per_channel).
TEST INFRASTRUCTURE ONLY.      python oracle/gen_golden_packraw.py [--ref /root/reference]"""
import argparse
import io
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G      # noqa: E402  (install_stubs, GOLD)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--ref', default='/root/reference')
    ref = os.path.abspath(ap.parse_args().ref)
    sys.path.insert(0, ref)
    os.chdir(ref)
    G.install_stubs()
    _popen = os.popen
    os.popen = lambda cmd, *a, **k: io.StringIO('24 80') if 'stty' in cmd else _popen(cmd, *a, **k)
    import dataset.sid_dataset as ref_sid
    os.popen = _popen
    rng = np.random.default_rng(17)
    out = {}
    cases = [('rggb', [[0, 1], [3, 2]], [512, 512, 512, 512], (12, 20)),          # Sony: R G / G B
             ('grbg', [[1, 0], [2, 3]], [511.5, 513.25, 510.0, 512.75], (8, 6)),   # per-channel black levels
             ('bggr', [[2, 3], [1, 0]], [2047, 2048, 2049, 2050], (6, 10)),
             ('gbrg', [[3, 2], [0, 1]], [0, 64, 128, 256], (4, 4))]
    for name, pat, black, hw in cases:
        im = rng.integers(0, 16384, size=hw).astype(np.uint16)
        im.flat[:4] = [0, 16383, 400, 16000]                                      # below black / at white
        raw = types.SimpleNamespace(raw_image_visible=im, raw_pattern=np.array(pat), black_level_per_channel=list(black))
        res = ref_sid.pack_raw_bayer(raw)                                         # <- the reference
        assert res.dtype == np.float32
        out[name + '_im'] = im
        out[name + '_pattern'] = np.array(pat, np.int32)
        out[name + '_black'] = np.array(black, np.float32)
        out[name + '_out'] = res
    np.savez_compressed(os.path.join(G.GOLD, 'pack_raw.npz'), **out)
    print('wrote', os.path.join(G.GOLD, 'pack_raw.npz'))


if __name__ == '__main__':
    main()
