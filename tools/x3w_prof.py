"""dev: stage timeline of conv_x3w_kernel (dev library: tools/build_dev.sh; ELD_AMD_LIB=tools/probe/libeld_dev.so).  Prints, for workgroup 0, the cycles between
the stamps of the first consumer wave (1 stage start, 2 MFMAs done / at the barrier, 3 epilogue start, 4 epilogue end), the first halo wave (10 stage start, 11 cut
done, 12 loads issued, 13 at the barrier) and the slab wave (20, 21 issued, 22 landed)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import eld_amd
from eld_amd import _lib as L
lib = eld_amd.load_library()
N, H, W, C0, Co = 8, 1424, 2128, 32, 32
x0 = torch.randn(N, H, W, C0, device='cuda'); w = torch.randn(Co, C0, 3, 3, device='cuda') * 0.05; b = torch.randn(Co, device='cuda')
out = torch.empty(N, H, W, Co, device='cuda')
ws = torch.empty(lib.eld_layer_workspace_bytes(N, H, W, C0, Co), dtype=torch.uint8, device='cuda')
buf = torch.zeros(8 * 3 * 1024, dtype=torch.int64, device='cuda')


def fn():
    L.check(lib.eld_conv3x3_forward(L.dptr(x0), C0, None, 0, L.dptr(w), L.dptr(b), L.dptr(out), N, H, W, Co, 1, L.dptr(ws), ws.numel(), L.cur_stream()))
fn(); fn(); torch.cuda.synchronize()
lib.eld_debug_conv_prof(L.dptr(buf))
fn(); torch.cuda.synchronize()
lib.eld_debug_conv_prof(None)
a = buf.cpu().numpy().astype(np.uint64).reshape(8, 3, 1024)
for blk in (0, 3):
    for role, name in enumerate(('consumer', 'halo', 'slab')):
        v = a[blk, role]
        v = v[v != 0]
        tags = (v >> np.uint64(56)).astype(int)
        t = (v & np.uint64((1 << 56) - 1)).astype(np.int64)
        if len(t) == 0:
            print(name, 'no stamps'); continue
        t0 = t[0]
        print('wg %d %s: %d stamps, span %d cycles' % (blk, name, len(t), t[-1] - t0))
        # aggregate deltas by (tag_from, tag_to)
        agg = {}
        for i in range(1, len(t)):
            k = (tags[i - 1], tags[i])
            agg.setdefault(k, []).append(int(t[i] - t[i - 1]))
        for k in sorted(agg):
            d = np.array(agg[k])
            print('   %2d -> %2d : n %4d  mean %8.0f  median %8.0f  min %7d  max %8d  total %9d' % (k[0], k[1], len(d), d.mean(), np.median(d), d.min(), d.max(), d.sum()))
