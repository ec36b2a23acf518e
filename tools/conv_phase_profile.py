"""Dev tool: where do conv_x3_kernel's cycles go?  s_memtime stamps per stage (barrier-1 wait, staging stores, barrier-2 wait,
first-tap LDS reads, MFMA phase, gap to the next stage) for the first 8 workgroups of one layer launch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import eld_amd
from eld_amd import _lib as L
lib = eld_amd.load_library()
N, H, W, Ci, Co = (int(v) for v in sys.argv[1:6]) if len(sys.argv) > 5 else (8, 178, 266, 256, 256)
x = torch.randn(N, H, W, Ci, device='cuda'); w = torch.randn(Co, Ci, 3, 3, device='cuda') * 0.02; b = torch.zeros(Co, device='cuda')
out = torch.empty(N, H, W, Co, device='cuda')
ws = torch.empty(lib.eld_layer_workspace_bytes(N, H, W, Ci, Co), dtype=torch.uint8, device='cuda')
prof = torch.zeros(8 * 4 * 128 * 6, dtype=torch.int64, device='cuda')
def run():
    L.check(lib.eld_conv3x3_forward(L.dptr(x), Ci, None, 0, L.dptr(w), L.dptr(b), L.dptr(out), N, H, W, Co, 1, L.dptr(ws), ws.numel(), L.cur_stream()))
run(); torch.cuda.synchronize()
lib.eld_debug_conv_prof(L.dptr(prof))
run(); torch.cuda.synchronize()
lib.eld_debug_conv_prof(None)
p = prof.cpu().numpy().reshape(8, 4, 128, 6).astype(np.float64)
ok = p[..., 5] > 0
names = ['barrier1 wait', 'staging (split+ds_write, drained)', 'barrier2 wait', 'first tap reads', 'MFMA phase (72)', 'gap to next stage']
d = [p[..., 1] - p[..., 0], p[..., 2] - p[..., 1], p[..., 3] - p[..., 2], p[..., 4] - p[..., 3], p[..., 5] - p[..., 4]]
gap = p[:, :, 1:, 0] - p[:, :, :-1, 5]
sel = ok[:, :, 1:-1]
print('layer %dx%dx%d %d->%d; s_memtime ticks (100 MHz constant clock?) per stage, mean / median over %d samples' % (N, H, W, Ci, Co, int(sel.sum())))
for n, v in zip(names[:5], d):
    v = v[:, :, 1:-1][sel]
    print('  %-36s mean %8.1f  median %8.1f  p90 %8.1f' % (n, v.mean(), np.median(v), np.percentile(v, 90)))
g = gap[:, :, :-1][sel[:, :, :]] if gap[:, :, :-1].shape == sel.shape else gap[ok[:, :, 1:] & ok[:, :, :-1]]
print('  %-36s mean %8.1f  median %8.1f  p90 %8.1f' % (names[5], g.mean(), np.median(g), np.percentile(g, 90)))
tot = (p[:, :, 2:, 0] - p[:, :, 1:-1, 0])[sel]
print('  %-36s mean %8.1f  median %8.1f' % ('stage period', tot.mean(), np.median(tot)))
# relative phase of the two workgroups sharing a CU is unknown; print the first stamps of each workgroup
print('first stamps per workgroup (wave 0):', (p[:, 0, 0, 0] - p[:, 0, 0, 0].min()).astype(int).tolist())
