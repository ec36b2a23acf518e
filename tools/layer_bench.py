"""Per-layer timing of the U-Net kernels at BASELINE.json config-2 shapes (dev tool): fwd / bwd-data / wgrad TF/s."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import eld_amd
from eld_amd import _lib as L
lib = eld_amd.load_library()
H0, W0 = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1424, 2128)
N = int(os.environ.get("LAYER_N", "1"))

def ev(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

layers = []   # name, level, C0, C1, Cout
ch = [32, 64, 128, 256, 512]
layers.append(('conv1_1', 0, 16, 0, 32)); layers.append(('conv1_2', 0, 32, 0, 32))
for l in range(1, 5):
    layers.append(('conv%d_1' % (l + 1), l, ch[l - 1], 0, ch[l])); layers.append(('conv%d_2' % (l + 1), l, ch[l], 0, ch[l]))
for i, l in enumerate(range(3, -1, -1)):
    layers.append(('conv%d_1' % (6 + i), l, ch[l], ch[l], ch[l])); layers.append(('conv%d_2' % (6 + i), l, ch[l], 0, ch[l]))
tot = {'fwd': 0.0, 'bwd': 0.0, 'wg': 0.0}
print('%-9s %11s %5s %5s | %8s %7s | %8s %7s | %8s %7s' % ('layer', 'HxW', 'Cin', 'Cout', 'fwd ms', 'TF/s', 'bwdd ms', 'TF/s', 'wgrad ms', 'TF/s'))
for name, l, C0, C1, Co in layers:
    H, W = H0 >> l, W0 >> l
    Cin = C0 + C1
    x0 = torch.randn(N, H, W, C0, device='cuda'); x1 = torch.randn(N, H, W, C1, device='cuda') if C1 else None
    w = torch.randn(Co, Cin, 3, 3, device='cuda') * 0.05; b = torch.randn(Co, device='cuda')
    out = torch.empty(N, H, W, Co, device='cuda'); g = torch.randn(N, H, W, Co, device='cuda')
    ws = torch.empty(lib.eld_layer_workspace_bytes(N, H, W, Cin, Co), dtype=torch.uint8, device='cuda')
    flop = 2.0 * N * H * W * Co * Cin * 9
    if name == 'conv1_1': flop = 2.0 * N * H * W * Co * 4 * 9
    tf = ev(lambda: L.check(lib.eld_conv3x3_forward(L.dptr(x0), C0, L.dptr(x1), C1, L.dptr(w), L.dptr(b), L.dptr(out), N, H, W, Co, 1, L.dptr(ws), ws.numel(), L.cur_stream())))
    tb = None
    if Cin % 32 == 0:
        d0 = torch.empty(N, H, W, C0, device='cuda'); d1 = torch.empty(N, H, W, C1, device='cuda') if C1 else None
        tb = ev(lambda: L.check(lib.eld_conv3x3_backward_data(L.dptr(g), L.dptr(w), L.dptr(d0), L.dptr(d1), C0, L.dptr(x0), None, N, H, W, Cin, Co, L.dptr(ws), ws.numel(), L.cur_stream())))
    dw = torch.empty(Co, Cin, 3, 3, device='cuda'); db = torch.empty(Co, device='cuda')
    tw = ev(lambda: L.check(lib.eld_conv3x3_backward_weight(L.dptr(g), L.dptr(x0), C0, L.dptr(x1), C1, L.dptr(dw), L.dptr(db), N, H, W, Co, L.dptr(ws), ws.numel(), L.cur_stream())))
    tot['fwd'] += tf; tot['wg'] += tw; tot['bwd'] += tb or 0
    print('%-9s %5dx%-5d %5d %5d | %8.3f %7.1f | %8s %7s | %8.3f %7.1f' % (name, H, W, Cin, Co, tf, flop / tf / 1e9,
          '%.3f' % tb if tb else '-', '%.1f' % (flop / tb / 1e9) if tb else '-', tw, flop / tw / 1e9))
    del x0, x1, out, g, ws
for i, l in enumerate(range(3, -1, -1)):
    H, W = H0 >> (l + 1), W0 >> (l + 1)
    Ci, Co = ch[l + 1], ch[l]
    x = torch.randn(N, H, W, Ci, device='cuda'); w = torch.randn(Ci, Co, 2, 2, device='cuda') * 0.05; b = torch.randn(Co, device='cuda')
    out = torch.empty(N, 2 * H, 2 * W, Co, device='cuda'); g = torch.randn(N, 2 * H, 2 * W, Co, device='cuda'); din = torch.empty(N, H, W, Ci, device='cuda')
    dw = torch.empty(Ci, Co, 2, 2, device='cuda'); db = torch.empty(Co, device='cuda')
    ws = torch.empty(lib.eld_layer_workspace_bytes(N, H, W, Ci, Co), dtype=torch.uint8, device='cuda')
    flop = 2.0 * N * H * W * Ci * Co * 4
    tf = ev(lambda: L.check(lib.eld_convt2x2_forward(L.dptr(x), L.dptr(w), L.dptr(b), L.dptr(out), N, H, W, Ci, Co, L.dptr(ws), ws.numel(), L.cur_stream())))
    tb = ev(lambda: L.check(lib.eld_convt2x2_backward_data(L.dptr(g), L.dptr(w), L.dptr(x), L.dptr(din), N, H, W, Ci, Co, L.dptr(ws), ws.numel(), L.cur_stream())))
    tw = ev(lambda: L.check(lib.eld_convt2x2_backward_weight(L.dptr(x), L.dptr(g), L.dptr(dw), L.dptr(db), N, H, W, Ci, Co, L.dptr(ws), ws.numel(), L.cur_stream())))
    tot['fwd'] += tf; tot['wg'] += tw; tot['bwd'] += tb
    print('%-9s %5dx%-5d %5d %5d | %8.3f %7.1f | %8.3f %7.1f | %8.3f %7.1f' % ('upv%d' % (6 + i), H, W, Ci, Co, tf, flop / tf / 1e9, tb, flop / tb / 1e9, tw, flop / tw / 1e9))
print('sum ms: fwd %.2f  bwd-data %.2f  wgrad %.2f' % (tot['fwd'], tot['bwd'], tot['wg']))
