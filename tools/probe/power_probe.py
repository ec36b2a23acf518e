import sys, os
sys.path.insert(0, '/root/repo')
import torch, eld_amd
from eld_amd import _lib as L
lib = eld_amd.load_library()
N, H, W, Ci, Co = 8, 178, 266, 256, 256
ws = torch.empty(lib.eld_layer_workspace_bytes(N, H, W, Ci, Co), dtype=torch.uint8, device='cuda')
out = torch.empty(N, H, W, Co, device='cuda'); b = torch.zeros(Co, device='cuda')
def bench(x, w, tag):
    def run(): L.check(lib.eld_conv3x3_forward(L.dptr(x), Ci, None, 0, L.dptr(w), L.dptr(b), L.dptr(out), N, H, W, Co, 1, L.dptr(ws), ws.numel(), L.cur_stream()))
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print('%-34s %.3f ms  %.1f TF/s' % (tag, ms, 2.0 * N * H * W * Co * Ci * 9 / ms / 1e9))
xr = torch.randn(N, H, W, Ci, device='cuda'); wr = torch.randn(Co, Ci, 3, 3, device='cuda') * 0.02
bench(xr, wr, 'random x, random w')
bench(torch.zeros_like(xr), torch.zeros_like(wr), 'zero x, zero w')
bench(xr, torch.zeros_like(wr), 'random x, zero w')
bench(xr.round(), wr, 'integer-valued x (1 piece), random w')
xb = xr.bfloat16().float(); wb = wr.bfloat16().float()
bench(xb, wb, 'bf16-representable x and w')

# weight gradient of the same layer
dw = torch.empty(Co, Ci, 3, 3, device='cuda'); db = torch.empty(Co, device='cuda')
def benchw(g, x, tag):
    def run(): L.check(lib.eld_conv3x3_backward_weight(L.dptr(g), L.dptr(x), Ci, None, 0, L.dptr(dw), L.dptr(db), N, H, W, Co, L.dptr(ws), ws.numel(), L.cur_stream()))
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print('wgrad %-28s %.3f ms  %.1f TF/s' % (tag, ms, 2.0 * N * H * W * Co * Ci * 9 / ms / 1e9))
gr = torch.randn(N, H, W, Co, device='cuda')
benchw(gr, xr, 'random g, random x')
benchw(torch.zeros_like(gr), torch.zeros_like(xr), 'zero g, zero x')
