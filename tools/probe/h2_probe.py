"""Dev probe: accuracy and speed of the experimental two-piece fp16 product scheme (ELD_FP32_CONV=h2) vs fp32 MFMA / bf16x3."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np, eld_amd
import torch.nn.functional as F
from eld_amd import _lib as L
lib = eld_amd.load_library()
def nhwc(t): return t.permute(0, 2, 3, 1).contiguous()
def nchw(t): return t.permute(0, 3, 1, 2).contiguous()
N, H, W, Cin, Cout = 1, 24, 64, 512, 64
g = torch.Generator().manual_seed(7)
x = torch.randn(N, Cin, H, W, generator=g); w = torch.randn(Cout, Cin, 3, 3, generator=g) / np.sqrt(9 * Cin); b = torch.zeros(Cout)
ref = F.conv2d(x.double(), w.double(), None, padding=1)
xd, wd, bd = nhwc(x).cuda(), w.cuda(), b.cuda()
ws = torch.empty(lib.eld_layer_workspace_bytes(N, H, W, Cin, Cout), dtype=torch.uint8, device='cuda')
for a in (0, 1, 2):
    lib.eld_conv_fp32_algo(a)
    out = torch.empty(N, H, W, Cout, device='cuda')
    L.check(lib.eld_conv3x3_forward(L.dptr(xd), Cin, None, 0, L.dptr(wd), L.dptr(bd), L.dptr(out), N, H, W, Cout, 0, L.dptr(ws), ws.numel(), L.cur_stream()))
    torch.cuda.synchronize()
    e = (nchw(out).cpu().double() - ref).abs()
    print('algo %d: max err %.3e rms err %.3e (output rms %.3f)' % (a, float(e.max()), float((e ** 2).mean().sqrt()), float(ref.pow(2).mean().sqrt())))
for (N, H, W, Ci, Co) in ((8, 178, 266, 256, 256), (8, 356, 532, 256, 128), (8, 712, 1064, 64, 64), (8, 1424, 2128, 32, 32)):
    x = torch.randn(N, H, W, Ci, device='cuda'); w = torch.randn(Co, Ci, 3, 3, device='cuda') * 0.02; b = torch.zeros(Co, device='cuda')
    out = torch.empty(N, H, W, Co, device='cuda')
    ws = torch.empty(lib.eld_layer_workspace_bytes(N, H, W, Ci, Co), dtype=torch.uint8, device='cuda')
    for a in (0, 1, 2):
        lib.eld_conv_fp32_algo(a)
        def run(): L.check(lib.eld_conv3x3_forward(L.dptr(x), Ci, None, 0, L.dptr(w), L.dptr(b), L.dptr(out), N, H, W, Co, 1, L.dptr(ws), ws.numel(), L.cur_stream()))
        for _ in range(2): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print('  %dx%dx%d %d->%d algo %d: %.3f ms %.1f TF/s' % (N, H, W, Ci, Co, a, ms, 2.0 * N * H * W * Co * Ci * 9 / ms / 1e9))
lib.eld_conv_fp32_algo(1)
