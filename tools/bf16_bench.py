"""Forward (inference) timing fp32 vs bf16 at full frame size (dev tool)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eld_amd.unet import UNetSeeInDark
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1
net = UNetSeeInDark(4, 4).cuda()
x = torch.rand(N, 4, 1424, 2128, device='cuda')
for prec in ('fp32', 'bf16'):
    net.inference_precision = prec
    with torch.no_grad():
        net(x); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): net(x)
        e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print('%s forward N=%d: %.3f ms  %.1f TFLOP/s  %.0f MPix/s' % (prec, N, ms, 1118.63e9 * N / ms / 1e9, N * 12.121 / ms * 1e3))
