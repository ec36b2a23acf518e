"""fp32 vs bf16 timing at full frame size (dev tool): forward, and forward+backward."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eld_amd.unet import UNetSeeInDark
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1
net = UNetSeeInDark(4, 4).cuda()
x = torch.rand(N, 4, 1424, 2128, device='cuda')
dout = torch.ones(N, 4, 1424, 2128, device='cuda') / x.numel()
grads = torch.empty(net._offsets[-1], device='cuda')
def ev(fn, reps=4):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for prec in ('fp32', 'bf16'):
    bf = prec == 'bf16'
    st = {}
    def fwd(): st['k'] = net._engine_forward(x, save=True, bf16=bf)[1]
    def bwd(): net._engine_backward(dout, st['k'], tuple(x.shape), grads=grads)
    tf = ev(fwd); tb = ev(bwd)
    print('%s N=%d: fwd %.3f ms (%.0f TF/s)  bwd %.3f ms  fwd+bwd %.3f ms = %.0f TF/s  %.0f MPix/s' % (
        prec, N, tf, 1118.63 * N / tf, tb, tf + tb, 3349.0 * N / (tf + tb), N * 12.121 / (tf + tb) * 1e3))
