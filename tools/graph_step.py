#!/usr/bin/env python3
"""U-Net forward + backward + Adam at the reference's training shape (one 4x512x512 patch), eager launches vs one hipGraph replay
(torch.cuda.CUDAGraph): the C ABI allocates nothing and never synchronises, so the chain is capturable as it is.  Dev tool."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                # noqa: E402
from eld_amd import _lib as L                    # noqa: E402
from eld_amd.unet import UNetSeeInDark           # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    H = W = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    net = UNetSeeInDark(4, 4).cuda()
    x = torch.rand(N, 4, H, W, device='cuda')
    dout = torch.randn(N, 4, H, W, device='cuda') / x.numel()
    n = net._offsets[-1]
    grads, m, v = torch.empty(n, device='cuda'), torch.zeros(n, device='cuda'), torch.zeros(n, device='cuda')
    lib = L.lib()

    def step():
        out, key, _ = net._engine_forward(x, save=True)
        net._engine_backward(dout, key, tuple(x.shape), grads=grads)
        L.check(lib.eld_adam_step(L.dptr(net.flat_params), L.dptr(grads), L.dptr(m), L.dptr(v), n, 1e-4, 0.9, 0.999, 1e-8, 0.0, 1, 1.0, L.cur_stream()), 'adam')
        return out

    def timed(fn, reps=200):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    t_eager = timed(step)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step()
    t_graph = timed(graph.replay)
    print('%dx4x%dx%d fwd+bwd+Adam: eager %.3f ms, hipGraph replay %.3f ms (%.1f %%)' % (N, H, W, t_eager, t_graph, 100.0 * (t_graph / t_eager - 1.0)))


if __name__ == '__main__':
    main()
