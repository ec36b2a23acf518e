"""dev: where the time of one fp32 transposed-conv launch goes -- the single-layer entry points timed under the ablation switches of the dev library
(tools/build_dev.sh -> tools/probe/libeld_dev.so; ELD_CONV_DBG is read once per process, so this script re-runs itself per switch value).
usage: ELD_AMD_LIB=tools/probe/libeld_dev.so python tools/convt_ablate.py [N=8]
bits (conv_x3_gemm_kernel): 1 no epilogue, 2 no fragment reads / MFMAs, 4 no staging loads, 16 no cut / LDS stores"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LAYERS = [('upv6 f 512->256', 4, 512, 256, 'f'), ('upv7 f 256->128', 3, 256, 128, 'f'), ('upv8 f 128->64', 2, 128, 64, 'f'), ('upv9 f 64->32', 1, 64, 32, 'f'),
          ('upv6 b', 4, 512, 256, 'b'), ('upv8 b', 2, 128, 64, 'b'), ('upv9 b', 1, 64, 32, 'b')]
DBGS = [int(v) for v in os.environ.get('ABLATE_DBGS', '0,1,2,3,4,16,18,20,22').split(',')]


def child():
    sys.path.insert(0, ROOT)
    import torch
    import eld_amd
    from eld_amd import _lib as L
    lib = eld_amd.load_library()
    N = int(sys.argv[2])
    out_line = []
    for name, l, Ci, Co, d in LAYERS:
        H, W = 1424 >> l, 2128 >> l                       # input resolution of the transposed conv
        x = torch.randn(N, H, W, Ci, device='cuda')
        w = torch.randn(Ci, Co, 2, 2, device='cuda') * 0.05; b = torch.randn(Co, device='cuda')
        out = torch.empty(N, 2 * H, 2 * W, Co, device='cuda'); g = torch.randn(N, 2 * H, 2 * W, Co, device='cuda'); din = torch.empty(N, H, W, Ci, device='cuda')
        ws = torch.empty(lib.eld_layer_workspace_bytes(N, 2 * H, 2 * W, Ci, Co), dtype=torch.uint8, device='cuda')
        if d == 'f':
            def fn():
                L.check(lib.eld_convt2x2_forward(L.dptr(x), L.dptr(w), L.dptr(b), L.dptr(out), N, H, W, Ci, Co, L.dptr(ws), ws.numel(), L.cur_stream()))
        else:
            def fn():
                L.check(lib.eld_convt2x2_backward_data(L.dptr(g), L.dptr(w), L.dptr(x), L.dptr(din), N, H, W, Ci, Co, L.dptr(ws), ws.numel(), L.cur_stream()))
        fn(); fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record(); torch.cuda.synchronize()
        out_line.append('%.3f' % (e0.elapsed_time(e1) / 5))
        del x, out, g, din, ws
    print(' '.join(out_line))


def main():
    N = sys.argv[1] if len(sys.argv) > 1 else '8'
    print('%-8s' % 'dbg' + ''.join(' | %-16s' % n for n, *_ in LAYERS))
    for rnd in range(2):
        for dbg in DBGS:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), '--child', N], env=dict(os.environ, ELD_CONV_DBG=str(dbg)), capture_output=True, text=True)
            vals = r.stdout.strip().split() if r.returncode == 0 else ['ERR'] * len(LAYERS)
            if r.returncode != 0:
                sys.stderr.write(r.stderr[-500:])
            print('%-8s' % dbg + ''.join(' | %-16s' % v for v in vals), flush=True)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == '--child':
        child()
    else:
        main()
