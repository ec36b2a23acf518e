"""dev: where the time of one fp32 3x3 launch goes -- the single-layer entry point timed under the ablation switches of the dev library
(tools/build_dev.sh -> tools/probe/libeld_dev.so; ELD_CONV_DBG is read once per process, so this script re-runs itself per switch value).
usage: ELD_AMD_LIB=tools/probe/libeld_dev.so python tools/conv_ablate.py [N=8]
bits: 1 no epilogue, 2 no fragment reads / MFMAs, 4 no staging loads, 8 no slab stores / DMA, 16 no halo cut + LDS stores, 64 one workgroup per CU"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LAYERS = [  # name, level, C0, C1, Cout, direction
    ('conv9_2 f 32->32 L0', 0, 32, 0, 32, 'f'), ('conv9_1 f 64->32 L0', 0, 32, 32, 32, 'f'), ('conv9_1 b 32->64 L0', 0, 32, 32, 32, 'b'),
    ('conv2_2 f 64->64 L1', 1, 64, 0, 64, 'f'), ('conv8_1 f 128->64 L1', 1, 64, 64, 64, 'f'), ('conv7_2 f 128->128 L2', 2, 128, 0, 128, 'f'),
]
DBGS = [int(v) for v in os.environ.get('ABLATE_DBGS', '0,64,1,2,3,4,16,8,5,18').split(',')]


def child():
    sys.path.insert(0, ROOT)
    import torch
    import eld_amd
    from eld_amd import _lib as L
    lib = eld_amd.load_library()
    N = int(sys.argv[2])
    out_line = []
    for name, l, C0, C1, Co, d in LAYERS:
        H, W = 1424 >> l, 2128 >> l
        Cin = C0 + C1
        x0 = torch.randn(N, H, W, C0, device='cuda'); x1 = torch.randn(N, H, W, C1, device='cuda') if C1 else None
        w = torch.randn(Co, Cin, 3, 3, device='cuda') * 0.05; b = torch.randn(Co, device='cuda')
        out = torch.empty(N, H, W, Co, device='cuda'); g = torch.randn(N, H, W, Co, device='cuda')
        d0 = torch.empty(N, H, W, C0, device='cuda'); d1 = torch.empty(N, H, W, C1, device='cuda') if C1 else None
        ws = torch.empty(lib.eld_layer_workspace_bytes(N, H, W, Cin, Co), dtype=torch.uint8, device='cuda')
        if d == 'f':
            def fn():
                L.check(lib.eld_conv3x3_forward(L.dptr(x0), C0, L.dptr(x1), C1, L.dptr(w), L.dptr(b), L.dptr(out), N, H, W, Co, 1, L.dptr(ws), ws.numel(), L.cur_stream()))
        else:
            def fn():
                L.check(lib.eld_conv3x3_backward_data(L.dptr(g), L.dptr(w), L.dptr(d0), L.dptr(d1), C0, None, None, N, H, W, Cin, Co, L.dptr(ws), ws.numel(), L.cur_stream()))
        fn(); fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record(); torch.cuda.synchronize()
        out_line.append('%.3f' % (e0.elapsed_time(e1) / 5))
        del x0, x1, out, g, d0, d1, ws
    print(' '.join(out_line))


def main():
    N = sys.argv[1] if len(sys.argv) > 1 else '8'
    print('%-8s' % 'dbg' + ''.join(' | %-22s' % n for n, *_ in LAYERS))
    for rnd in range(2):
        for dbg in DBGS:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), '--child', N], env=dict(os.environ, ELD_CONV_DBG=str(dbg)), capture_output=True, text=True)
            vals = r.stdout.strip().split() if r.returncode == 0 else ['ERR'] * len(LAYERS)
            if r.returncode != 0:
                sys.stderr.write(r.stderr[-500:])
            print('%-8s' % dbg + ''.join(' | %-22s' % v for v in vals), flush=True)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == '--child':
        child()
    else:
        main()
