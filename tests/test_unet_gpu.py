"""GPU parity tests of the U-Net kernels and of the whole forward/backward, through the C ABI.
Checker = torch CPU float32/float64 (oracle/unet_ref.py, pinned to the reference's golden output).
Tolerance: fp32 within 1e-5 (BASELINE.json north_star) on O(1) data; where the contraction is long
(K up to 4608) the bound is scaled by the fp64-measured magnitude sum|a*b|*2^-23*sqrt(K)-class term."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')
import torch.nn.functional as F      # noqa: E402

from oracle import unet_ref as U     # noqa: E402  (checker only)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib(eld_lib):
    assert torch.cuda.is_available()
    return eld_lib


@pytest.fixture(params=[0, 1, 2], ids=['fp32mfma', 'bf16x3', 'fp16x2'])
def algo(request, lib):
    """Run the test under both fp32 conv product schemes (include/eld_amd.h eld_conv_fp32_algo); same tolerance for both."""
    prev = lib.eld_conv_fp32_algo(request.param)
    yield request.param
    lib.eld_conv_fp32_algo(prev)


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


def ws_for(lib, N, H, W, Cin, Cout):
    n = lib.eld_layer_workspace_bytes(N, H, W, Cin, Cout)
    assert n > 0
    return torch.empty(n, dtype=torch.uint8, device='cuda')


def close(got, ref64, tol=1e-5, scale=None):
    """|got - ref| <= tol * (1 + scale) elementwise; scale = magnitude of the accumulated terms (fp64)."""
    got = got.detach().cpu().double()
    err = (got - ref64).abs()
    bound = tol * (1.0 + (scale if scale is not None else ref64.abs()))
    bad = (err > bound)
    assert not bad.any(), 'max err %.3e (bound %.3e) at %d elements' % (float(err.max()), float(bound.max() if torch.is_tensor(bound) else bound), int(bad.sum()))


CASES = [  # N, H, W, C0, C1, Cout
    (2, 20, 37, 32, 0, 32), (1, 9, 133, 64, 0, 64), (1, 16, 40, 32, 32, 32), (2, 12, 33, 64, 64, 64),
    (1, 8, 34, 16, 0, 32), (1, 7, 31, 128, 0, 256), (1, 5, 17, 512, 0, 64),
    (1, 16, 96, 32, 0, 32), (3, 16, 32, 32, 0, 32), (1, 48, 160, 32, 32, 32),      # odd tile counts of the 32-output-channel kernel (3, 3, 15 tiles)
]


@pytest.mark.parametrize('N,H,W,C0,C1,Cout', CASES)
@pytest.mark.parametrize('act', [1, 0])
def test_conv3x3_forward(lib, N, H, W, C0, C1, Cout, act, algo):
    from eld_amd import _lib as L
    g = torch.Generator().manual_seed(H * W + C0 + Cout)
    x = torch.randn(N, C0 + C1, H, W, generator=g)
    w = torch.randn(Cout, C0 + C1, 3, 3, generator=g) / np.sqrt(9 * (C0 + C1))
    b = torch.randn(Cout, generator=g)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    mag = F.conv2d(x.double().abs(), w.double().abs(), b.double().abs(), padding=1)
    if act:
        ref = torch.max(0.2 * ref, ref)
    x0 = nhwc(x[:, :C0]).cuda()
    x1 = nhwc(x[:, C0:]).cuda() if C1 else None
    out = torch.empty(N, H, W, Cout, device='cuda')
    ws = ws_for(lib, N, H, W, C0 + C1, Cout)
    wd_, bd_ = w.cuda(), b.cuda()           # keep device operands alive across the async call
    L.check(lib.eld_conv3x3_forward(L.dptr(x0), C0, L.dptr(x1), C1, L.dptr(wd_), L.dptr(bd_), L.dptr(out), N, H, W, Cout, act,
                                    L.dptr(ws), ws.numel(), L.cur_stream()))
    close(nchw(out), ref, tol=2e-6, scale=mag)
    ref32 = F.conv2d(x, w, b, padding=1)
    if act:
        ref32 = torch.max(0.2 * ref32, ref32)
    assert (nchw(out).cpu() - ref32).abs().max() < 1e-5 * (1 + float(mag.max()))


@pytest.mark.parametrize('N,H,W,Cin,Cout,split', [(2, 20, 37, 32, 32, 32), (1, 16, 40, 64, 32, 32), (1, 9, 133, 128, 64, 64),
                                                     (1, 6, 18, 256, 512, 256), (1, 8, 32, 32, 64, 32)])
def test_conv3x3_backward_data(lib, N, H, W, Cin, Cout, split, algo):
    from eld_amd import _lib as L
    g = torch.Generator().manual_seed(7 + Cin)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / np.sqrt(9 * Cout)
    gy = torch.randn(N, Cout, H, W, generator=g)
    act = torch.randn(N, Cin, H, W, generator=g)
    act[:, :, ::3, ::5] = 0.0                                   # exercise the tie slope 0.6
    ref = torch.nn.grad.conv2d_input((N, Cin, H, W), w.double(), gy.double(), padding=1)
    mag = torch.nn.grad.conv2d_input((N, Cin, H, W), w.double().abs(), gy.double().abs(), padding=1)
    slope = torch.where(act > 0, 1.0, torch.where(act < 0, 0.2, 0.6)).double()
    d0 = torch.empty(N, H, W, split, device='cuda')
    d1 = torch.empty(N, H, W, Cin - split, device='cuda') if split < Cin else None
    a0 = nhwc(act[:, :split]).cuda()
    ws = ws_for(lib, N, H, W, Cin, Cout)
    gyd_, wd_ = nhwc(gy).cuda(), w.cuda()
    L.check(lib.eld_conv3x3_backward_data(L.dptr(gyd_), L.dptr(wd_), L.dptr(d0), L.dptr(d1), split, L.dptr(a0), None,
                                          N, H, W, Cin, Cout, L.dptr(ws), ws.numel(), L.cur_stream()))
    close(nchw(d0), ref[:, :split] * slope[:, :split], tol=2e-6, scale=mag[:, :split])
    if d1 is not None:
        close(nchw(d1), ref[:, split:], tol=2e-6, scale=mag[:, split:])      # second half: no activation in front


@pytest.mark.parametrize('N,H,W,C0,C1,Cout', CASES + [(2, 64, 96, 32, 0, 32), (1, 21, 70, 16, 0, 32)])
def test_conv3x3_backward_weight(lib, N, H, W, C0, C1, Cout, algo):
    from eld_amd import _lib as L
    g = torch.Generator().manual_seed(11 + H)
    x = torch.randn(N, C0 + C1, H, W, generator=g)
    gy = torch.randn(N, Cout, H, W, generator=g)
    ref = torch.nn.grad.conv2d_weight(x.double(), (Cout, C0 + C1, 3, 3), gy.double(), padding=1)
    mag = torch.nn.grad.conv2d_weight(x.double().abs(), (Cout, C0 + C1, 3, 3), gy.double().abs(), padding=1)
    x0 = nhwc(x[:, :C0]).cuda()
    x1 = nhwc(x[:, C0:]).cuda() if C1 else None
    dw = torch.full((Cout, C0 + C1, 3, 3), float('nan'), device='cuda')
    db = torch.full((Cout,), float('nan'), device='cuda')
    ws = ws_for(lib, N, H, W, C0 + C1, Cout)
    gyd_ = nhwc(gy).cuda()
    for _ in range(2):      # run twice: bit-stable (fixed reduction order, no atomics)
        L.check(lib.eld_conv3x3_backward_weight(L.dptr(gyd_), L.dptr(x0), C0, L.dptr(x1), C1, L.dptr(dw), L.dptr(db), N, H, W, Cout,
                                                L.dptr(ws), ws.numel(), L.cur_stream()))
        cur = (dw.clone(), db.clone())
        if _:
            assert torch.equal(cur[0], prev[0]) and torch.equal(cur[1], prev[1])
        prev = cur
    close(dw, ref, tol=2e-6, scale=mag)
    close(db, gy.double().sum(dim=(0, 2, 3)), tol=2e-6, scale=gy.double().abs().sum(dim=(0, 2, 3)))


@pytest.mark.parametrize('N,H,W,C,Cout', [(3, 45, 77, 64, 64), (5, 21, 40, 128, 128), (4, 34, 50, 64, 128), (2, 16, 64, 64, 64)])
def test_batch_strip_tiling_is_image_independent(lib, N, H, W, C, Cout):
    """Round 4: the 3x3 kernels tile the batch as ONE strip of rows (csrc/conv.h vrow_*: image i's row y at strip row i * (H + S) + y), so a
    tile may hold the last rows of one image and the first rows of the next, and the weight gradient walks tiles across image seams.  Whatever
    the strip / tile shape, every image of a batch must come out exactly as it does alone: forward and backward-data bit for bit (odd and even
    heights: S = 1 / 2; a height that is a multiple of the tile: no seam), and the batch weight gradient == the sum over images (fp32 summation
    order differs: 2e-6 of the accumulated magnitude)."""
    from eld_amd import _lib as L
    g = torch.Generator().manual_seed(N * H + W)
    x = torch.randn(N, H, W, C, generator=g).cuda()
    gy = torch.randn(N, H, W, Cout, generator=g).cuda()
    act = torch.randn(N, H, W, C, generator=g).cuda()
    w = (torch.randn(Cout, C, 3, 3, generator=g) / np.sqrt(9 * C)).cuda()
    b = torch.randn(Cout, generator=g).cuda()

    def fwd(xx):
        n = xx.shape[0]
        out = torch.empty(n, H, W, Cout, device='cuda')
        ws = ws_for(lib, n, H, W, C, Cout)
        L.check(lib.eld_conv3x3_forward(L.dptr(xx), C, None, 0, L.dptr(w), L.dptr(b), L.dptr(out), n, H, W, Cout, 1, L.dptr(ws), ws.numel(), L.cur_stream()))
        torch.cuda.synchronize()
        return out

    def bwd(gg, aa):
        n = gg.shape[0]
        d = torch.empty(n, H, W, C, device='cuda')
        ws = ws_for(lib, n, H, W, C, Cout)
        L.check(lib.eld_conv3x3_backward_data(L.dptr(gg), L.dptr(w), L.dptr(d), None, C, L.dptr(aa), None, n, H, W, C, Cout, L.dptr(ws), ws.numel(), L.cur_stream()))
        torch.cuda.synchronize()
        return d

    def wgrad(gg, xx):
        n = gg.shape[0]
        dw = torch.empty(Cout, C, 3, 3, device='cuda'); db = torch.empty(Cout, device='cuda')
        ws = ws_for(lib, n, H, W, C, Cout)
        L.check(lib.eld_conv3x3_backward_weight(L.dptr(gg), L.dptr(xx), C, None, 0, L.dptr(dw), L.dptr(db), n, H, W, Cout, L.dptr(ws), ws.numel(), L.cur_stream()))
        torch.cuda.synchronize()
        return dw.double().cpu(), db.double().cpu()
    out, din = fwd(x), bwd(gy, act)
    dw, db = wgrad(gy, x)
    dw_sum, db_sum = torch.zeros_like(dw), torch.zeros_like(db)
    for i in range(N):
        xi, gi, ai = x[i:i + 1].contiguous(), gy[i:i + 1].contiguous(), act[i:i + 1].contiguous()
        assert torch.equal(fwd(xi)[0], out[i]), 'forward: image %d of the batch differs from the image alone' % i
        assert torch.equal(bwd(gi, ai)[0], din[i]), 'backward-data: image %d of the batch differs from the image alone' % i
        a, c = wgrad(gi, xi)
        dw_sum += a; db_sum += c
    mag = torch.nn.grad.conv2d_weight(nchw(x).double().abs().cpu(), (Cout, C, 3, 3), nchw(gy).double().abs().cpu(), padding=1)
    assert float(((dw - dw_sum).abs() / (1.0 + mag)).max()) <= 2e-6
    assert float((db - db_sum).abs().max()) <= 2e-6 * (1.0 + float(gy.double().abs().sum(dim=(0, 1, 2)).max()))


@pytest.mark.parametrize('N,H,W,Cin,Cout', [(2, 10, 19, 64, 32), (1, 5, 33, 512, 256), (1, 8, 8, 128, 64)])
def test_conv_transpose2x2(lib, N, H, W, Cin, Cout, algo):
    from eld_amd import _lib as L
    g = torch.Generator().manual_seed(3 + Cin)
    x = torch.randn(N, Cin, H, W, generator=g)
    x[:, :, ::2, ::3] = 0.0
    w = torch.randn(Cin, Cout, 2, 2, generator=g) / np.sqrt(Cin)
    b = torch.randn(Cout, generator=g)
    gy = torch.randn(N, Cout, 2 * H, 2 * W, generator=g)
    ws = ws_for(lib, N, H, W, Cin, Cout)
    xd = x.double().requires_grad_(True)
    wd = w.double().requires_grad_(True)
    bd = b.double().requires_grad_(True)
    ref = F.conv_transpose2d(xd, wd, bd, stride=2)
    ref.backward(gy.double())
    out = torch.empty(N, 2 * H, 2 * W, Cout, device='cuda')
    xd_, wd_, bd_ = nhwc(x).cuda(), w.cuda(), b.cuda()
    L.check(lib.eld_convt2x2_forward(L.dptr(xd_), L.dptr(wd_), L.dptr(bd_), L.dptr(out), N, H, W, Cin, Cout,
                                     L.dptr(ws), ws.numel(), L.cur_stream()))
    mag = F.conv_transpose2d(x.double().abs(), w.double().abs(), b.double().abs(), stride=2)
    close(nchw(out), ref.detach(), tol=2e-6, scale=mag)
    din = torch.empty(N, H, W, Cin, device='cuda')
    gyd = nhwc(gy).cuda()
    L.check(lib.eld_convt2x2_backward_data(L.dptr(gyd), L.dptr(wd_), L.dptr(xd_), L.dptr(din), N, H, W, Cin, Cout,
                                           L.dptr(ws), ws.numel(), L.cur_stream()))
    slope = torch.where(x > 0, 1.0, torch.where(x < 0, 0.2, 0.6)).double()
    magd = F.conv2d(gy.double().abs(), w.double().abs().permute(0, 1, 2, 3), stride=2)       # |gy| * |w| summed: conv with (Cin,Cout,2,2)
    close(nchw(din), xd.grad * slope, tol=2e-6, scale=magd)
    dw = torch.empty(Cin, Cout, 2, 2, device='cuda')
    db = torch.empty(Cout, device='cuda')
    L.check(lib.eld_convt2x2_backward_weight(L.dptr(xd_), L.dptr(gyd), L.dptr(dw), L.dptr(db), N, H, W, Cin, Cout,
                                             L.dptr(ws), ws.numel(), L.cur_stream()))
    close(dw, wd.grad, tol=2e-6, scale=wd.grad.abs() + float(np.sqrt(N * H * W)) * 3)
    close(db, bd.grad, tol=2e-6, scale=gy.double().abs().sum(dim=(0, 2, 3)))


def test_maxpool(lib):
    from eld_amd import _lib as L
    g = torch.Generator().manual_seed(1)
    N, Ho, Wo, C = 2, 9, 13, 64
    x = torch.randn(N, C, 2 * Ho, 2 * Wo, generator=g).round(decimals=1)        # coarse values -> plenty of ties
    x[0, :, :2, :2] = 1.5                                                        # all-equal window
    x[:, :, 4:6, 4:6] = 0.0                                                      # tie + zero activation
    dp = torch.randn(N, C, Ho, Wo, generator=g)
    skip = torch.randn(N, C, 2 * Ho, 2 * Wo, generator=g)
    out = torch.empty(N, Ho, Wo, C, device='cuda')
    xd_, dpd_, skd_ = nhwc(x).cuda(), nhwc(dp).cuda(), nhwc(skip).cuda()
    L.check(lib.eld_maxpool2x2_forward(L.dptr(xd_), L.dptr(out), N, Ho, Wo, C, L.cur_stream()))
    assert torch.equal(nchw(out).cpu(), F.max_pool2d(x, 2))
    xr = x.clone().requires_grad_(True)
    F.max_pool2d(xr, 2).backward(dp)
    slope = torch.where(x > 0, 1.0, torch.where(x < 0, 0.2, 0.6))
    for sk in (skip, None):
        gout = torch.empty(N, 2 * Ho, 2 * Wo, C, device='cuda')
        L.check(lib.eld_maxpool2x2_backward(L.dptr(xd_), L.dptr(dpd_), L.dptr(skd_) if sk is not None else None,
                                            L.dptr(gout), N, Ho, Wo, C, L.cur_stream()))
        ref = (xr.grad + (sk if sk is not None else 0)) * slope
        assert torch.equal(nchw(gout).cpu(), ref)


def test_l1_and_adam(lib):
    from eld_amd import _lib as L
    g = torch.Generator().manual_seed(2)
    n = 4 * 37 * 53 + 3
    out = torch.randn(n, generator=g)
    tgt = torch.randn(n, generator=g)
    tgt[:10] = out[:10]
    dout = torch.empty(n, device='cuda')
    loss = torch.zeros(1, device='cuda')
    ws = torch.empty(lib.eld_l1_workspace_bytes(), dtype=torch.uint8, device='cuda')
    od_, td_ = out.cuda(), tgt.cuda()
    L.check(lib.eld_l1_loss(L.dptr(od_), L.dptr(td_), L.dptr(dout), L.dptr(loss), L.dptr(ws), n, 1.0, L.cur_stream()))
    o = out.clone().requires_grad_(True)
    lref = F.l1_loss(o, tgt)
    lref.backward()
    assert abs(float(loss) - float(lref)) < 1e-6
    assert torch.equal(dout.cpu(), o.grad)
    # --loss l2 (models/losses.py:34): nn.MSELoss and its gradient
    L.check(lib.eld_mse_loss(L.dptr(od_), L.dptr(td_), L.dptr(dout), L.dptr(loss), L.dptr(ws), n, 1.0, L.cur_stream()))
    o2 = out.clone().requires_grad_(True)
    lref2 = F.mse_loss(o2, tgt)
    lref2.backward()
    assert abs(float(loss) - float(lref2)) < 1e-6 * float(lref2)
    assert float((dout.cpu() - o2.grad).abs().max()) <= 1e-7 * float(o2.grad.abs().max())
    # Adam: 5 steps vs torch.optim.Adam on CPU
    p = torch.randn(10007, generator=g)
    pr = p.clone().requires_grad_(True)
    opt = torch.optim.Adam([pr], lr=1e-4, betas=(0.9, 0.999))
    pd, m, v = p.cuda(), torch.zeros(10007, device='cuda'), torch.zeros(10007, device='cuda')
    for step in range(1, 6):
        gr = torch.randn(10007, generator=g) * 0.01
        pr.grad = gr.clone()
        opt.step()
        grd_ = gr.cuda()
        L.check(lib.eld_adam_step(L.dptr(pd), L.dptr(grd_), L.dptr(m), L.dptr(v), 10007, 1e-4, 0.9, 0.999, 1e-8, 0.0, step, 1.0, L.cur_stream()))
        torch.cuda.synchronize()
    assert (pd.cpu() - pr.detach()).abs().max() < 5e-7      # a few ulp at |p| ~ 1


# ------------------------------------------------------------------------------------------------ whole network
def test_unet_forward_backward_vs_reference_golden(lib, golden_dir, algo):
    """Same seeded init as the reference module, same input: output within 1e-5, L1 gradients of all 46 tensors."""
    from eld_amd.unet import UNetSeeInDark
    d = np.load(os.path.join(golden_dir, 'unet.npz'))
    torch.manual_seed(2018)
    net = UNetSeeInDark(4, 4)
    names = [n for n, _ in net.named_parameters()]
    assert names == [str(n) for n in d['names']]
    if str(d['torch_version']) == torch.__version__:
        wsum = np.array([float(p.detach().double().sum()) for _, p in net.named_parameters()])
        assert np.array_equal(wsum, d['wsum'])
    net = net.cuda()
    x, t = torch.from_numpy(d['x']).cuda(), torch.from_numpy(d['t']).cuda()
    out = net(x)
    assert out.shape == (2, 4, 32, 48)
    assert float((out.detach().cpu() - torch.from_numpy(d['out'])).abs().max()) <= 1e-5
    loss = torch.nn.L1Loss()(out, t)                       # the reference's own criterion (models/losses.py:32)
    loss.backward()
    assert abs(float(loss) - float(d['loss'])) < 1e-6
    gsum = np.array([float(p.grad.double().sum()) for _, p in net.named_parameters()])
    gabs = np.array([float(p.grad.double().abs().sum()) for _, p in net.named_parameters()])
    assert np.allclose(gabs, d['gabs'], rtol=2e-4, atol=1e-7), np.max(np.abs(gabs - d['gabs']) / (d['gabs'] + 1e-12))
    assert np.allclose(gsum, d['gsum'], rtol=0, atol=2e-4 * np.maximum(d['gabs'], 1e-6).max())
    for k in ('conv1_1.weight', 'conv1_1.bias', 'conv10_1.weight', 'conv10_1.bias', 'upv9.bias', 'conv9_2.bias'):
        ref = d['grad_' + k.replace('.', '__')]
        got = dict(net.named_parameters())[k].grad.cpu().numpy()
        assert np.max(np.abs(got - ref)) <= 1e-5 * (1 + np.abs(ref).max()), k


@pytest.mark.parametrize('shape', [(1, 4, 16, 16), (3, 4, 48, 80), (1, 4, 64, 144)])
def test_unet_all_gradients_vs_oracle(lib, shape, algo):
    from eld_amd.unet import UNetSeeInDark
    torch.manual_seed(7)
    net = UNetSeeInDark(4, 4)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    g = torch.Generator().manual_seed(3)
    x, t = torch.rand(*shape, generator=g), torch.rand(*shape, generator=g)
    out_ref, loss_ref, grads = U.loss_and_grads({k: v.double() for k, v in sd.items()}, x.double(), t.double())
    net = net.cuda()
    out = net(x.cuda())
    loss = torch.nn.functional.l1_loss(out, t.cuda())
    loss.backward()
    assert float((out.detach().cpu().double() - out_ref).abs().max()) <= 1e-5
    assert abs(float(loss) - loss_ref) < 1e-6
    for n, p in net.named_parameters():
        ref = grads[n]
        err = float((p.grad.cpu().double() - ref).abs().max())
        assert err <= 1e-5 * (1 + float(ref.abs().max())), (n, err, float(ref.abs().max()))      # plain north_star bound; measured: profiles/r02_parity.md
        # ... which the deep layers' tiny gradients meet even when they are zero: a relative bound against the float64 oracle too
        assert err <= 2e-4 * float(ref.abs().max()), (n, err, float(ref.abs().max()))
    # eval-mode forward (no autograd) gives the same bits; state_dict round-trips through the reference key set
    with torch.no_grad():
        assert torch.equal(net(x.cuda()), out.detach())
    net2 = UNetSeeInDark(4, 4)
    net2.load_state_dict({k: v.cpu() for k, v in net.state_dict().items()})
    net2 = net2.cuda()
    with torch.no_grad():
        assert torch.equal(net2(x.cuda()), out.detach())


def test_unet_rejects_bad_input(lib):
    from eld_amd.unet import UNetSeeInDark
    net = UNetSeeInDark(4, 4).cuda()
    with pytest.raises(RuntimeError):
        net(torch.rand(1, 4, 30, 32, device='cuda'))
    with pytest.raises(RuntimeError):
        net(torch.rand(1, 4, 32, 32))


@pytest.mark.parametrize('shape', [(1, 4, 32, 48), (2, 4, 64, 144), (1, 4, 176, 272)])
def test_unet_bf16_inference_vs_fp32(lib, shape):
    """BASELINE config 3 precision: bf16 activations/weights with fp32 accumulation.  Judged as SURVEY.md App. E-4 says:
    output PSNR >= 60 dB against the fp32 engine (255-scaled, like util/index.py:79), and against the fp64 oracle."""
    from eld_amd.unet import UNetSeeInDark
    torch.manual_seed(11)
    net = UNetSeeInDark(4, 4).cuda()
    sd = {k: v.detach().cpu().double() for k, v in net.state_dict().items()}
    g = torch.Generator().manual_seed(1)
    x = torch.rand(*shape, generator=g)
    with torch.no_grad():
        ref32 = net(x.cuda())
        net.inference_precision = 'bf16'
        out = net(x.cuda())
        out2 = net(x.cuda())
        net.inference_precision = 'fp32'
        again32 = net(x.cuda())
    assert torch.equal(out, out2) and torch.equal(ref32, again32)          # deterministic; switching precision leaves fp32 untouched
    assert not torch.equal(out, ref32)
    def psnr(a, b):
        mse = torch.mean((a.double() * 255 - b.double() * 255) ** 2)
        return float(10 * torch.log10(255.0 ** 2 / mse))
    assert psnr(out, ref32) >= 60.0, psnr(out, ref32)
    with torch.no_grad():
        ref64 = U.unet_forward(sd, x.double())
    assert psnr(out.cpu(), ref64) >= 60.0
    assert float((out.cpu().double() - ref64).abs().max()) < 2e-2


@pytest.mark.parametrize('shape', [(2, 4, 32, 48), (1, 4, 64, 144)])
def test_unet_bf16_training_gradients(lib, shape):
    """bf16 forward/backward (fp32 parameter gradients): every gradient tensor within bf16 round-off of the fp64 oracle --
    relative L2 error <= 8 % and cosine >= 0.997 per tensor (the deepest layers of a tiny image see 2^-8-relative
    activation noise through ~20 layers), <= 3 % for the median tensor."""
    from eld_amd.unet import UNetSeeInDark
    torch.manual_seed(7)
    net = UNetSeeInDark(4, 4)
    sd = {k: v.detach().clone().double() for k, v in net.state_dict().items()}
    g = torch.Generator().manual_seed(3)
    x, t = torch.rand(*shape, generator=g), torch.rand(*shape, generator=g)
    out_ref, loss_ref, grads = U.loss_and_grads(sd, x.double(), t.double())
    net = net.cuda()
    net.train_precision = 'bf16'
    out = net(x.cuda())
    loss = torch.nn.functional.l1_loss(out, t.cuda())
    loss.backward()
    assert abs(float(loss) - loss_ref) < 2e-3
    worst, rels = 0.0, []
    for n, p in net.named_parameters():
        ref = grads[n].reshape(-1)
        got = p.grad.cpu().double().reshape(-1)
        rel = float((got - ref).norm() / (ref.norm() + 1e-30))
        cos = float(torch.dot(got, ref) / (got.norm() * ref.norm() + 1e-30))
        worst = max(worst, rel)
        rels.append(rel)
        assert rel <= 0.08 and cos >= 0.997, (n, rel, cos)
    assert sorted(rels)[len(rels) // 2] <= 0.03
    # fp32 path untouched by the switch
    net.train_precision = 'fp32'
    net.zero_grad()
    out32 = net(x.cuda())
    assert float((out32.detach().cpu().double() - out_ref).abs().max()) <= 1e-5


def test_bf16x3_is_as_accurate_as_fp32_mfma(lib):
    """Long contraction (K = 9*512), O(1) data: the error of the split-bf16 scheme against fp64 must not exceed the
    fp32-MFMA kernel's by more than a rounding's worth (csrc/conv_x3.hip header)."""
    from eld_amd import _lib as L
    N, H, W, Cin, Cout = 1, 24, 64, 512, 64
    g = torch.Generator().manual_seed(7)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / np.sqrt(9 * Cin)
    b = torch.zeros(Cout)
    ref = F.conv2d(x.double(), w.double(), None, padding=1)
    xd, wd, bd = nhwc(x).cuda(), w.cuda(), b.cuda()
    ws = ws_for(lib, N, H, W, Cin, Cout)
    errs = []
    prev = lib.eld_conv_fp32_algo(-1)
    try:
        for a in (0, 1, 2):
            lib.eld_conv_fp32_algo(a)
            out = torch.empty(N, H, W, Cout, device='cuda')
            L.check(lib.eld_conv3x3_forward(L.dptr(xd), Cin, None, 0, L.dptr(wd), L.dptr(bd), L.dptr(out), N, H, W, Cout, 0,
                                            L.dptr(ws), ws.numel(), L.cur_stream()))
            torch.cuda.synchronize()
            e = (nchw(out).cpu().double() - ref).abs()
            errs.append((float(e.max()), float((e ** 2).mean().sqrt())))
    finally:
        lib.eld_conv_fp32_algo(prev)
    (m0, r0), (m1, r1), (m2, r2) = errs
    assert m1 < 3e-5 and r1 < 2e-6, errs          # O(1) outputs up to |4|, K = 4608
    assert r1 <= 1.5 * r0 + 1e-8 and m1 <= 2.0 * m0 + 1e-8, errs
    # two fp16 pieces (22-bit products, opt-in): fewer accumulation roundings -- no worse than the fp32 MFMA either
    assert r2 <= 1.5 * r0 + 1e-8 and m2 <= 2.0 * m0 + 1e-8, errs


def test_unet_bf16_dma_kernels_full_frame(lib):
    """BASELINE.json configs[2] at frame size (1 x 4 x 1424 x 2128): here the bf16 3x3 layers run on conv_bfd_kernel (both operands by LDS-DMA;
    small problems stay on conv_igemm_kernel<bf16>).  Checked against independently tested paths:
      * inference vs the fp32 engine: PSNR >= 60 dB;
      * crop consistency ACROSS kernels: a 512 x 512 crop runs on the register-staged kernel; away from the crop border its output matches the
        full-frame output of the DMA kernel to PSNR >= 60 dB (same arithmetic, different accumulation order and tile geometry);
      * training gradients (DMA backward-data, fused pool epilogue, re-blocked bf16 weight gradient) vs the fp32 engine: cosine >= 0.995."""
    from eld_amd.unet import UNetSeeInDark
    torch.manual_seed(13)
    net = UNetSeeInDark(4, 4).cuda()
    g = torch.Generator(device='cuda').manual_seed(17)
    H, W = 1424, 2128
    x = torch.rand(1, 4, H, W, device='cuda', generator=g)

    def psnr(a, b):
        mse = torch.mean((a.double() * 255 - b.double() * 255) ** 2)
        return float(10 * torch.log10(255.0 ** 2 / mse))
    with torch.no_grad():
        ref32 = net(x)
        net.inference_precision = 'bf16'
        full = net(x)
        y0, x0 = 448, 800
        crop = net(x[:, :, y0:y0 + 512, x0:x0 + 512].contiguous())
        net.inference_precision = 'fp32'
    assert psnr(full, ref32) >= 60.0, psnr(full, ref32)
    m = 200
    a, b = full[:, :, y0 + m:y0 + 512 - m, x0 + m:x0 + 512 - m], crop[:, :, m:512 - m, m:512 - m]
    assert psnr(a, b) >= 60.0, psnr(a, b)
    t = torch.rand(1, 4, H, W, device='cuda', generator=g)
    grads = {}
    for prec in ('fp32', 'bf16'):
        net.train_precision = prec
        net.zero_grad()
        loss = torch.nn.functional.l1_loss(net(x), t)
        loss.backward()
        grads[prec] = {n: p.grad.detach().double().reshape(-1).clone() for n, p in net.named_parameters()}
    net.train_precision = 'fp32'
    for n in grads['fp32']:
        r, q = grads['fp32'][n], grads['bf16'][n]
        cos = float(torch.dot(r, q) / (r.norm() * q.norm() + 1e-300))
        assert cos >= 0.995, (n, cos)


@pytest.mark.parametrize('shape', [(2, 4, 272, 560), (1, 4, 512, 512), (2, 4, 1040, 1072), (1, 4, 1424, 2128)])
def test_bf16_specialised_kernels_equal_the_generic_kernel_bit_for_bit(lib, shape):
    """The LDS-DMA kernels -- conv_bfs_kernel (32-output-channel layers: resident weights, 3-deep activation ring, hand-issued activation
    loads), conv_bfd_kernel (the wider layers: slab ring, counted waits across the epilogue's stores) and conv_bfg_kernel (transposed convs)
    with their fused pools and full-line bf16 stores -- accumulate in the same order on the same MFMA as conv_igemm_kernel<bf16>
    (chunk -> ky -> kx -> 16-k block), so routing those launches back to the generic kernel (eld_debug_kernel_mask) must reproduce output
    AND every parameter gradient bit for bit: ragged right / bottom tiles, two images, virtual concats, the slope epilogues of the
    backward-data launches.  A stale LDS tile (a wait that lets a needed DMA piece fly) shows up here as a wrong tile.
    conv_bfw_kernel (64-output-channel layers with K <= 64 at the two larger shapes: resident weights, ring of 16-channel half chunks) sums K
    in the order (chunk, half, ky, kx): it must agree with conv_bfd_kernel<64> on the same launches up to that fp32 summation order -- a
    handful of bf16 roundings that flip -- which a stale tile or a wrong channel block would exceed by orders of magnitude."""
    import json
    from eld_amd.unet import UNetSeeInDark
    torch.manual_seed(5)
    net = UNetSeeInDark(4, 4).cuda()
    net.train_precision = net.inference_precision = 'bf16'
    g = torch.Generator(device='cuda').manual_seed(9)
    x = torch.rand(*shape, device='cuda', generator=g)
    t = torch.rand(*shape, device='cuda', generator=g)
    res = {}
    for mask in (0, 8, 15):
        prev = lib.eld_debug_kernel_mask(mask)
        try:
            net.zero_grad()
            out = net(x)
            torch.nn.functional.l1_loss(out, t).backward()
            res[mask] = (out.detach().clone(), {n: p.grad.detach().clone() for n, p in net.named_parameters()})
        finally:
            lib.eld_debug_kernel_mask(prev)
    assert torch.equal(res[8][0], res[15][0])
    for n in res[8][1]:
        assert torch.equal(res[8][1][n], res[15][1][n]), n

    def rel(a, b):
        return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-300))
    worst = {'out': rel(res[0][0], res[8][0])}
    for n in res[0][1]:
        worst[n] = rel(res[0][1][n], res[8][1][n])
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    with open(os.path.join(ROOT, 'gpurun_out', 'r03_bfw_vs_bfd_%dx%dx%d.json' % (shape[0], shape[2], shape[3])), 'w') as f:
        json.dump(worst, f, indent=1)
    assert worst['out'] <= 3e-3, worst['out']
    for n in res[0][1]:
        assert worst[n] <= 1e-2, (n, worst[n])


@pytest.mark.parametrize('shape', [(3, 4, 272, 560), (1, 4, 1424, 2128)])
def test_bf16_dma_weight_gradient_equals_the_register_staged_kernel(lib, shape):
    """wgrad8d_kernel (round 4: both operand tiles of the 128 x 64 weight-gradient blocks by LDS-DMA, two tile buffers, 16 x 8 tiles on the
    virtual-row strip, bias sums read back from the landed tile) against wgrad8_kernel<bf16> (register-staged, 32 x 8 tiles) on the same
    activations: the two sum the same bf16 products in fp32 over the pixels in a different order, so every parameter gradient must agree to
    1e-4 relative L2 (measured ~1e-6); a tile published before its DMA pieces landed, or a wrong seam row, is off by orders of magnitude.
    Three images of odd-height levels (17 x 35 at the deepest level: strip seams inside tiles) and the full frame."""
    from eld_amd.unet import UNetSeeInDark
    torch.manual_seed(5)
    net = UNetSeeInDark(4, 4).cuda()
    net.train_precision = net.inference_precision = 'bf16'
    g = torch.Generator(device='cuda').manual_seed(9)
    x = torch.rand(*shape, device='cuda', generator=g)
    t = torch.rand(*shape, device='cuda', generator=g)
    res = {}
    for mask in (0, 16):
        prev = lib.eld_debug_kernel_mask(mask)
        try:
            net.zero_grad()
            out = net(x)
            torch.nn.functional.l1_loss(out, t).backward()
            res[mask] = (out.detach().clone(), {n: p.grad.detach().clone() for n, p in net.named_parameters()})
        finally:
            lib.eld_debug_kernel_mask(prev)
    assert torch.equal(res[0][0], res[16][0])                      # the forward does not depend on the weight-gradient kernel
    for n in res[0][1]:
        a, b = res[0][1][n].double(), res[16][1][n].double()
        assert float((a - b).norm() / b.norm().clamp_min(1e-300)) <= 1e-4, n
    # the 128-channel layers are the ones that switched kernels: their gradients must not be bit-identical by accident of routing
    assert any(not torch.equal(res[0][1][n], res[16][1][n]) for n in res[0][1] if n.startswith(('conv3', 'conv4', 'conv5', 'conv6', 'conv7')))


def test_unet_full_frame_properties(lib):
    """BASELINE.json configs[1] size (1 x 4 x 1424 x 2128), where the fp64 oracle is out of reach: size-independent properties.
      * crop consistency: away from the crop border (beyond the receptive field) the output of a 16-aligned 512x512 crop
        equals the full-frame output (the result must not depend on tile position / image size);
      * the two fp32 product schemes (fp32 MFMA, 3-piece bf16 split) agree within the fp32 tolerance on the whole frame;
      * the backward is linear in dout and its bias gradient of the head is the plain sum of dout (a checksum of checksums)."""
    from eld_amd.unet import UNetSeeInDark
    torch.manual_seed(5)
    net = UNetSeeInDark(4, 4).cuda()
    g = torch.Generator(device='cuda').manual_seed(9)
    H, W = 1424, 2128
    x = torch.rand(1, 4, H, W, device='cuda', generator=g)
    prev = lib.eld_conv_fp32_algo(-1)
    try:
        lib.eld_conv_fp32_algo(1)
        with torch.no_grad():
            full = net(x)
            y0, x0 = 448, 800
            crop = net(x[:, :, y0:y0 + 512, x0:x0 + 512].contiguous())
        m = 200                                              # > receptive-field radius of the 5-scale U-Net
        a, b = full[:, :, y0 + m:y0 + 512 - m, x0 + m:x0 + 512 - m], crop[:, :, m:512 - m, m:512 - m]
        assert float((a - b).abs().max()) <= 1e-6
        lib.eld_conv_fp32_algo(0)
        with torch.no_grad():
            full_mfma = net(x)
        assert float((full - full_mfma).abs().max()) <= 1e-5 * (1.0 + float(full.abs().max()))
        lib.eld_conv_fp32_algo(1)
        out, key, _ = net._engine_forward(x, save=True)
        d1 = torch.randn(1, 4, H, W, device='cuda', generator=g) / (4.0 * H * W)
        d2 = torch.randn(1, 4, H, W, device='cuda', generator=g) / (4.0 * H * W)
        g1 = net._engine_backward(d1, key, tuple(x.shape)).clone()
        g2 = net._engine_backward(d2, key, tuple(x.shape)).clone()
        g12 = net._engine_backward((d1 + d2).contiguous(), key, tuple(x.shape)).clone()
        scale = float(torch.maximum(g1.abs(), g2.abs()).max())
        assert float((g12 - (g1 + g2)).abs().max()) <= 2e-5 * scale
        offs = net._offsets
        head_bias = g12[offs[-2]:offs[-1]]                   # conv10_1.bias: d loss / d bias[c] = sum of dout[:, c]
        ref = (d1 + d2).double().sum(dim=(0, 2, 3)).float()
        assert float((head_bias - ref).abs().max()) <= 1e-5 * float(ref.abs().max()) + 1e-9
    finally:
        lib.eld_conv_fp32_algo(prev)


@pytest.mark.parametrize('prec', ['fp32', 'bf16'])
@pytest.mark.parametrize('mse', [False, True], ids=['l1', 'mse'])
def test_fused_head_equals_the_three_kernel_path(lib, prec, mse):
    """eld_unet_forward_loss_ex (last layer + nn.L1Loss / nn.MSELoss + the head's backward as ONE pass over conv9_2's output, ELD_model.py:469-475,
    models/losses.py:30-34) against eld_unet_forward_ex + eld_l1_loss / eld_mse_loss + eld_unet_backward_ex on the same activations: the fused
    kernel forms the 32-channel sum, the loss gradient and the weight-gradient partials in the order of the separate kernels, so the output and
    EVERY gradient must be the same bits; only the loss value (a different order of partial sums) is compared with a tolerance (2e-6 relative)."""
    from eld_amd.unet import UNetSeeInDark
    from eld_amd import _lib as L
    torch.manual_seed(11)
    net = UNetSeeInDark(4, 4).cuda()
    bf16 = prec == 'bf16'
    g = torch.Generator(device='cuda').manual_seed(3)
    shape = (2, 4, 272, 560)
    x = torch.rand(*shape, device='cuda', generator=g)
    t = torch.rand(*shape, device='cuda', generator=g)
    out0, key, _ = net._engine_forward(x, save=True, bf16=bf16)
    out0 = out0.clone()
    dout = torch.empty_like(out0)
    loss0 = torch.zeros(1, device='cuda')
    ws = torch.empty(lib.eld_l1_workspace_bytes(), dtype=torch.uint8, device='cuda')
    fn = lib.eld_mse_loss if mse else lib.eld_l1_loss
    L.check(fn(L.dptr(out0), L.dptr(t), L.dptr(dout), L.dptr(loss0), L.dptr(ws), out0.numel(), 1.0, L.cur_stream()), 'loss')
    g0 = net._engine_backward(dout, key, shape).clone()
    loss1 = torch.zeros(1, device='cuda')
    out1, key, _ = net._engine_forward_loss(x, t, loss1, bf16=bf16, mse=mse)
    g1 = net._engine_backward(None, key, shape).clone()
    torch.cuda.synchronize()
    assert torch.equal(out1, out0)
    assert abs(float(loss1) - float(loss0)) <= 2e-6 * abs(float(loss0))
    assert torch.equal(g1, g0)


@pytest.mark.parametrize('prec', ['fp32', 'bf16'])
def test_explicit_dout_backward_after_the_fused_forward(lib, prec):
    """eld_unet_forward_loss_ex leaves the input with the caller (no copy in the workspace); a backward with an EXPLICIT dout that follows
    it on the same workspace is a valid call order (include/eld_amd.h) and must read the first layer's operand from the caller's x too -- same
    gradients, bit for bit, as the plain forward + backward pair."""
    from eld_amd.unet import UNetSeeInDark
    torch.manual_seed(4)
    net = UNetSeeInDark(4, 4).cuda()
    bf16 = prec == 'bf16'
    g = torch.Generator(device='cuda').manual_seed(9)
    shape = (2, 4, 96, 208)
    x = torch.rand(*shape, device='cuda', generator=g)
    t = torch.rand(*shape, device='cuda', generator=g)
    dout = torch.randn(*shape, device='cuda', generator=g) / x.numel()
    _, key, _ = net._engine_forward(x, save=True, bf16=bf16)
    g0 = net._engine_backward(dout, key, shape).clone()
    # poison the workspace copy a plain forward leaves behind: the fused forward must not depend on it
    junk = torch.rand(*shape, device='cuda', generator=g)
    net._engine_forward(junk, save=True, bf16=bf16)
    lb = torch.zeros(1, device='cuda')
    _, key, _ = net._engine_forward_loss(x, t, lb, bf16=bf16)
    g1 = net._engine_backward(dout, key, shape).clone()
    torch.cuda.synchronize()
    assert torch.equal(g1, g0), int((g1 != g0).sum())


def _codes_case(prec, shape, seed=11):
    from eld_amd.unet import UNetSeeInDark
    torch.manual_seed(seed)
    net = UNetSeeInDark(4, 4).cuda()
    if prec == 'fp32':
        net.fp32_products = 1
    g = torch.Generator(device='cuda').manual_seed(seed)
    x = torch.rand(*shape, device='cuda', generator=g) ** 2.2
    junk = torch.rand(*shape, device='cuda', generator=g)
    dout = torch.randn(*shape, device='cuda', generator=g) / x.numel()
    return net, x, junk, dout


@pytest.mark.parametrize('prec,shape', [('fp32', (1, 4, 528, 1072)), ('bf16', (1, 4, 256, 512)), ('bf16', (2, 4, 272, 560)), ('bf16', (1, 4, 1424, 2128))])
def test_slope_codes_give_the_gradients_of_the_saved_activations_bit_for_bit(lib, prec, shape):
    """Round 5: the forward epilogues of levels 0 / 1 (fp32 three-piece scheme) and of conv1_1 / conv9_1 (bf16) also write 2-bit slope codes, and the
    backward-data epilogues read them instead of the saved activations (conv.h ConvArgs::codes_out / codes0).  eld_debug_kernel_mask bit 7 switches
    the codes off: the same kernels then multiply by lrelu_slope(saved activation) -- the same three slope values, so every gradient must agree to
    the last bit (ragged tiles included: 272 x 560 is 17 x 17.5 tiles)."""
    net, x, junk, dout = _codes_case(prec, shape)
    bf16 = prec == 'bf16'
    _, key, _ = net._engine_forward(x, save=True, bf16=bf16)
    g1 = net._engine_backward(dout, key, shape).clone()
    old = lib.eld_debug_kernel_mask(128)
    try:
        _, key, _ = net._engine_forward(x, save=True, bf16=bf16)
        g0 = net._engine_backward(dout, key, shape).clone()
    finally:
        lib.eld_debug_kernel_mask(old)
    torch.cuda.synchronize()
    assert torch.isfinite(g0).all() and float(g0.abs().max()) > 0
    assert torch.equal(g1, g0), int((g1 != g0).sum())


@pytest.mark.parametrize('prec,shape', [('fp32', (1, 4, 528, 1072)), ('fp32', (2, 4, 144, 208)), ('bf16', (1, 4, 256, 512)), ('bf16', (2, 4, 272, 560))])
def test_pool_codes_route_ties_like_the_saved_activations(lib, prec, shape):
    """Round 6: the forward epilogues of conv1_2 / conv2_2 also write the ARGMAX of every 2x2 pooling window (2 bits per pooled element,
    ConvArgs::pool_codes_out) and their own slope codes, and the pools' backward reads those instead of the saved un-pooled tensors.  The
    winner of a window with equal maxima is the FIRST in row-major order (torch's CPU max_pool2d backward; unet_misc.hip POOL_BWD_1), which random
    inputs never exercise -- so this input is piecewise constant on 6 x 10 blocks (offset against the 2 x 2 windows): inside a block all four
    window elements of conv1_2's output are the same bits, across a block edge two of them are.  eld_debug_kernel_mask bit 8 switches the pool
    codes off (bit 7: all codes): every gradient must agree to the last bit."""
    net, x, junk, dout = _codes_case(prec, shape, seed=13)
    N, C, H, W = shape
    g = torch.Generator(device='cuda').manual_seed(21)
    coarse = torch.rand(N, C, (H + 5) // 6 + 1, (W + 9) // 10 + 1, device='cuda', generator=g)
    x = coarse.repeat_interleave(6, 2).repeat_interleave(10, 3)[:, :, 1:H + 1, 3:W + 3].contiguous()
    assert x.shape == shape
    bf16 = prec == 'bf16'
    _, key, _ = net._engine_forward(x, save=True, bf16=bf16)
    g1 = net._engine_backward(dout, key, shape).clone()
    res = []
    for mask in (256, 128):
        old = lib.eld_debug_kernel_mask(mask)
        try:
            _, key, _ = net._engine_forward(x, save=True, bf16=bf16)
            res.append(net._engine_backward(dout, key, shape).clone())
        finally:
            lib.eld_debug_kernel_mask(old)
    torch.cuda.synchronize()
    for g0 in res:
        assert torch.isfinite(g0).all() and float(g0.abs().max()) > 0
        assert torch.equal(g1, g0), int((g1 != g0).sum())


@pytest.mark.parametrize('prec', ['fp32', 'bf16'])
def test_backward_reads_slope_codes_only_from_the_forward_that_wrote_them(lib, prec):
    """Which code regions are valid is a property of the LAST forward on a workspace (its fp32 scheme, its kernels), not of the backward's own
    switches: a forward that writes none (fp32: the two-piece scheme; bf16: conv_bfs switched off) followed by a backward whose kernels could read
    them (the saved tensors have one layout; include/eld_amd.h asks for one scheme per forward / backward pair, but a caller who mixes them must not be
    handed another input's slopes silently) falls back to the saved activations -- not to what an earlier forward left."""
    shape = (1, 4, 528, 1072) if prec == 'fp32' else (1, 4, 256, 512)
    net, x, junk, dout = _codes_case(prec, shape, seed=12)
    bf16 = prec == 'bf16'

    def step(codes_off):
        old = lib.eld_debug_kernel_mask(128) if codes_off else None
        try:
            net.fp32_products = 1
            net._engine_forward(junk, save=True, bf16=bf16)                   # leaves junk's codes in the workspace (unless switched off)
            if bf16:
                m = lib.eld_debug_kernel_mask(0)                              # (returns the previous mask)
                lib.eld_debug_kernel_mask(m | 1)                              # the forward of x: no conv_bfs, so no codes
                _, key, _ = net._engine_forward(x, save=True, bf16=True)
                lib.eld_debug_kernel_mask(m)                                  # the backward: conv_bfs again
            else:
                net.fp32_products = 2                                         # the forward of x: a scheme that writes no codes
                _, key, _ = net._engine_forward(x, save=True)
                net._ws.algo[key] = 1                                         # the backward: the three-piece kernels
            return net._engine_backward(dout, key, shape).clone()
        finally:
            if codes_off:
                lib.eld_debug_kernel_mask(old)

    ga = step(False)
    gb = step(True)
    torch.cuda.synchronize()
    assert torch.equal(ga, gb), int((ga != gb).sum())


@pytest.mark.parametrize('prec', ['fp32', 'bf16'])
@pytest.mark.parametrize('shape', [(1, 4, 64, 144), (2, 4, 272, 560), (1, 4, 528, 1072)])
def test_inference_entry_point_equals_the_training_forward(lib, prec, shape):
    """eld_unet_infer_ex (the forward under torch.no_grad(): ELD_model.py:203-307) skips what only a backward reads (input copy, slope codes) and must
    give eld_unet_forward_ex's output bit for bit; a backward on a workspace it filled last is refused, not served stale operands."""
    from eld_amd import _lib as L
    from eld_amd.unet import UNetSeeInDark
    torch.manual_seed(3)
    net = UNetSeeInDark(4, 4).cuda()
    N, _, H, W = shape
    g = torch.Generator(device='cuda').manual_seed(H)
    x = torch.rand(*shape, device='cuda', generator=g) ** 2.2
    p = 1 if prec == 'bf16' else 0
    nbytes = lib.eld_unet_workspace_bytes(N, H, W, 4, 4)
    ws = torch.empty(nbytes, dtype=torch.uint8, device='cuda')
    o_train, o_inf, grads = torch.empty_like(x), torch.empty_like(x), torch.empty_like(net.flat_params)
    args = (L.dptr(ws), nbytes, N, H, W, 4, 4, p, 1, L.cur_stream())
    assert lib.eld_unet_forward_ex(L.dptr(x), L.dptr(net.flat_params), L.dptr(o_train), *args) == 0
    assert lib.eld_unet_infer_ex(L.dptr(x), L.dptr(net.flat_params), L.dptr(o_inf), *args) == 0
    dout = torch.randn(*shape, device='cuda', generator=g)
    rc = lib.eld_unet_backward_ex(L.dptr(dout), L.dptr(net.flat_params), L.dptr(grads), L.dptr(ws), nbytes, N, H, W, 4, 4, p, 1, None, None, 0, L.cur_stream())
    assert rc != 0                                                   # nothing was kept for a backward
    assert lib.eld_unet_forward_ex(L.dptr(x), L.dptr(net.flat_params), L.dptr(o_train), *args) == 0
    rc = lib.eld_unet_backward_ex(L.dptr(dout), L.dptr(net.flat_params), L.dptr(grads), L.dptr(ws), nbytes, N, H, W, 4, 4, p, 1, None, None, 0, L.cur_stream())
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.equal(o_inf, o_train)
    # the module: no_grad forwards go through the inference entry point, and a training step on the same module still works afterwards
    net.inference_precision = prec
    net.fp32_products = 1
    with torch.no_grad():
        assert torch.equal(net(x), o_train)


def test_forward_loss_argument_checks(lib):
    """eld_unet_forward_loss_ex rejects what it cannot run (include/eld_amd.h): a missing target / loss pointer, an unknown loss kind, a workspace
    that is too small -- error codes, no launch."""
    from eld_amd.unet import UNetSeeInDark
    from eld_amd import _lib as L
    net = UNetSeeInDark(4, 4).cuda()
    N, H, W = 1, 64, 64
    x = torch.rand(N, 4, H, W, device='cuda')
    t = torch.rand(N, 4, H, W, device='cuda')
    out = torch.empty_like(x)
    loss = torch.zeros(1, device='cuda')
    nbytes = lib.eld_unet_workspace_bytes(N, H, W, 4, 4)
    ws = torch.empty(nbytes, dtype=torch.uint8, device='cuda')

    def call(target, lossp, wsbytes, kind):
        return lib.eld_unet_forward_loss_ex(L.dptr(x), L.dptr(net.flat_params), target, L.dptr(out), lossp, L.dptr(ws), wsbytes, N, H, W, 4, 4, 0, 1, kind, 1.0,
                                            L.cur_stream())
    assert call(L.dptr(t), L.dptr(loss), nbytes, 0) == 0
    assert call(None, L.dptr(loss), nbytes, 0) != 0
    assert call(L.dptr(t), None, nbytes, 0) != 0
    assert call(L.dptr(t), L.dptr(loss), nbytes, 2) != 0
    assert call(L.dptr(t), L.dptr(loss), nbytes // 2, 0) != 0
    torch.cuda.synchronize()


def test_backward_without_dout_needs_the_fused_forward(lib):
    """eld_unet_backward_ex(dout = NULL) consumes head state only eld_unet_forward_loss_ex leaves in the workspace (include/eld_amd.h): after a
    plain forward on the same workspace, or for another shape / precision, it must return an error code instead of stale gradients; the module
    raises before it gets that far."""
    from eld_amd.unet import UNetSeeInDark
    from eld_amd import _lib as L
    net = UNetSeeInDark(4, 4).cuda()
    N, H, W = 1, 64, 64
    x = torch.rand(N, 4, H, W, device='cuda')
    t = torch.rand(N, 4, H, W, device='cuda')
    out = torch.empty_like(x)
    loss = torch.zeros(1, device='cuda')
    grads = torch.empty(net.flat_params.numel(), device='cuda')
    nbytes = lib.eld_unet_workspace_bytes(N, H, W, 4, 4)
    ws = torch.empty(nbytes, dtype=torch.uint8, device='cuda')

    def fwd_loss(prec=0):
        return lib.eld_unet_forward_loss_ex(L.dptr(x), L.dptr(net.flat_params), L.dptr(t), L.dptr(out), L.dptr(loss), L.dptr(ws), nbytes, N, H, W, 4, 4, prec, 1, 0, 1.0,
                                            L.cur_stream())

    def bwd(prec=0, n=N):
        return lib.eld_unet_backward_ex(None, L.dptr(net.flat_params), L.dptr(grads), L.dptr(ws), nbytes, n, H, W, 4, 4, prec, 1, None, None, 0, L.cur_stream())
    assert fwd_loss() == 0 and bwd() == 0               # the pair the training step issues
    assert bwd(prec=1) != 0                             # the head state is fp32
    assert lib.eld_unet_forward_ex(L.dptr(x), L.dptr(net.flat_params), L.dptr(out), L.dptr(ws), nbytes, N, H, W, 4, 4, 0, 1, L.cur_stream()) == 0
    assert bwd() != 0                                   # a plain forward overwrote it
    assert fwd_loss(prec=1) == 0 and bwd(prec=1) == 0 and bwd(prec=0) != 0
    torch.cuda.synchronize()
    lb = torch.zeros(1, device='cuda')
    _, key, _ = net._engine_forward_loss(x, t, lb)
    net._engine_forward(x, save=True)                   # same workspace key ('train', N, H, W)
    with pytest.raises(RuntimeError):
        net._engine_backward(None, key, (N, 4, H, W))
