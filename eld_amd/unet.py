"""Arch plugin: drop-in for the reference `models.arch.unet` factory (models/arch/__init__.py:6-7) and
`UNetSeeInDark` (models/arch/Unet.py:6-104).

Same constructor, same 46 parameter names/shapes/default init (so state_dicts interchange with the
reference, models/ELD_model.py:516-523 and App. B of SURVEY.md), same forward(x: Bx4xHxW) -> BxCxHxW
with autograd support -- but the whole forward and the whole backward are each ONE call into the HIP
engine (include/eld_amd.h: eld_unet_forward / eld_unet_backward).  All parameters are views into one
flat float32 buffer (`flat_params`), which is what the engine consumes and what the data-parallel
gradient all-reduce and the fused Adam operate on.
"""
import ctypes as C

import torch
import torch.nn as nn

from . import _lib as L

NAMES = (['conv%d_%d' % (l, k) for l in range(1, 6) for k in (1, 2)] +
         [n for l in range(6, 10) for n in ('upv%d' % l, 'conv%d_1' % l, 'conv%d_2' % l)] + ['conv10_1'])


def param_offsets(in_ch, out_ch):
    offs = (C.c_int64 * 47)()
    L.check(L.lib().eld_unet_param_offsets(in_ch, out_ch, offs), 'eld_unet_param_offsets')
    return list(offs)


class _Workspace:
    """Scratch owned by the module, one per (mode, N, H, W); holds the saved activations between the
    forward and the backward of a training step."""
    def __init__(self):
        self.bufs = {}
        self.gen = {}
        self.algo = {}       # fp32 product scheme the forward that filled the buffer ran with
        self.fused_head = {} # True: the buffer was filled by eld_unet_forward_loss_ex (the only forward a dout = None backward may follow)
        self.x_ref = {}      # the fused-loss forward's (converted) input: the library reads it again in the backward (include/eld_amd.h), so it must outlive this call

    def get(self, key, nbytes, device):
        b = self.bufs.get(key)
        if b is None or b.numel() < nbytes or b.device != device:
            if len(self.bufs) > 4:          # shapes changed (e.g. chop tiles): drop old scratch
                self.bufs.clear(); self.gen.clear(); self.algo.clear(); self.fused_head.clear(); self.x_ref.clear()
            b = torch.empty(nbytes, dtype=torch.uint8, device=device)
            self.bufs[key] = b
            self.gen[key] = 0
        return b


class _UNetFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, x, *params):
        out, key, gen = net._engine_forward(x, save=True, bf16=net.train_precision == 'bf16')
        ctx.net, ctx.key, ctx.gen, ctx.shape = net, key, gen, tuple(x.shape)
        return out

    @staticmethod
    def backward(ctx, dout):
        net = ctx.net
        if net._ws.gen.get(ctx.key) != ctx.gen:
            raise RuntimeError('eld_amd U-Net: the saved activations of this forward were overwritten by a later '
                               'forward of the same shape before backward ran')
        grads = net._engine_backward(dout.contiguous(), ctx.key, ctx.shape)
        views = [grads[o:o + p.numel()].view_as(p) for o, p in zip(net._offsets[:-1], net._plist)]
        return (None, None) + tuple(views)


class UNetSeeInDark(nn.Module):
    def __init__(self, in_channels=4, out_channels=3):
        super(UNetSeeInDark, self).__init__()
        ch = [32, 64, 128, 256, 512]
        # holders with the reference's names, construction order and default init (Unet.py:11-46):
        # same torch.manual_seed => same initial weights as the reference module.
        self.conv1_1 = nn.Conv2d(in_channels, 32, kernel_size=3, stride=1, padding=1)
        self.conv1_2 = nn.Conv2d(32, 32, kernel_size=3, stride=1, padding=1)
        for l in range(1, 5):
            setattr(self, 'conv%d_1' % (l + 1), nn.Conv2d(ch[l - 1], ch[l], kernel_size=3, stride=1, padding=1))
            setattr(self, 'conv%d_2' % (l + 1), nn.Conv2d(ch[l], ch[l], kernel_size=3, stride=1, padding=1))
        for i, l in enumerate(range(3, -1, -1)):
            k = 6 + i
            setattr(self, 'upv%d' % k, nn.ConvTranspose2d(ch[l + 1], ch[l], 2, stride=2))
            setattr(self, 'conv%d_1' % k, nn.Conv2d(ch[l + 1], ch[l], kernel_size=3, stride=1, padding=1))
            setattr(self, 'conv%d_2' % k, nn.Conv2d(ch[l], ch[l], kernel_size=3, stride=1, padding=1))
        self.conv10_1 = nn.Conv2d(32, out_channels, kernel_size=1, stride=1)
        self.in_channels, self.out_channels = in_channels, out_channels
        self._offsets = param_offsets(in_channels, out_channels)
        self._ws = _Workspace()
        self._flat = None
        self._flatten()

    # ---- flat parameter buffer -------------------------------------------------------------------------
    @property
    def _plist(self):
        return [p for n in NAMES for p in (getattr(self, n).weight, getattr(self, n).bias)]

    def _flatten(self):
        plist = self._plist
        assert [p.numel() for p in plist] == [b - a for a, b in zip(self._offsets[:-1], self._offsets[1:])]
        dev = plist[0].device
        flat = torch.empty(self._offsets[-1], dtype=torch.float32, device=dev)
        for o, p in zip(self._offsets[:-1], plist):
            flat[o:o + p.numel()].copy_(p.data.reshape(-1))
            p.data = flat[o:o + p.numel()].view(p.shape)
        self._flat = flat

    def _is_flat(self):
        base = self._flat.data_ptr()
        return all(p.data_ptr() == base + 4 * o and p.device == self._flat.device for o, p in zip(self._offsets[:-1], self._plist))

    @property
    def flat_params(self):
        if not self._is_flat():
            self._flatten()
        return self._flat

    def _apply(self, fn, *a, **k):       # .to(device) / .cuda() / .float(): keep the views coherent
        r = super()._apply(fn, *a, **k)
        self._flatten()
        return r

    # ---- engine calls ----------------------------------------------------------------------------------
    inference_precision = 'fp32'          # 'bf16': no-grad forwards run eld_unet_forward_bf16 (BASELINE config 3 precision)
    fp32_products = None                  # None: process default (eld_conv_fp32_algo / env ELD_FP32_CONV); 0 / 1 / 2 pins the scheme for this module
    train_precision = 'fp32'              # 'bf16': training forwards/backwards run the bf16 engine (fp32 master weights/grads)

    def _engine_forward(self, x, save, bf16=False):
        if not x.is_cuda:
            raise RuntimeError('eld_amd U-Net runs on the GPU only (no CPU fallback); got a CPU tensor')
        x = x.contiguous().float()
        N, Cc, H, W = x.shape
        if Cc != self.in_channels:
            raise RuntimeError('expected %d input channels, got %d' % (self.in_channels, Cc))
        if H % 16 or W % 16:
            raise RuntimeError('U-Net input H, W must be multiples of 16 (4 pooling levels), got %dx%d' % (H, W))
        nbytes = L.lib().eld_unet_workspace_bytes(N, H, W, self.in_channels, self.out_channels)
        key = (('train_bf16' if bf16 else 'train') if save else ('eval_bf16' if bf16 else 'eval'), N, H, W)
        ws = self._ws.get(key, nbytes, x.device)
        self._ws.gen[key] += 1
        out = torch.empty((N, self.out_channels, H, W), dtype=torch.float32, device=x.device)
        # the fp32 product scheme is named per call and remembered with the saved activations: the backward of THIS forward
        # runs the same scheme even if the process default (eld_conv_fp32_algo) is changed in between
        algo = self.fp32_products if self.fp32_products is not None else L.lib().eld_conv_fp32_algo(-1)
        self._ws.algo[key] = algo
        self._ws.fused_head[key] = False
        self._ws.x_ref.pop(key, None)
        # save = False (torch.no_grad(): ELDModel.eval / test): the inference entry point -- same output bits, nothing kept for a backward
        fwd, name = (L.lib().eld_unet_forward_ex, 'eld_unet_forward_ex') if save else (L.lib().eld_unet_infer_ex, 'eld_unet_infer_ex')
        L.check(fwd(L.dptr(x), L.dptr(self.flat_params), L.dptr(out), L.dptr(ws), ws.numel(),
                    N, H, W, self.in_channels, self.out_channels, 1 if bf16 else 0, algo, L.cur_stream()), name)
        return out, key, self._ws.gen[key]

    def _engine_forward_loss(self, x, target, loss_buf, bf16=False, mse=False, grad_scale=1.0):
        """The training forward with the loss fused into the head (include/eld_amd.h eld_unet_forward_loss_ex): returns (out, key, generation);
        the mean loss lands in loss_buf (a 1-element CUDA float tensor).  The matching backward is _engine_backward(None, key, shape)."""
        if not x.is_cuda:
            raise RuntimeError('eld_amd U-Net runs on the GPU only (no CPU fallback); got a CPU tensor')
        x = x.contiguous().float()
        target = target.contiguous().float()
        N, Cc, H, W = x.shape
        if Cc != self.in_channels:
            raise RuntimeError('expected %d input channels, got %d' % (self.in_channels, Cc))
        if H % 16 or W % 16:
            raise RuntimeError('U-Net input H, W must be multiples of 16 (4 pooling levels), got %dx%d' % (H, W))
        if tuple(target.shape) != (N, self.out_channels, H, W) or not target.is_cuda:
            raise RuntimeError('target must be a CUDA tensor of shape %s, got %s' % ((N, self.out_channels, H, W), tuple(target.shape)))
        nbytes = L.lib().eld_unet_workspace_bytes(N, H, W, self.in_channels, self.out_channels)
        key = ('train_bf16' if bf16 else 'train', N, H, W)
        ws = self._ws.get(key, nbytes, x.device)
        self._ws.gen[key] += 1
        out = torch.empty((N, self.out_channels, H, W), dtype=torch.float32, device=x.device)
        algo = self.fp32_products if self.fp32_products is not None else L.lib().eld_conv_fp32_algo(-1)
        self._ws.algo[key] = algo
        self._ws.fused_head[key] = True
        self._ws.x_ref[key] = x          # x.contiguous().float() may be a temporary: keep it until the next forward of this shape replaces it
        L.check(L.lib().eld_unet_forward_loss_ex(L.dptr(x), L.dptr(self.flat_params), L.dptr(target), L.dptr(out), L.dptr(loss_buf), L.dptr(ws), ws.numel(),
                                                 N, H, W, self.in_channels, self.out_channels, 1 if bf16 else 0, algo, 1 if mse else 0, float(grad_scale),
                                                 L.cur_stream()), 'eld_unet_forward_loss_ex')
        return out, key, self._ws.gen[key]

    def _engine_backward(self, dout, key, shape, grads=None, buckets=None):
        """buckets: eld_amd.dist.GradBuckets -- record one event per gradient bucket as soon as it is final (data-parallel overlap).
        dout = None: the forward was _engine_forward_loss (the head's share of the backward is already in the workspace)."""
        N, _, H, W = shape
        ws = self._ws.bufs[key]
        if dout is None and not self._ws.fused_head.get(key):
            raise RuntimeError('eld_amd U-Net: backward without an output gradient needs the fused-loss forward (_engine_forward_loss) '
                               'to be the last forward of this shape; a plain forward overwrote its head state')
        if grads is None:
            grads = torch.empty(self._offsets[-1], dtype=torch.float32, device=ws.device)
        starts, events, nb = (buckets.starts_c, buckets.events_c, buckets.n) if buckets is not None else (None, None, 0)
        L.check(L.lib().eld_unet_backward_ex(L.dptr(dout) if dout is not None else None, L.dptr(self.flat_params), L.dptr(grads), L.dptr(ws), ws.numel(), N, H, W,
                                             self.in_channels, self.out_channels, 1 if key[0] == 'train_bf16' else 0, self._ws.algo[key],
                                             starts, events, nb, L.cur_stream()), 'eld_unet_backward_ex')
        return grads

    def release_workspaces(self):
        """Drop the module's scratch (saved activations included): a sweep over sensor shapes holds tens of GB per shape otherwise."""
        ws = self._ws
        ws.bufs.clear(); ws.gen.clear(); ws.algo.clear(); ws.fused_head.clear(); ws.x_ref.clear()

    def forward(self, x):
        if torch.is_grad_enabled() and any(p.requires_grad for p in self._plist):
            return _UNetFunction.apply(self, x, *self._plist)
        return self._engine_forward(x, save=False, bf16=self.inference_precision == 'bf16')[0]

    def lrelu(self, x):                  # Unet.py:102-104 (kept for API parity; the engine fuses it)
        return torch.max(0.2 * x, x)


def unet(in_channels, out_channels, **kwargs):      # models/arch/__init__.py:6-7
    return UNetSeeInDark(in_channels, out_channels)
