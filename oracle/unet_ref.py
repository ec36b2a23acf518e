"""CPU oracle for the U-Net step -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Functional torch-CPU float32/float64 restatement of UNetSeeInDark.forward
(models/arch/Unet.py:48-91), lrelu (Unet.py:102-104), nn.L1Loss (models/losses.py:32) and
torch.optim.Adam as configured in models/ELD_model.py:400-401.  Floating-point kernels keep a torch
reference (task tier rule); this one is pinned on CPU against tests/golden/unet.npz, which was minted
by running the reference module itself (oracle/gen_golden.py).
Third-party arithmetic (torch's CPU conv / pool / Adam kernels) is the pin at this boundary; the
reference ran cuDNN (train_syn.py:17) and pins no version.
"""
import torch
import torch.nn.functional as F


def lrelu(x):                                     # Unet.py:102-104
    return torch.max(0.2 * x, x)


def unet_forward(sd, x):
    """sd: state_dict-like mapping with the reference's 46 keys; x: (B,C,H,W) CPU tensor."""
    def conv(name, t, act=True):
        y = F.conv2d(t, sd[name + '.weight'], sd[name + '.bias'], stride=1, padding=sd[name + '.weight'].shape[-1] // 2)
        return lrelu(y) if act else y

    def up(name, t):
        return F.conv_transpose2d(t, sd[name + '.weight'], sd[name + '.bias'], stride=2)
    skips = []
    t = x
    for l in range(1, 5):                         # Unet.py:49-63
        t = conv('conv%d_2' % l, conv('conv%d_1' % l, t))
        skips.append(t)
        t = F.max_pool2d(t, kernel_size=2)
    t = conv('conv5_2', conv('conv5_1', t))       # Unet.py:65-66
    for l in range(6, 10):                        # Unet.py:68-86, cat order [up, skip]
        t = torch.cat([up('upv%d' % l, t), skips[9 - l]], 1)
        t = conv('conv%d_2' % l, conv('conv%d_1' % l, t))
    return conv('conv10_1', t, act=False)         # Unet.py:88


def forward_chop(sd, x, base=16):
    """models/ELD_model.py:434-467: the frame as four overlapping quadrants, each h//2 (w//2) plus a shave that is the
    round-up of the half to a multiple of `base`, +base more when that is below 10 (:438-442); each output quadrant is cut
    from its tile (:456-463).  Pinned by tests/golden/eval.npz (minted by the reference method, oracle/gen_golden_eval.py)."""
    import math
    b, c, h, w = x.shape
    hh, wh = h // 2, w // 2
    sh = math.ceil(hh / base) * base - hh
    sw = math.ceil(wh / base) * base - wh
    hs, ws = hh + (sh if sh >= 10 else sh + base), wh + (sw if sw >= 10 else sw + base)
    o = [unet_forward(sd, t) for t in (x[:, :, :hs, :ws], x[:, :, :hs, w - ws:], x[:, :, h - hs:, :ws], x[:, :, h - hs:, w - ws:])]
    out = x.new_empty(b, o[0].shape[1], h, w)
    out[:, :, :hh, :wh] = o[0][:, :, :hh, :wh]
    out[:, :, :hh, wh:] = o[1][:, :, :hh, ws - w + wh:]
    out[:, :, hh:, :wh] = o[2][:, :, hs - h + hh:, :wh]
    out[:, :, hh:, wh:] = o[3][:, :, hs - h + hh:, ws - w + wh:]
    return out


def seeded_state_dict(in_ch=4, out_ch=4, seed=2018, dtype=torch.float32):
    """Default-initialised parameters in the reference's construction order (Unet.py:11-46), so that
    torch.manual_seed(seed) reproduces the reference module's initial weights."""
    import torch.nn as nn
    torch.manual_seed(seed)
    ch = [32, 64, 128, 256, 512]
    mods = {'conv1_1': nn.Conv2d(in_ch, 32, 3, 1, 1), 'conv1_2': nn.Conv2d(32, 32, 3, 1, 1)}
    for l in range(1, 5):
        mods['conv%d_1' % (l + 1)] = nn.Conv2d(ch[l - 1], ch[l], 3, 1, 1)
        mods['conv%d_2' % (l + 1)] = nn.Conv2d(ch[l], ch[l], 3, 1, 1)
    for i, l in enumerate(range(3, -1, -1)):
        k = 6 + i
        mods['upv%d' % k] = nn.ConvTranspose2d(ch[l + 1], ch[l], 2, stride=2)
        mods['conv%d_1' % k] = nn.Conv2d(ch[l + 1], ch[l], 3, 1, 1)
        mods['conv%d_2' % k] = nn.Conv2d(ch[l], ch[l], 3, 1, 1)
    mods['conv10_1'] = nn.Conv2d(32, out_ch, 1, 1)
    sd = {}
    for n, m in mods.items():
        sd[n + '.weight'] = m.weight.detach().to(dtype)
        sd[n + '.bias'] = m.bias.detach().to(dtype)
    return sd


def loss_and_grads(sd, x, target, loss='l1'):
    """Forward + L1 (or MSE) mean loss + backward (models/ELD_model.py:411-420).  Returns (out, loss, grads)."""
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    out = unet_forward(p, x)
    crit = F.l1_loss if loss == 'l1' else F.mse_loss
    lv = crit(out, target)
    lv.backward()
    return out.detach(), float(lv), {k: v.grad for k, v in p.items()}


def adam_step(p, g, m, v, step, lr=1e-4, b1=0.9, b2=0.999, eps=1e-8, wd=0.0):
    """torch.optim.Adam single-tensor math (amsgrad off), in place on CPU tensors."""
    if wd:
        g = g + wd * p
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = (v.sqrt() / (bc2 ** 0.5)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bc1))
