import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import eld_amd
lib = eld_amd.load_library()
from eld_amd.unet import UNetSeeInDark
d = np.load('tests/golden/unet.npz')
torch.manual_seed(2018)
net = UNetSeeInDark(4, 4).cuda()
x, t = torch.from_numpy(d['x']).cuda(), torch.from_numpy(d['t']).cuda()
out = net(x)
print('out err', float((out.detach().cpu() - torch.from_numpy(d['out'])).abs().max()))
loss = torch.nn.L1Loss()(out, t)
print('loss', float(loss), 'golden', float(d['loss']), 'diff', float(loss) - float(d['loss']))
print('mean abs via double', float((out.detach().double() - t.double()).abs().mean()))
loss.backward()
g = dict(net.named_parameters())['conv10_1.bias'].grad
print('grad', g.cpu().numpy(), d['grad_conv10_1__bias'])
