"""Model plugin: mirror of the reference's ELDModel (models/ELD_model.py:172-523, models/base_model.py)
for the raw->raw path, with the whole training iteration on the MI355X:

    set_input        device transfer (ELD_model.py:173-200); when the batch carries no 'input' the noisy
                     input is synthesised ON DEVICE from 'target' by the fused sampler + clip
                     (what SynDataset.__getitem__ does per sample on the CPU, sid_dataset.py:259-280)
    optimize_parameters   forward -> L1 -> backward -> [RCCL all-reduce] -> Adam  (ELD_model.py:469-475),
                     five calls into the C ABI, no autograd graph
    get_current_errors    OrderedDict(Pixel=loss.item())                         (ELD_model.py:477-482)
    save / load / state_dict   the reference's checkpoint dict {'netG','opt_g','epoch','iterations'} with a
                     torch-Adam-shaped 'opt_g' (ELD_model.py:492-523, base_model.py:55-66)

`eld_model()` is the factory `models.__dict__[opt.model]()` resolves (engine.py:26, models/__init__.py:3-4).
The sRGB stages (--stage_in/out srgb, util/process.py) run through eld_isp_process (3-channel ends); an unknown stage raises
NotImplementedError('Invalid Stage') like ELD_model.py:377-389.
"""
import os
from collections import OrderedDict

import numpy as np
import torch

from . import _lib as L
from . import dist as D
from . import unet as arch_unet
from .data import records_from_batch
from .noise import (NoiseModel, augment, decode_augment_u16, is_u16_codes, make_records, model_flags, sample_noise_records, set_sample_ids)

ARCH = {'unet': arch_unet.unet}          # the `arch.__dict__[opt.netG]` registry (ELD_model.py:391)


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam semantics (ELD_model.py:400-401) as ONE HIP launch over the flat parameter buffer.
    It is a real torch Optimizer: `param_groups` is what Engine.set_learning_rate edits (engine.py:109-112),
    and state_dict()/load_state_dict() speak torch Adam's per-parameter format so checkpoints interchange."""

    def __init__(self, net, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.net = net
        super().__init__(list(net.parameters()), dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        flat = net.flat_params
        self.exp_avg = torch.zeros_like(flat)
        self.exp_avg_sq = torch.zeros_like(flat)
        self.step_count = 0
        self.grads = torch.zeros_like(flat)        # flat gradient buffer the engine writes and RCCL reduces

    @torch.no_grad()
    def step(self, grad_scale=1.0):
        g = self.param_groups[0]
        self.step_count += 1
        flat = self.net.flat_params
        L.check(L.lib().eld_adam_step(L.dptr(flat), L.dptr(self.grads), L.dptr(self.exp_avg), L.dptr(self.exp_avg_sq), flat.numel(),
                                      float(g['lr']), float(g['betas'][0]), float(g['betas'][1]), float(g['eps']),
                                      float(g['weight_decay']), self.step_count, float(grad_scale), L.cur_stream()), 'eld_adam_step')

    def zero_grad(self, set_to_none=True):          # the engine overwrites every gradient element each step
        pass

    def state_dict(self):
        offs = self.net._offsets
        state = {}
        for i, (a, b) in enumerate(zip(offs[:-1], offs[1:])):
            shape = self.net._plist[i].shape
            if self.step_count > 0:
                state[i] = {'step': torch.tensor(float(self.step_count)), 'exp_avg': self.exp_avg[a:b].view(shape).clone(),
                            'exp_avg_sq': self.exp_avg_sq[a:b].view(shape).clone()}
        g = dict(self.param_groups[0])
        g['params'] = list(range(len(offs) - 1))
        for k, v in dict(amsgrad=False, maximize=False, foreach=None, capturable=False, differentiable=False, fused=None).items():
            g.setdefault(k, v)
        return {'state': state, 'param_groups': [g]}

    def load_state_dict(self, sd):
        offs = self.net._offsets
        pg = sd['param_groups'][0]
        for k in ('lr', 'betas', 'eps', 'weight_decay', 'initial_lr'):
            if k in pg:
                self.param_groups[0][k] = pg[k]
        self.step_count = 0
        for i, st in sd['state'].items():
            a, b = offs[int(i)], offs[int(i) + 1]
            self.exp_avg[a:b].copy_(st['exp_avg'].reshape(-1))
            self.exp_avg_sq[a:b].copy_(st['exp_avg_sq'].reshape(-1))
            self.step_count = int(float(st['step']))


class ELDModel:
    def name(self):
        return self.__class__.__name__.lower()

    # ---- BaseModel.initialize (base_model.py:10-16) + ELDModel.initialize (ELD_model.py:370-409) ----------
    def initialize(self, opt):
        self.opt = opt
        self.gpu_ids = opt.gpu_ids
        self.isTrain = opt.isTrain
        self.save_dir = os.path.join(getattr(opt, 'checkpoints_dir', './checkpoints'), opt.name)
        self.epoch = 0
        self.iterations = 0
        if not torch.cuda.is_available() or not self.gpu_ids:
            raise RuntimeError('eld_amd.model.ELDModel needs a GPU (gpu_ids=%r): there is no CPU fallback' % (self.gpu_ids,))
        self.device = torch.device('cuda', self.gpu_ids[0] if isinstance(self.gpu_ids, (list, tuple)) else int(self.gpu_ids))
        torch.cuda.set_device(self.device)
        self.stage_in, self.stage_out = getattr(opt, 'stage_in', 'raw'), getattr(opt, 'stage_out', 'raw')
        for st in (self.stage_in, self.stage_out):           # ELD_model.py:377-389
            if st not in ('raw', 'srgb'):
                raise NotImplementedError('Invalid Stage: {}'.format(st))
        ch = getattr(opt, 'channels', 4)
        # Input planes.  The reference always builds arch(opt.channels, ...) (ELD_model.py:391); a burst SynDataset (sid_dataset.py:267-273) hands it
        # num_burst * channels planes.  opt.in_channels (set by eld_amd.launch --num-burst) names the count explicitly; without it, a TRAINING model
        # infers it from the SynDataset the entry script built -- eval / test / resume of a non-burst checkpoint never silently change shape.
        cin = getattr(opt, 'in_channels', None)
        if cin is None:
            burst = _plugin_num_burst() if self.isTrain else 1
            if burst > 1:
                print('[i] eld_amd: input planes inferred from SynDataset(num_burst=%d): %d x %d = %d (set opt.in_channels to pin it)' % (burst, burst, ch, burst * ch))
            cin = ch * burst
        cin = 3 if self.stage_in == 'srgb' else int(cin)
        cout = 3 if self.stage_out == 'srgb' else ch
        # CRF tables (E, fs) of process.load_CRF for the sRGB input stage: opt.crf_tables, else what the entry script handed to
        # ISPDataset(CRF=...) (train_syn.py:42-58); None = gamma 2.2.  --crf without tables anywhere is an error, not a silent gamma.
        from .data import ISPDataset
        self.CRF = getattr(opt, 'crf_tables', None)
        if self.CRF is None and ISPDataset.last() is not None:
            self.CRF = ISPDataset.last().CRF
        # (The reference model loads the tables itself, ELD_model.py:374-375; this plugin does not import the host tree: the tables arrive as data.)
        if self.CRF is None and getattr(opt, 'crf', False) and self.stage_in == 'srgb' and self.isTrain:
            # only the training step renders the input on the device (set_input of a deferred ISPDataset sample); eval / test entry points
            # (test_ELD.py) get inputs rendered by their own datasets
            raise RuntimeError('--crf: no CRF tables reached the model: pass them as opt.crf_tables = (E, fs) or through ISPDataset(CRF=...) as '
                               'train_syn.py:42-58 does; refusing to render the input with gamma 2.2 against a CRF-rendered target')
        prec = getattr(opt, 'precision', os.environ.get('ELD_AMD_PRECISION', 'fp32'))      # 'bf16' = BASELINE config 3
        if prec == 'bf16' and cin > 4:
            raise NotImplementedError('precision=bf16 supports up to 4 input planes (got %d: burst inputs run in fp32)' % cin)
        self.netG = ARCH[getattr(opt, 'netG', 'unet')](cin, cout).to(self.device)
        self.netG.train_precision = self.netG.inference_precision = prec
        self.world, self.rank = D.world_size(), D.rank()
        self.exchange = True                                 # False: skip the gradient all-reduce (bench.py measures its exposed cost)
        self._buckets = None
        if self.world > 1:
            D.broadcast_(self.netG.flat_params, 0)          # identical replicas
        self.loss_pixel = None
        self.loss_name = getattr(opt, 'loss', 'l1')
        if self.loss_name not in ('l1', 'l2'):                   # models/losses.py:31-36
            raise NotImplementedError('loss %r: the reference knows l1 and l2' % (self.loss_name,))
        if self.isTrain:
            self.optimizer_G = FusedAdam(self.netG, lr=opt.lr, betas=(getattr(opt, 'beta1', 0.9), 0.999),
                                         weight_decay=getattr(opt, 'wd', 0.0))
            for g in self.optimizer_G.param_groups:          # base_model.py:68-74
                g['initial_lr'] = opt.lr
            self.optimizers = [self.optimizer_G]
            self.schedulers = []
        self._l1_ws = torch.empty(L.lib().eld_l1_workspace_bytes(), dtype=torch.uint8, device=self.device)
        # last layer + loss + the head's backward as one pass over conv9_2's output (eld_unet_forward_loss_ex); ELD_FUSED_HEAD=0: the three separate kernels
        self.fused_head = os.environ.get('ELD_FUSED_HEAD', '1') != '0'
        self._loss_buf = torch.zeros(1, dtype=torch.float32, device=self.device)
        self.noise_model = NoiseModel.last_instance      # the plugin instance the entry script built (train_syn.py:38); may be None
        self._sample_counter = 0
        self._synth_stream, self._prefetched = None, None    # prefetch_input(): synthesis of the NEXT batch on a side stream
        # Philox key: the run's --seed (base_option.py:22) unless overridden; per-image counters are global sample indices
        self.seed = int(os.environ.get('ELD_AMD_SEED', getattr(opt, 'seed', None) if getattr(opt, 'seed', None) is not None else 2018))
        if getattr(opt, 'resume', False):
            self.load(self, getattr(opt, 'resume_epoch', None))

    def set_noise_model(self, noise_model):
        """Attach the noise plugin used for on-device synthesis of the training input."""
        self.noise_model = noise_model

    # ---- ELD_model.py:173-200 -----------------------------------------------------------------------------
    def set_input(self, data, mode='train'):
        """Same contract as the reference (picks 'input'/'target' by mode, moves them to the device).  Additionally, a TRAIN batch
        without 'input' is a batch of DEFERRED samples (eld_amd.data: clean 'target' codes + 'params' records + 'aug' bits +
        'burst'): what SynDataset / ELDTrainDataset.__getitem__ do per sample on the CPU (sid_dataset.py:259-280, 332-363) then
        happens here, batched, on the device: decode -> sampler + clip (burst frames concatenated on the channel axis) ->
        flips / transpose of input AND target -> clip.
        A batch that prefetch_input() already started (same dict object) is picked up from the synthesis stream instead."""
        pre, self._prefetched = self._prefetched, None
        if pre is not None and pre[0] is data and pre[1] == mode.lower():
            _, _, prepared, ev = pre
            cur = torch.cuda.current_stream()
            cur.wait_event(ev)
            for t in prepared[:2]:
                if t is not None and t.is_cuda:
                    t.record_stream(cur)                 # allocated on the synthesis stream, consumed (and released) on this one
        else:
            if pre is not None:                          # a prefetch nobody picks up: its draws were consumed; keep stream order sane and drop it
                torch.cuda.current_stream().wait_event(pre[3])
            prepared = self._prepare(data, mode)
        self.input, self.target, self.data_name, self.rawpath = prepared

    def prefetch_input(self, data, mode='train'):
        """Start set_input(data, mode) on the synthesis stream and return at once: the sampler launch of iteration i+1 then runs beside the
        U-Net kernels of iteration i (what the reference gets from its DataLoader workers, train_syn.py:78-80 / sid_dataset.py:259-280, where
        synthesis of the next batch overlaps the training step).  Call it AFTER set_input() of the current batch and BEFORE optimize_parameters():
        the synthesis stream first waits for everything the current stream holds at this point (the previous iteration), so host-side draws
        (_sample_params, sample ids) and device results are exactly those of the serial order.  set_input(data) with the same dict picks the
        tensors up; any other set_input() discards the prefetch."""
        if D.ranks_share_device():                   # several ranks time-slice this GPU (one-GPU smoke runs of the N > 1 path): no second stream, set_input() does the work
            return
        if self._synth_stream is None:
            self._synth_stream = torch.cuda.Stream(self.device)
        s = self._synth_stream
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            prepared = self._prepare(data, mode)
            ev = torch.cuda.Event()
            ev.record(s)
        self._prefetched = (data, mode.lower(), prepared, ev)

    def _prepare(self, data, mode):
        mode = mode.lower()
        if mode not in ('train', 'eval', 'test'):
            raise NotImplementedError('Mode [%s] is not implemented' % mode)
        target = data.get('target') if mode != 'test' else None
        inp = data.get('input')
        if target is not None:
            target = target.to(device=self.device, non_blocking=True)
        if inp is not None:
            inp = inp.to(device=self.device, non_blocking=True)
            if is_u16_codes(target):
                target = decode_augment_u16(target)
        elif mode == 'train' and data.get('wb') is not None:
            # --stage_in srgb (ISPDataset.__getitem__, sid_dataset.py:301-316): noise on the raw patch, clip, raw -> sRGB, clip
            clean = data['clean'].to(device=self.device, non_blocking=True)
            if data.get('params') is not None:
                inp = self.synthesize(clean, data.get('params'), data.get('sample_ids'))
            else:
                inp = decode_augment_u16(clean) if is_u16_codes(clean) else clean.float().clamp(0, 1)
            from .isp import process
            inp = process(inp, data['wb'], data['ccm'], CRF=self.CRF)          # quantised to k/255 in [0,1]: the second clip is the identity
        elif mode == 'train':
            if target is None:
                raise KeyError('target')
            # the clean patch the noise applies to: SynDataset's own sample when it differs from the target (sid_dataset.py:265-275)
            clean = data.get('clean')
            clean = target if clean is None else clean.to(device=self.device, non_blocking=True)
            inp = self.synthesize(clean, data.get('params'), data.get('sample_ids'), burst=_burst_of(data))
        else:
            raise KeyError('input')
        aug = data.get('aug')
        if aug is not None and mode == 'train':
            # ELDTrainDataset augmentation (sid_dataset.py:344-354) AFTER synthesis, as the reference does, so that
            # row banding follows the sensor rows of the un-augmented frame: same flips/transpose on input and target.
            bits = [int(b) for b in (aug.tolist() if hasattr(aug, 'tolist') else aug)]
            inp = augment(inp, bits, clip=True)
            target = decode_augment_u16(target, bits) if is_u16_codes(target) else augment(target, bits, clip=False)
        elif is_u16_codes(target):
            target = decode_augment_u16(target)
        return inp, target, data.get('fn'), data.get('rawpath')

    def synthesize(self, clean, params=None, sample_ids=None, burst=1):
        """noisy = clip(noise_model(clean)) on device, one Philox sample id per synthesised frame (global index: rank-strided so
        that a given global batch gets the same noise for every world size).  clean: CUDA float32 (N,C,H,W) or the LMDB uint16
        codes (int16 view) -- decoded inside the sampler.  params: None (drawn here with _sample_params, once per image as
        sid_dataset.py:269 does for a burst), a list of N tuples, or the (N,64) record bytes of a collated deferred batch.
        burst > 1: `burst` frames per image share the image's parameters and are concatenated on the channel axis
        (sid_dataset.py:267-273)."""
        nm = self.noise_model
        if nm is None:
            raise RuntimeError('no noise model attached (set_noise_model) and the batch has no "input"')
        N = clean.shape[0]
        if params is None:
            recs = make_records([nm._sample_params() for _ in range(N)], [0] * N)
        elif hasattr(params, 'dtype') and not isinstance(params, (list, tuple)):
            recs = records_from_batch(params.cpu().numpy() if hasattr(params, 'cpu') else params)
        else:
            recs = make_records(list(params), [0] * N)
        burst = max(1, int(burst))
        if sample_ids is None:
            base = self._sample_counter * self.world
            sample_ids = [base + self.rank + self.world * i for i in range(N * burst)]
            self._sample_counter += N * burst
        sample_ids = [int(v) for v in sample_ids]
        if len(sample_ids) != N * burst:
            raise ValueError('need %d sample ids (N x burst), got %d' % (N * burst, len(sample_ids)))
        in_u16 = is_u16_codes(clean)
        clean = clean.contiguous() if in_u16 else clean.contiguous().float()
        flags = model_flags(nm.model) | L.CLIP
        out = None
        for k in range(burst):                   # frame k of image i carries id sample_ids[i*burst + k]
            out = sample_noise_records(clean, set_sample_ids(recs, sample_ids[k::burst]), flags, self.seed, in_u16=in_u16, out=out,
                                       burst_index=k, burst=burst)
        return out

    # ---- ELD_model.py:422-432 -------------------------------------------------------------------------------
    def forward(self):
        if getattr(self.opt, 'chop', False):
            self.output = self.forward_chop(self.input)
        else:
            self.output = self.netG(self.input)
        return self.output

    def forward_chop(self, x, base=16):              # ELD_model.py:434-467, same tile arithmetic
        b, c, h, w = x.size()
        h_half, w_half = h // 2, w // 2
        shave_h = np.ceil(h_half / base) * base - h_half
        shave_w = np.ceil(w_half / base) * base - w_half
        shave_h = shave_h if shave_h >= 10 else shave_h + base
        shave_w = shave_w if shave_w >= 10 else shave_w + base
        h_size, w_size = int(h_half + shave_h), int(w_half + shave_w)
        tiles = [x[:, :, 0:h_size, 0:w_size], x[:, :, 0:h_size, (w - w_size):w],
                 x[:, :, (h - h_size):h, 0:w_size], x[:, :, (h - h_size):h, (w - w_size):w]]
        with torch.no_grad():
            outs = [self.netG(t.contiguous()) for t in tiles]
        out = x.new_empty(b, outs[0].shape[1], h, w)
        out[:, :, 0:h_half, 0:w_half] = outs[0][:, :, 0:h_half, 0:w_half]
        out[:, :, 0:h_half, w_half:w] = outs[1][:, :, 0:h_half, (w_size - w + w_half):w_size]
        out[:, :, h_half:h, 0:w_half] = outs[2][:, :, (h_size - h + h_half):h_size, 0:w_half]
        out[:, :, h_half:h, w_half:w] = outs[3][:, :, (h_size - h + h_half):h_size, (w_size - w + w_half):w_size]
        return out

    # ---- ELD_model.py:469-475: the training iteration -------------------------------------------------------
    def optimize_parameters(self, **kwargs):
        net, opt = self.netG, self.optimizer_G
        x = self.input.contiguous().float()
        tgt = self.target.contiguous().float()            # a float64 / half target (custom datasets) must not reach the float4 loads
        if self.fused_head:
            # forward() + the loss of backward_G() with the last layer, the loss and the head's backward in one pass (eld_unet_forward_loss_ex)
            if tuple(tgt.shape) != (x.shape[0], net.out_channels, x.shape[2], x.shape[3]):
                raise RuntimeError('target shape %s does not match the network output %s' % (tuple(tgt.shape), (x.shape[0], net.out_channels, x.shape[2], x.shape[3])))
            out, key, _ = net._engine_forward_loss(x, tgt, self._loss_buf, bf16=net.train_precision == 'bf16', mse=self.loss_name == 'l2')
            self.output = out
            dout = None
        else:
            out, key, _ = net._engine_forward(x, save=True, bf16=net.train_precision == 'bf16')      # forward()
            self.output = out
            dout = torch.empty_like(out)
            if tgt.shape != out.shape:
                raise RuntimeError('target shape %s does not match the network output %s' % (tuple(tgt.shape), tuple(out.shape)))
            loss_fn = L.lib().eld_mse_loss if self.loss_name == 'l2' else L.lib().eld_l1_loss
            L.check(loss_fn(L.dptr(out), L.dptr(tgt), L.dptr(dout), L.dptr(self._loss_buf), L.dptr(self._l1_ws),
                            out.numel(), 1.0, L.cur_stream()), 'eld_%s_loss' % self.loss_name)      # backward_G(): loss + its gradient
        if not self.exchange:
            net._engine_backward(dout, key, tuple(x.shape), grads=opt.grads)
            w = 1
        elif self.world > 1 and opt.grads.is_cuda:                            # data-parallel exchange (new):
            if self._buckets is None:                                         # buckets all-reduced under the rest of the backward
                self._buckets = D.GradBuckets(opt.grads.numel(), opt.grads.device)
            net._engine_backward(dout, key, tuple(x.shape), grads=opt.grads, buckets=self._buckets)      # loss.backward()
            w = self._buckets.allreduce_sum_(opt.grads)
        else:
            net._engine_backward(dout, key, tuple(x.shape), grads=opt.grads)  # loss.backward()
            w = D.allreduce_sum_(opt.grads)
        opt.step(grad_scale=1.0 / w)                                          # optimizer_G.step()
        self.loss_pixel = self._loss_buf

    def get_current_errors(self):                    # ELD_model.py:477-482 (one device sync per call, as the reference)
        ret = OrderedDict()
        if self.loss_pixel is not None:
            ret['Pixel'] = float(D.allreduce_mean_scalar(self.loss_pixel).item())
        return ret

    def update_learning_rate(self):                  # base_model.py:44-48 (no schedulers in the reference either)
        lr = self.optimizers[0].param_groups[0]['lr']
        if self.rank == 0:
            print('learning rate = %.7f' % lr)

    # ---- evaluation: forward only; PSNR as util/index.py:76-81 on the x255-clipped tensors --------------------
    @torch.no_grad()
    def eval(self, data, savedir=None, suffix=None, correct=False, crop=True, frame_id=None):
        self.set_input(data, 'eval')
        if crop:                                     # ELD_model.py:219-223
            def cc(t):
                _, _, h, w = t.shape
                y0, x0 = h // 2 - 256, w // 2 - 256
                return t[:, :, y0:y0 + 512, x0:x0 + 512].contiguous()
            self.input, self.target = cc(self.input), cc(self.target)
        out = self.forward()
        if correct:                                  # IlluminanceCorrect, ELD_model.py:138-169
            out = illuminance_correct(out, self.target)
        from .metrics import quality_assess_frames
        q = quality_assess_frames(out[:1], self.target[:1])      # tensor2im (ELD_model.py:23-38: image 0, x255, clipped, not rounded)
        psnr, ssim = q[0].tolist()                               # + util/index.py:76-81, fused on the device
        return {'PSNR': psnr, 'SSIM': ssim}

    @torch.no_grad()
    def test(self, data, savedir=None, video_mode=False):
        """ELD_model.py:309-350: forward on data['input'] (no target), returns the output tensor.  The reference then renders
        JPEGs through the author's customised rawpy (postprocess_bayer, ELD_model.py:41-60) when the sample carries a raw path;
        that renderer is outside the hot path: with `savedir` the raw network output is stored as
        <savedir>/<name>/<opt.name>.npy instead (same skip-if-present rule, ELD_model.py:321-324)."""
        self.set_input(data, 'test')
        name = None
        if self.data_name is not None and savedir is not None:
            fn = self.data_name[0] if isinstance(self.data_name, (list, tuple)) else self.data_name
            name = os.path.splitext(os.path.basename(fn))[0]
            d = os.path.join(savedir, self.opt.name) if video_mode else os.path.join(savedir, name)
            os.makedirs(d, exist_ok=True)
            path = os.path.join(d, ('%s.npy' % name) if video_mode else ('%s.npy' % self.opt.name))
            if not video_mode and os.path.exists(path):
                return None
        output = self.forward()
        if name is not None:
            np.save(path, output[0].cpu().numpy())
        return output

    # ---- checkpoints (base_model.py:55-66, ELD_model.py:492-523) -----------------------------------------------
    def state_dict(self):
        # the reference's four keys (ELD_model.py:516-523) + the sampler's position, so that --resume continues the noise stream
        # instead of replaying sample ids from 0 (the reference loads with plain dict indexing: an extra key is harmless there)
        return {'netG': self.netG.state_dict(), 'opt_g': self.optimizer_G.state_dict(), 'epoch': self.epoch, 'iterations': self.iterations,
                'eld_amd': {'sample_counter': self._sample_counter, 'seed': self.seed}}

    def save(self, label=None):
        if self.rank != 0:
            return
        os.makedirs(self.save_dir, exist_ok=True)
        name = 'model_%03d_%08d.pt' % (self.epoch, self.iterations) if label is None else 'model_' + label + '.pt'
        torch.save(self.state_dict(), os.path.join(self.save_dir, name))

    @staticmethod
    def load(model, resume_epoch=None):
        path = getattr(model.opt, 'model_path', None)
        if path is None:
            path = os.path.join(model.save_dir, 'model_latest.pt') if resume_epoch is None else \
                sorted(f for f in (os.path.join(model.save_dir, n) for n in os.listdir(model.save_dir)) if ('model_%03d' % resume_epoch) in f)[-1]
        sd = torch.load(path, map_location='cpu')
        model.netG.load_state_dict(sd['netG'])
        model.epoch, model.iterations = sd['epoch'], sd['iterations']
        extra = sd.get('eld_amd') or {}
        model._sample_counter = int(extra.get('sample_counter', 0))
        if 'seed' in extra and 'ELD_AMD_SEED' not in os.environ:
            model.seed = int(extra['seed'])
        if model.isTrain:
            model.optimizer_G.load_state_dict(sd['opt_g'])
        print('Resume from epoch %d, iteration %d' % (model.epoch, model.iterations))
        return sd


def _plugin_num_burst():
    """num_burst of the SynDataset the entry script (or eld_amd.launch --num-burst) built, 1 if none."""
    from .data import SynDataset
    inst = SynDataset.last()
    return max(1, int(getattr(inst, 'num_burst', 1) or 1)) if inst is not None else 1


def _burst_of(data):
    b = data.get('burst')
    if b is None:
        return 1
    vals = set(int(v) for v in (b.tolist() if hasattr(b, 'tolist') else ([b] if isinstance(b, int) else b)))
    if len(vals) != 1:
        raise ValueError('mixed burst counts in one batch: %r' % (sorted(vals),))
    return vals.pop()


def illuminance_correct(predict, source):
    """ELD_model.py:138-169: alpha = <p,s>/<p,p> over source != 1 on the [0,1]-clamped prediction, per image (csrc/eval.hip)."""
    from .metrics import illuminance_correct as _ic
    return _ic(predict, source)


def eld_model():                                     # models/__init__.py:3-4
    return ELDModel()
