#!/usr/bin/env python3
"""End-metric substitute for north_star's "PSNR within 0.05 dB of reference" (SID-Sony is not available here: SURVEY.md sec. 7).

Four trainings on the SAME data, from the SAME seeded initial weights (models/arch/Unet.py default init), with the reference's
recipe (train_syn.py defaults: one 4x512x512 patch per step, L1, Adam lr 1e-4, on-the-fly noise):

    oracle   torch-CPU float32 functional restatement of the reference step (oracle/unet_ref.py) + torch.optim.Adam
    fp32     HIP engine, default scheme (fp32 products as three bf16 pieces)
    fp16x2   HIP engine, eld_conv_fp32_algo(2)
    bf16     HIP engine, BASELINE configs[2]

Clean patches are synthetic scenes (smooth fields + edges, dark-heavy, on the LMDB uint16 grid).  The noisy input of every iteration
is synthesised ONCE by the fused HIP sampler (full model 'PGRU', per-iteration parameters from NoiseModel._sample_params under a fixed
NumPy seed) and handed to all four trainings -- identical noise by construction; on a few iterations the dumped Philox variates are
replayed through the oracle's reference arithmetic (oracle/noise_ref.py, noise.py:155-169) and must give the same bits.
After `--iters` iterations every weight set denoises a held-out synthetic set (fixed noise) and is scored with PSNR / SSIM as
util/index.py:76-81 computes them (eld_quality_assess on the device), on the fp32 HIP engine for all four weight sets; the
oracle-trained weights are additionally run through the torch-CPU forward to show the inference engines agree.

    python tools/psnr_parity.py [--iters 300] [--out gpurun_out/psnr_parity]

TEST INFRASTRUCTURE (imports oracle/): writes <out>.json and <out>.md.
"""
import argparse
import contextlib
import io
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np      # noqa: E402
import torch            # noqa: E402


def scene(rng, h, w):
    """One clean packed-raw scene (4,h,w) on the uint16 grid: low-frequency illumination, a few hard-edged objects, fine texture,
    per-channel white-balance-like gains, random exposure; dark-heavy like long-exposure SID frames."""
    yy, xx = np.meshgrid(np.linspace(0, 1, h, dtype=np.float32), np.linspace(0, 1, w, dtype=np.float32), indexing='ij')
    f = np.zeros((h, w), np.float32)
    for _ in range(6):
        fx, fy, ph, a = rng.uniform(-3, 3), rng.uniform(-3, 3), rng.uniform(0, 2 * np.pi), rng.uniform(0.1, 0.5)
        f += a * np.sin(2 * np.pi * (fx * xx + fy * yy) + ph).astype(np.float32)
    for _ in range(8):
        cx, cy, r, v = rng.uniform(0, 1), rng.uniform(0, 1), rng.uniform(0.03, 0.2), rng.uniform(-0.8, 0.8)
        if rng.uniform() < 0.5:
            m = ((xx - cx) ** 2 + (yy - cy) ** 2) < r * r
        else:
            m = (np.abs(xx - cx) < r) & (np.abs(yy - cy) < r * rng.uniform(0.3, 1.0))
        f = np.where(m, f + v, f)
    tex = np.zeros((h, w), np.float32)
    for _ in range(4):
        fx, fy, ph = rng.uniform(20, 60), rng.uniform(20, 60), rng.uniform(0, 2 * np.pi)
        tex += 0.04 * np.sin(2 * np.pi * (fx * xx + fy * yy) + ph).astype(np.float32)
    lum = 1.0 / (1.0 + np.exp(-(f + tex) * 2.0))
    lum = (lum ** 2.2) * rng.uniform(0.1, 1.0)
    gains = np.array([rng.uniform(0.4, 0.9), 1.0, rng.uniform(0.4, 0.9), 1.0], np.float32)
    img = np.clip(lum[None] * gains[:, None, None], 0, 1)
    return (np.floor(65535.0 * img) / 65535.0).astype(np.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=300)
    ap.add_argument('--size', type=int, default=512)
    ap.add_argument('--train-scenes', type=int, default=48)
    ap.add_argument('--eval-scenes', type=int, default=8)
    ap.add_argument('--threads', type=int, default=32)
    ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'psnr_parity'))
    args = ap.parse_args()

    import eld_amd
    lib = eld_amd.load_library()
    from eld_amd import _lib as L
    from eld_amd.metrics import quality_assess_frames
    from eld_amd.noise import NoiseModel, NoiseParams, model_flags, sample_noise
    from eld_amd.unet import UNetSeeInDark
    from oracle import noise_ref as O
    from oracle import unet_ref as U
    assert torch.cuda.is_available()
    dev = torch.device('cuda', 0)
    torch.set_num_threads(min(os.cpu_count() or 1, args.threads))
    S = args.size
    rng = np.random.default_rng(2018)
    train = [scene(rng, S, S) for _ in range(args.train_scenes)]
    held = [scene(rng, S, S) for _ in range(args.eval_scenes)]
    with contextlib.redirect_stdout(io.StringIO()):
        nm = NoiseModel(model='PGRU', include=4)
    np.random.seed(2018)
    flags = model_flags('PGRU') | L.CLIP
    order = np.random.RandomState(7).randint(0, len(train), size=args.iters)
    params = [NoiseParams.coerce(nm._sample_params()) for _ in range(args.iters)]
    eval_params = [NoiseParams.coerce(nm._sample_params()) for _ in range(len(held))]

    def oparams(p):
        return O.Params(K=p[0], g_scale=p[1], saturation=p[2], ratio=p[3], tl_lambda=p.tl_lambda, tl_scale=p.tl_scale, row_scale=p.row_scale,
                        q_step=getattr(p, 'q_step', 1.0))

    # ---- data: every iteration's noisy input, once ---------------------------------------------------------------
    replayed = 0
    inputs = []
    for it in range(args.iters):
        y = torch.from_numpy(train[order[it]][None]).to(dev)
        check = it % 50 == 0
        dump = torch.zeros(L.NPLANES, y.numel(), device=dev) if check else None
        z = sample_noise(y, [params[it]], flags, 2018, [it], dump=dump)
        if check:                                   # the oracle's reference arithmetic on the variates the kernel used: same bits
            dv = dump.cpu().numpy()
            v = {k: dv[j].reshape(y.shape[1:]) for k, j in L.PLANE.items()}
            ref = O.noise_arith(train[order[it]], oparams(params[it]), flags, **v)
            assert np.array_equal(z[0].cpu().numpy(), ref), 'iteration %d: sampler output differs from the oracle replay' % it
            replayed += 1
        inputs.append(z.cpu())
    targets = [torch.from_numpy(train[order[it]][None]) for it in range(args.iters)]
    held_y = torch.from_numpy(np.stack(held)).to(dev)
    held_z = sample_noise(held_y, eval_params, flags, 2018, [10 ** 6 + i for i in range(len(held))])

    # ---- trainings -------------------------------------------------------------------------------------------------
    sd0 = U.seeded_state_dict(4, 4, seed=2018)
    curves, weights, times = {}, {}, {}

    def train_hip(tag, precision, algo):
        net = UNetSeeInDark(4, 4)
        net.load_state_dict(sd0)
        net = net.to(dev)
        net.train_precision = precision
        from eld_amd.model import FusedAdam
        opt = FusedAdam(net, lr=1e-4)
        ws = torch.empty(lib.eld_l1_workspace_bytes(), dtype=torch.uint8, device=dev)
        lossb = torch.zeros(1, device=dev)
        prev = lib.eld_conv_fp32_algo(algo) if algo is not None else None
        cur = []
        t0 = time.time()
        try:
            for it in range(args.iters):
                x, t = inputs[it].to(dev), targets[it].to(dev)
                out, key, _ = net._engine_forward(x, save=True, bf16=precision == 'bf16')
                dout = torch.empty_like(out)
                L.check(lib.eld_l1_loss(L.dptr(out), L.dptr(t), L.dptr(dout), L.dptr(lossb), L.dptr(ws), out.numel(), 1.0, L.cur_stream()), 'eld_l1_loss')
                net._engine_backward(dout, key, tuple(x.shape), grads=opt.grads)
                opt.step()
                cur.append(float(lossb.item()))
        finally:
            if prev is not None:
                lib.eld_conv_fp32_algo(prev)
        torch.cuda.synchronize()
        times[tag] = time.time() - t0
        curves[tag] = cur
        weights[tag] = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}

    def train_oracle():
        p = {k: v.clone().requires_grad_(True) for k, v in sd0.items()}
        opt = torch.optim.Adam(list(p.values()), lr=1e-4, betas=(0.9, 0.999), weight_decay=0.0)
        cur = []
        t0 = time.time()
        for it in range(args.iters):
            opt.zero_grad()
            loss = torch.nn.functional.l1_loss(U.unet_forward(p, inputs[it]), targets[it])
            loss.backward()
            opt.step()
            cur.append(float(loss.item()))
        times['oracle'] = time.time() - t0
        curves['oracle'] = cur
        weights['oracle'] = {k: v.detach().clone() for k, v in p.items()}

    train_hip('fp32', 'fp32', 1)
    train_hip('fp16x2', 'fp32', 2)
    train_hip('bf16', 'bf16', None)
    train_oracle()

    # ---- held-out PSNR / SSIM ----------------------------------------------------------------------------------------
    def score(sd, precision='fp32'):
        net = UNetSeeInDark(4, 4)
        net.load_state_dict(sd)
        net = net.to(dev)
        net.inference_precision = precision
        with torch.no_grad():
            q = torch.cat([quality_assess_frames(net(held_z[i:i + 1].contiguous()), held_y[i:i + 1].contiguous()) for i in range(len(held))]).cpu().numpy()
        return q            # (n, 2): PSNR, SSIM

    rows = {}
    base = score(weights['oracle'])
    q_in = torch.cat([quality_assess_frames(held_z[i:i + 1].contiguous(), held_y[i:i + 1].contiguous()) for i in range(len(held))]).cpu().numpy()
    q_init = score(sd0)
    for tag in ('oracle', 'fp32', 'fp16x2', 'bf16'):
        q = score(weights[tag])
        rows[tag] = {'psnr_mean': float(q[:, 0].mean()), 'ssim_mean': float(q[:, 1].mean()), 'psnr_per_image': [float(v) for v in q[:, 0]],
                     'dpsnr_vs_oracle': float(q[:, 0].mean() - base[:, 0].mean()), 'max_abs_dpsnr_per_image': float(np.abs(q[:, 0] - base[:, 0]).max()),
                     'final_loss': curves[tag][-1], 'mean_loss_last20': float(np.mean(curves[tag][-20:])), 'train_s': round(times[tag], 2)}
    rows['bf16']['psnr_mean_bf16_inference'] = float(score(weights['bf16'], 'bf16')[:, 0].mean())
    # the oracle-trained weights through the torch-CPU forward: the two inference engines agree
    with torch.no_grad():
        o_cpu = U.unet_forward(weights['oracle'], held_z[:2].cpu())
    q_cpu = quality_assess_frames(o_cpu.to(dev), held_y[:2].contiguous()).cpu().numpy()
    dev_curve = {tag: float(np.max(np.abs(np.array(curves[tag]) - np.array(curves['oracle'])) / np.array(curves['oracle']))) for tag in ('fp32', 'fp16x2', 'bf16')}
    rec = {'iters': args.iters, 'patch': [4, S, S], 'train_scenes': args.train_scenes, 'eval_scenes': args.eval_scenes, 'noise': 'PGRU, SonyA7S2, _sample_params under np.random.seed(2018)',
           'sampler_replays_bit_exact': replayed, 'threads': torch.get_num_threads(), 'noisy_input_psnr_mean': float(q_in[:, 0].mean()),
           'untrained_psnr_mean': float(q_init[:, 0].mean()), 'models': rows, 'max_rel_loss_curve_deviation_vs_oracle': dev_curve,
           'oracle_weights_torch_cpu_forward_psnr': [float(v) for v in q_cpu[:, 0]], 'oracle_weights_hip_forward_psnr': [float(v) for v in base[:2, 0]],
           'loss_curves_every_25': {tag: [curves[tag][i] for i in range(0, args.iters, 25)] for tag in curves}}
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out + '.json', 'w') as f:
        json.dump(rec, f, indent=1)
    md = ['# End-metric parity: %d training iterations on synthetic scenes, PSNR on a held-out set' % args.iters, '',
          '`python tools/psnr_parity.py --iters %d` on one MI355X + its host (%d torch threads).  One 4x%dx%d patch per step, L1, Adam lr 1e-4, PGRU noise' % (
              args.iters, torch.get_num_threads(), S, S),
          'synthesised once per iteration by the HIP sampler and fed to all four trainings (%d iterations replayed through the oracle arithmetic: same bits).' % replayed,
          'Held-out set: %d scenes; noisy input PSNR %.2f dB, untrained network %.2f dB.  PSNR/SSIM as util/index.py:76-81 (eld_quality_assess), all weight sets' % (
              len(held), q_in[:, 0].mean(), q_init[:, 0].mean()),
          'scored on the fp32 HIP engine.', '',
          '| training | held-out PSNR (dB) | delta vs oracle (dB) | max per-image delta (dB) | SSIM | final loss | mean loss, last 20 | max rel. loss-curve deviation | train time (s) |',
          '|---|---|---|---|---|---|---|---|---|']
    names = {'oracle': 'torch-CPU fp32 oracle (reference step)', 'fp32': 'HIP fp32 (3 bf16 pieces, default)', 'fp16x2': 'HIP fp32, 2 fp16 pieces (opt-in)', 'bf16': 'HIP bf16 (configs[2])'}
    for tag in ('oracle', 'fp32', 'fp16x2', 'bf16'):
        r = rows[tag]
        md.append('| %s | %.4f | %+.4f | %.4f | %.5f | %.6f | %.6f | %s | %.1f |' % (
            names[tag], r['psnr_mean'], r['dpsnr_vs_oracle'], r['max_abs_dpsnr_per_image'], r['ssim_mean'], r['final_loss'], r['mean_loss_last20'],
            ('%.2e' % dev_curve[tag]) if tag in dev_curve else '-', r['train_s']))
    md += ['', 'bf16-trained weights scored with bf16 inference: %.4f dB.' % rows['bf16']['psnr_mean_bf16_inference'],
           'Oracle-trained weights, first two held-out scenes: torch-CPU forward %s dB, HIP fp32 forward %s dB.' % (
               ', '.join('%.4f' % v for v in q_cpu[:, 0]), ', '.join('%.4f' % v for v in base[:2, 0])), '']
    with open(args.out + '.md', 'w') as f:
        f.write('\n'.join(md) + '\n')
    print('\n'.join(md))
    ok = abs(rows['fp32']['dpsnr_vs_oracle']) <= 0.05
    print('fp32 |dPSNR| <= 0.05 dB:', ok)
    return 0 if ok else 1


if __name__ == '__main__':
    sys.exit(main())
