#!/usr/bin/env python3
"""BASELINE.json configs[4]: multi-camera ELD evaluation sweep at full sensor resolution on the device.

The reference's evaluation (test_ELD.py -> ELDModel.eval, models/ELD_model.py:205-247) runs per (camera, ISO, ratio) over
captured pairs that are not available here; this harness drives the SAME device path on synthetic frames of each camera's
packed resolution: noise synthesis with that camera's calibrated tables (full model PGRU, K tied to the ISO) -> U-Net
inference -> IlluminanceCorrect -> tensor2im + PSNR/SSIM (csrc/eval.hip), and reports throughput per setting.  With
random-init weights the PSNR/SSIM columns only exercise the metric path; load a checkpoint with --weights for real numbers.
Replicas only: under torchrun the settings are sharded over ranks, no data-path collective.

Round 5: the frames of a setting go through the chain `--batch` at a time (default 8, capped so that the U-Net workspace of one launch stays
below 96 GB: 8 frames for the Sony / Canon sensors, 3 for the D850) -- one sampler launch with per-image parameters, one U-Net inference, one
IlluminanceCorrect and one quality launch per batch; a single full frame per launch left the deep levels of the U-Net with too few tiles to
fill the chip (169 vs 199 TF/s fp32).  tests/test_eval_sweep_gpu.py pins the batched chain to the single-frame launches bit for bit.

  python tools/eval_sweep.py [--precision fp32|bf16] [--frames 2] [--batch 8] [--weights ckpt.pt]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np          # noqa: E402
import torch                # noqa: E402

# packed (H/2, W/2) Bayer resolutions of the ELD cameras, trimmed to a multiple of 16 (Unet.py:51-63 needs 4 poolings)
CAMERAS = {'SonyA7S2': (1424, 2128), 'NikonD850': (2752, 4128), 'CanonEOS70D': (1824, 2736), 'CanonEOS700D': (1728, 2592)}
ISOS = (800, 1600, 3200)
RATIOS = (100, 200)


def params_for(cam_tables, iso, ratio, rng):
    """System gain tied to the ISO inside the camera's calibrated [Kmin, Kmax] range (ISO 100..25600 log-linear), the other
    terms from the camera's regressions at that K (noise.py:201-225 form)."""
    from eld_amd.noise import NoiseParams
    lk0, lk1 = np.log(float(cam_tables['Kmin'])), np.log(float(cam_tables['Kmax']))
    log_K = lk0 + (lk1 - lk0) * (np.log(iso / 100.0) / np.log(256.0))
    prof = cam_tables['Profile-1']

    def reg(name):
        r = prof[name]
        return float(np.exp(rng.standard_normal() * float(r['sigma']) + float(r['slope']) * log_K + float(r['bias'])))
    i = rng.randint(len(cam_tables['G_shape']))
    return NoiseParams(float(np.exp(log_K)), reg('g_scale'), 16383 - 800, float(ratio), tl_lambda=float(cam_tables['G_shape'][i]),
                       tl_scale=reg('G_scale'), row_scale=reg('R_scale'), q_step=1.0)


def frames_per_launch(lib, H, W, want, budget_bytes=96e9):
    """Frames per launch of the chain at H x W: `want`, capped so that the U-Net workspace of the launch stays below the budget."""
    one = lib.eld_unet_workspace_bytes(1, H, W, 4, 4)
    return max(1, min(int(want), int(budget_bytes // max(one, 1))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--precision', default='fp32', choices=['fp32', 'bf16'])
    ap.add_argument('--frames', type=int, default=2, help='timed launches per setting (each of --batch frames)')
    ap.add_argument('--batch', type=int, default=8, help='frames per launch (capped by the workspace: see frames_per_launch)')
    ap.add_argument('--weights', default=None, help="checkpoint with a 'netG' state_dict (ELD_model.py:518)")
    ap.add_argument('--noise', default='PGRU')
    args = ap.parse_args()
    from eld_amd import dist as D
    world, rank, local = D.init()
    if not torch.cuda.is_available():
        raise SystemExit('eval_sweep.py needs a GPU: eld_amd has no CPU fallback')
    dev = torch.device('cuda', local % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    import eld_amd
    eld_amd.load_library()
    from eld_amd import _lib as L
    from eld_amd.metrics import illuminance_correct, quality_assess_frames
    from eld_amd.noise import load_camera_params, model_flags, sample_noise
    from eld_amd.unet import UNetSeeInDark
    torch.manual_seed(2018)
    net = UNetSeeInDark(4, 4).to(dev)
    if args.weights:
        net.load_state_dict(torch.load(args.weights, map_location=dev)['netG'])
    net.inference_precision = args.precision
    flags = model_flags(args.noise) | L.CLIP
    settings = [(c, i, r) for c in CAMERAS for i in ISOS for r in RATIOS]
    mine = settings[rank::world]
    rows = []
    for k, (cam, iso, ratio) in enumerate(mine):
        H, W = CAMERAS[cam]
        tables = load_camera_params(cam)
        rng = np.random.RandomState(1000 * k + rank)
        g = torch.Generator(device=dev).manual_seed(77 + k)
        psnr = ssim = 0.0
        t_total = 0.0
        B = frames_per_launch(L.lib(), H, W, args.batch)
        for f in range(args.frames + 1):                          # launch 0 warms up (workspace allocation)
            clean = (torch.floor(65535.0 * torch.rand(B, 4, H, W, device=dev, generator=g) ** 2.2) / 65535.0).contiguous()
            ps = [params_for(tables, iso, ratio, rng) for _ in range(B)]
            ids = [(len(settings) * f + k) * 16 + b for b in range(B)]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            noisy = sample_noise(clean, ps, flags, 2018, ids)
            with torch.no_grad():
                out = net(noisy)
            out = illuminance_correct(out, clean)
            q = quality_assess_frames(out, clean)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            if f:
                t_total += dt
                qa = q.mean(dim=0).tolist()
                psnr += qa[0]; ssim += qa[1]
            del noisy, out, clean
        n = args.frames
        rows.append({'camera': cam, 'iso': iso, 'ratio': ratio, 'packed_hw': [H, W], 'frames_per_launch': B, 'ms_per_frame': round(t_total / (n * B) * 1e3, 3),
                     'raw_mpix_s': round(4.0 * H * W * B / (t_total / n) / 1e6, 1), 'psnr': round(psnr / n, 3), 'ssim': round(ssim / n, 5)})
        if k + 1 == len(mine) or CAMERAS[mine[k + 1][0]] != (H, W):      # next setting runs another sensor shape: give this one's scratch back
            net.release_workspaces()
            torch.cuda.empty_cache()
    if world > 1:
        gathered = [None] * world
        torch.distributed.all_gather_object(gathered, rows)
        rows = [r for part in gathered for r in part]
    if rank == 0:
        tot_pix = sum(4.0 * r['packed_hw'][0] * r['packed_hw'][1] for r in rows)
        tot_t = sum(r['ms_per_frame'] for r in rows) / 1e3
        print(json.dumps({'config': 'BASELINE.json configs[4]: multi-camera eval sweep, synthetic full-resolution frames', 'precision': args.precision,
                          'weights': args.weights or 'random init (PSNR/SSIM exercise the metric path only)', 'n_gpus': world,
                          'raw_mpix_s_mean': round(tot_pix / tot_t / 1e6 * (world if world > 1 else 1), 1), 'settings': rows}))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
