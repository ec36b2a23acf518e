#!/usr/bin/env python3
"""bench.py -- the hot path on MI355X: noise synthesis + U-Net training step, raw megapixels/s.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic clean raw already resident in HBM:
  fused HIP sampler (full ELD model 'PGRU': Poisson shot + Tukey-lambda read + row + quantisation, SonyA7S2
  parameters, + clip) -> U-Net forward (fp32, exact-fp32 MFMA) -> L1 loss -> U-Net backward -> [RCCL gradient
  all-reduce] -> Adam.  Workload = BASELINE.json configs[1]: 4x1424x2128 packed raw per image, fp32; 8 frames per GPU
  (weak scaling: the per-GPU batch is fixed as N grows, so N=8 is configs[3]: 8xMI355X data parallel, global batch 64).
Prints ONE JSON line (rank 0).  value = total raw pixels of all ranks / wall time of the timed K steps
(barrier + device sync on both sides, MAX over ranks).
"""
import argparse
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np          # noqa: E402
import torch                # noqa: E402

H_FULL, W_FULL = 1424, 2128
FLOP_FWD_PER_PIX = 92288.0          # 2*MAC of the U-Net forward per raw pixel
FLOP_STEP_PER_PIX = 276300.0        # forward + backward-data + backward-weight (no bwd-data for conv1_1)
PEAK_BF16_MFMA_TF = 2500.0           # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_bf16 dense peak
PEAK_F32_MFMA_TF = 157.3            # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_HBM_GBS = 8000.0


def synth_clean(n, h, w, device, seed):
    """Clean packed raw on the LMDB uint16 grid, dark-heavy: floor(65535*U^2.2)/65535."""
    g = torch.Generator(device=device).manual_seed(seed)
    u = torch.rand(n, 4, h, w, device=device, generator=g)
    return (torch.floor(65535.0 * u ** 2.2) / 65535.0).contiguous()


def make_opt(local_rank, precision='fp32'):
    return types.SimpleNamespace(precision=precision, gpu_ids=[local_rank], isTrain=True, checkpoints_dir='/tmp/eld_amd_bench', name='bench', netG='unet',
                                 channels=4, stage_in='raw', stage_out='raw', lr=1e-4, beta1=0.9, wd=0.0, loss='l1', resume=False,
                                 no_log=True, chop=False, model='eld_model')


def load_traffic():
    """HBM bytes from the committed rocprofv3 PMC passes (profiles/traffic.json, tools/traffic_from_pmc.py); None if absent.
    The figures belong to ONE kernel set: the file records the source hash of the library that ran the PMC passes (library_src_hash[_bf16]) and a
    precision's bytes are dropped (traffic = null on the line) when the loaded library was built from other sources -- a stale constant cannot
    ride a new kernel set onto the driver line."""
    try:
        with open(os.path.join(ROOT, 'profiles', 'traffic.json')) as f:
            t = json.load(f)
    except Exception:
        return None
    from eld_amd import _lib as L
    have = L.build_src_hash()
    t['_stale'] = {}
    for key, hk in (('unet_conv_bytes_per_pass', 'library_src_hash'), ('unet_conv_bytes_per_pass_bf16', 'library_src_hash_bf16')):
        if t.get(hk) != have:
            t['_stale'][key] = 'profiles/traffic.json was measured on library src=%s, this run loaded src=%s: not reported' % (t.get(hk), have)
            t.pop(key, None)
            if key == 'unet_conv_bytes_per_pass':
                t.pop('sampler_bytes_per_pixel', None)
                t.pop('sampler_valu_lane_ops_per_pixel', None)
    return t


def timed_events(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def unet_roofline(model, B, Hh, Ww, precision, dev, traffic):
    """Roofline of the U-Net launches of one step, timed live with HIP events on the launch stream: eld_unet_forward + eld_unet_backward
    over the model's current input.  achieved = SURVEY.md 8(d)'s 276,300 FLOP per raw pixel x the pixels of one pass / that time."""
    import eld_amd
    net, x = model.netG, model.input
    dout = torch.ones(B, 4, Hh, Ww, device=dev) / (B * 4.0 * Hh * Ww)
    state = {}

    def fwd():
        state['k'] = net._engine_forward(x, save=True, bf16=precision == 'bf16')[1]

    def bwd():
        net._engine_backward(dout, state['k'], tuple(x.shape), grads=model.optimizer_G.grads)
    fwd(); bwd(); torch.cuda.synchronize()
    t_f = timed_events(fwd, 3)
    t_b = timed_events(bwd, 3)
    full_frame = (Hh, Ww) == (H_FULL, W_FULL)
    flop_step = FLOP_STEP_PER_PIX * B * 4.0 * Hh * Ww
    ach = flop_step / ((t_f + t_b) * 1e-3) / 1e12
    # fp32 step: with eld_conv_fp32_algo = 1 every fp32 product is six bf16 MFMA products (csrc/conv_x3.hip), so the pipe
    # that bounds the kernels is the bf16 MFMA and its fp32-equivalent peak is 2500 / 6; algo 0 runs on the fp32 MFMA.
    x3 = precision == 'fp32' and eld_amd.load_library().eld_conv_fp32_algo(-1) == 1
    peak = PEAK_BF16_MFMA_TF if precision == 'bf16' else (PEAK_BF16_MFMA_TF / 6.0 if x3 else PEAK_F32_MFMA_TF)
    key = 'unet_conv_bytes_per_pass' if precision == 'fp32' else 'unet_conv_bytes_per_pass_bf16'
    tr = None
    if traffic and full_frame and key in traffic:
        tr = round(traffic[key] * B / traffic.get('frames_per_pass', 1))
    if precision == 'bf16':
        kern = 'conv_bfd / conv_bfw / conv_bfs / conv_bfg kernels fwd + bwd-data, wgrad8_kernel<bf16>: bf16 operands, v_mfma_f32_32x32x16_bf16, fp32 accumulate'
        note = 'bf16 dense MFMA peak 2500 TFLOP/s (MI355X_MICROARCH.md)'
    elif x3:
        kern = 'conv_x3d_kernel / conv_x3_kernel fwd + bwd-data, wgrad8_kernel: fp32 operands as 3 bf16 pieces, 6 x v_mfma_f32_32x32x16_bf16 per k-block'
        note = ('bf16 dense MFMA peak 2500 TFLOP/s / 6 piece products per fp32 product: the work runs on the bf16 pipe; against the native fp32 MFMA peak '
                '(157.3 TFLOP/s) the same figure is %.2fx' % (ach / PEAK_F32_MFMA_TF))
    else:
        kern, note = 'conv_igemm_kernel fwd/bwd-data + wgrad_kernel', 'dense fp32 MFMA peak'
    return {'bound': 'mfma', 'kernel': 'U-Net convolution launches of one step (%s), timed as eld_unet_forward + eld_unet_backward' % kern,
            'achieved': round(ach, 2), 'peak': round(peak, 1), 'unit': 'TFLOP/s', 'frac': round(ach / peak, 4), 'peak_note': note,
            'traffic': tr, 'traffic_source': ('profiles/traffic.json (committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command on a library with '
                                              'the same source hash, scaled to the frames of this run; not re-measured in this run)' if tr is not None else
                                              ((traffic or {}).get('_stale', {}).get(key))),
            'fwd_ms': round(t_f, 3), 'bwd_ms': round(t_b, 3),
            'fwd_tflops': round(FLOP_FWD_PER_PIX * B * 4.0 * Hh * Ww / (t_f * 1e-3) / 1e12, 2)}, peak, x3


def _cpu_sampler_worker(args):
    """One forked DataLoader-style worker: the NumPy sampler port on one image (its own NumPy stream, like worker_init_fn)."""
    y, model, seed = args
    from oracle import noise_ref as O
    rs = np.random.RandomState(seed)
    p = O.Params(K=2.288, g_scale=6.451, ratio=208.98, tl_lambda=-0.14285714, tl_scale=3.3, row_scale=0.9)
    t0 = time.time()
    O.noise_numpy_full(y, p, O.model_flags(model) | O.CLIP, rng=rs)
    return time.time() - t0


def cpu_baseline(h, w, seed=2018):
    """The reference-style CPU path timed on THIS box's host cores, bounded to ~20-30 s of CPU work (kind "port": the oracle's
    NumPy / torch-CPU restatement -- /root/reference does not exist on the GPU box):
      * sampler: NumPy port, single thread like noise.py (np.random is single-threaded), on ONE full 4 x h x w image, for the
        full model 'PGRU' and for the reference's own 'Pg' (noise.py:158-166); plus the reference's real deployment, 8 forked
        DataLoader workers (--nThreads 8, base_option.py:26), one 'Pg' image each, as an aggregate rate;
      * U-Net: one torch-CPU fp32 training step (forward + L1 + backward + Adam) on ONE full 4 x h x w frame, with the fastest thread count of a
        sweep over {32, 64, 128, all host cores} made on a 512 x 512 crop (recorded on the line).
    value = combined per-pixel rate of (PGRU sampler, 1 thread) + (U-Net step)."""
    import multiprocessing as mp
    from oracle import noise_ref as O
    from oracle import unet_ref as U
    t_begin = time.time()
    rs = np.random.RandomState(seed)
    y = (np.floor(65535.0 * rs.uniform(size=(4, h, w)) ** 2.2) / 65535.0).astype(np.float32)
    t_full = min(_cpu_sampler_worker((y, 'PGRU', seed + i)) for i in range(2))
    t_pg = _cpu_sampler_worker((y, 'Pg', seed))
    workers = 8
    t0 = time.time()
    with mp.get_context('fork').Pool(workers) as pool:
        pool.map(_cpu_sampler_worker, [(y, 'Pg', seed + 100 + i) for i in range(workers)])
    t_pool = time.time() - t0
    sd = U.seeded_state_dict(4, 4, seed=seed)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    opt = torch.optim.Adam(list(params.values()), lr=1e-4)

    def cpu_step(x):
        t0 = time.time()
        opt.zero_grad()
        loss = torch.nn.functional.l1_loss(U.unet_forward(params, x), x)
        loss.backward()
        opt.step()
        loss.item()                      # the reference reads loss.item() every iteration (ELD_model.py:480)
        return time.time() - t0
    # torch-CPU on the host cores of THIS box.  The thread count is swept once on the reference's own training shape (one 4x512x512
    # crop: 32, 64, 128, all cores; each setting one warm-up step + min of 2) and the fastest setting runs the full frame -- torch-CPU convolutions
    # stop scaling, and then lose, well before 256 threads, so "all cores" is measured rather than assumed.
    ncpu = os.cpu_count() or 1
    x512 = torch.from_numpy(y[None, :, :min(512, h), :min(512, w)].copy())
    npx512 = 4.0 * min(512, h) * min(512, w)
    sweep = {}
    for nt in sorted(set(min(t, ncpu) for t in (32, 64, 128, ncpu))):
        torch.set_num_threads(nt)
        cpu_step(x512)                                       # warm-up (thread pool, oneDNN primitives)
        sweep[nt] = min(cpu_step(x512) for _ in range(2))
    cores = min(sweep, key=sweep.get)
    torch.set_num_threads(cores)
    t_512 = sweep[cores]                                     # the reference's own training shape (BASELINE.md sec. 2 quotes this one)
    xf = torch.from_numpy(y[None].copy())
    t_unet = cpu_step(xf)                                    # full frame (a second run only while the whole baseline stays near 30 s)
    if time.time() - t_begin < 22.0:
        t_unet = min(t_unet, cpu_step(xf))
    npx = 4.0 * h * w
    per_pix = t_full / npx + t_unet / npx
    return {'value': round(1e-6 / per_pix, 4), 'unit': 'raw MPix/s', 'cores': cores, 'kind': 'port',
            'sample': 'one 4x%dx%d frame: NumPy sampler port PGRU %.2f s (1 thread, min of 2), Pg %.2f s (1 thread), 8 forked workers x 1 Pg image '
                      '%.2f s wall; torch-CPU fp32 U-Net train step (fwd, L1, bwd, Adam, loss.item()) on the full frame %.2f s (%d threads = the fastest of the '
                      'thread sweep, after 512x512 warm-up steps) and on one 4x512x512 crop %.3f s (min of 2); value = PGRU sampler + full-frame U-Net step per pixel' % (
                          h, w, t_full, t_pg, t_pool, t_unet, cores, t_512),
            'sampler_mpix_s': round(npx / t_full / 1e6, 3), 'sampler_Pg_mpix_s': round(npx / t_pg / 1e6, 3),
            'sampler_Pg_8workers_mpix_s': round(workers * npx / t_pool / 1e6, 3), 'unet_step_mpix_s': round(npx / t_unet / 1e6, 4),
            'unet_step_512_mpix_s': round(npx512 / t_512 / 1e6, 4), 'unet_step_512_s': round(t_512, 3),
            'threads_sweep_512_mpix_s': {str(k): round(npx512 / v / 1e6, 4) for k, v in sorted(sweep.items())},
            'os_cpu_count': ncpu, 'seconds': round(time.time() - t_begin, 1),
            # kind "port": what the port costs relative to the reference itself, measured where both run (the build container, 8 cores, identical inputs,
            # sampler outputs asserted bit-equal): profiles/r04_cpu_port_vs_reference.md
            'port_over_reference_time': {'sampler_Pg_full_frame': 1.11, 'sampler_g_full_frame': 1.22, 'unet_step_512': 0.93,
                                         'source': 'profiles/r04_cpu_port_vs_reference.md (tools/cpu_port_vs_reference.py, build container)',
                                         'note': 'the port is 7-22 % slower than the reference on the sampler (it also records its variates) and 7 % faster on '
                                                 'the U-Net step, which dominates value: the stated baseline is the reference\'s cost to within that'}}


def eval_sweep_leg(dev, precision, frames=3, batch=8):
    """BASELINE.json configs[4] on the driver line: one setting of the reference's evaluation sweep (test_ELD.py:18-52 -> ELDModel.eval,
    ELD_model.py:203-307) at sensor resolution, on the device: SonyA7S2 packed frames, ISO 1600, ratio 100 -- noise synthesis with the camera's
    tables -> U-Net inference -> IlluminanceCorrect -> tensor2im + PSNR + SSIM (csrc/eval.hip), HIP events per stage, `batch` frames per launch
    (round 5: one frame per launch left the deep U-Net levels with too few tiles; tests/test_eval_sweep_gpu.py pins the batched chain to single
    frames bit for bit).  tools/eval_sweep.py runs all 4 cameras x 3 ISOs x 2 ratios (profiles/r05_eval_sweep_*.json)."""
    import importlib.util
    import eld_amd
    from eld_amd import _lib as L
    from eld_amd.metrics import illuminance_correct, quality_assess_frames
    from eld_amd.noise import load_camera_params, model_flags, sample_noise
    from eld_amd.unet import UNetSeeInDark
    spec = importlib.util.spec_from_file_location('eval_sweep', os.path.join(ROOT, 'tools', 'eval_sweep.py'))
    es = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(es)
    cam, iso, ratio = 'SonyA7S2', 1600, 100
    H, W = es.CAMERAS[cam]
    B = es.frames_per_launch(eld_amd.load_library(), H, W, batch, budget_bytes=64e9)
    torch.manual_seed(2018)
    net = UNetSeeInDark(4, 4).to(dev)
    net.inference_precision = precision
    flags = model_flags('PGRU') | L.CLIP
    rng = np.random.RandomState(5)
    g = torch.Generator(device=dev).manual_seed(77)
    tables = load_camera_params(cam)
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(frames + 1)]
    q = None
    for f in range(frames + 1):                               # launch 0 warms up (workspace allocation)
        clean = (torch.floor(65535.0 * torch.rand(B, 4, H, W, device=dev, generator=g) ** 2.2) / 65535.0).contiguous()
        ps = [es.params_for(tables, iso, ratio, rng) for _ in range(B)]
        ev[f][0].record()
        noisy = sample_noise(clean, ps, flags, 2018, [16 * f + b for b in range(B)])
        ev[f][1].record()
        with torch.no_grad():
            out = net(noisy)
        ev[f][2].record()
        out = illuminance_correct(out, clean)
        ev[f][3].record()
        q = quality_assess_frames(out, clean)
        ev[f][4].record()
    torch.cuda.synchronize()
    st = [min(ev[f][i].elapsed_time(ev[f][i + 1]) for f in range(1, frames + 1)) for i in range(4)]      # ms per launch of B frames
    tot = min(ev[f][0].elapsed_time(ev[f][4]) for f in range(1, frames + 1))
    npx = 4.0 * H * W * B
    # the two evaluation kernels are single passes over the frames: IlluminanceCorrect reads predict + source for the two dot products, then
    # reads predict and writes the output (16 B per element); the quality kernel reads both images once (8 B per element)
    gb_corr, gb_q = 16.0 * npx / (st[2] * 1e-3) / 1e9, 8.0 * npx / (st[3] * 1e-3) / 1e9
    psnr, ssim = q.mean(dim=0).tolist()
    net.release_workspaces()
    # the reference evaluates with batch size 1 (test_ELD.py:44-52 -> DataLoader(batch_size=1)): the like-for-like single-frame figure rides beside
    # the batched one (same chain, one frame per launch), so that neither is mistaken for the other
    single = None
    if B > 1:
        r1 = eval_sweep_leg(dev, precision, frames=frames, batch=1)
        single = {'value': r1['value'], 'unit': 'raw MPix/s', 'ms_per_frame': r1['ms_per_frame'], 'frames_per_launch': 1,
                  'unet_inference_tflops': r1['unet_inference_tflops'], 'stage_ms_per_frame': r1['stage_ms_per_frame'],
                  'note': "the reference's evaluation batch size (1 frame per launch); `value` of this leg is the BATCHED rate (%d frames per launch)" % B}
    return {'value': round(npx / (tot * 1e-3) / 1e6, 1), 'unit': 'raw MPix/s', 'ms_per_frame': round(tot / B, 3), 'frames_per_launch': B, 'launches': frames,
            'batched': B > 1, 'single_frame_launches': single,
            'dtype': 'f32' if precision == 'fp32' else 'bf16',
            'config': {'workload': 'BASELINE.json configs[4], one setting: %s packed %dx%d, ISO %d, ratio x%d, %d synthetic frames per launch; synth (PGRU) -> U-Net '
                                   'inference -> IlluminanceCorrect -> tensor2im + PSNR + SSIM, all on the device' % (cam, H, W, iso, ratio, B)},
            'stage_ms_per_frame': {'sampler': round(st[0] / B, 4), 'unet_inference': round(st[1] / B, 3), 'illuminance_correct': round(st[2] / B, 4),
                                   'quality_assess': round(st[3] / B, 4)},
            'unet_inference_tflops': round(FLOP_FWD_PER_PIX * npx / (st[1] * 1e-3) / 1e12, 1),
            'eval_kernels_hbm': {'illuminance_correct': {'algorithmic_bytes': 16 * int(npx), 'achieved_GBps': round(gb_corr, 1), 'frac': round(gb_corr / PEAK_HBM_GBS, 4)},
                                 'quality_assess': {'algorithmic_bytes': 8 * int(npx), 'achieved_GBps': round(gb_q, 1), 'frac': round(gb_q / PEAK_HBM_GBS, 4)}},
            'psnr_ssim_random_init': [round(psnr, 3), round(ssim, 5)]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=8, help='full frames per GPU (8 => N=8 GPUs is BASELINE config 3/4: global batch 64)')
    ap.add_argument('--height', type=int, default=H_FULL)
    ap.add_argument('--width', type=int, default=W_FULL)
    ap.add_argument('--noise', default='PGRU')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-alt', action='store_true', help='skip the alt_fp16x2_products leg (profiling runs: only the default kernels in the trace)')
    ap.add_argument('--prefetch', type=int, default=1, help='1: synthesise batch i+1 on the side stream during step i (Engine.train default); 0: serial')
    ap.add_argument('--precision', default='fp32', choices=['fp32', 'bf16'], help="U-Net precision; the contract's metric is quoted on fp32 (BASELINE configs[1]); bf16 = configs[2]")
    args = ap.parse_args()

    from eld_amd import dist as D
    world, rank, local = D.init()
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run --nproc-per-node %d)' % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: eld_amd has no CPU fallback')
    local = local % torch.cuda.device_count()       # (ranks may share a GPU only in the gloo smoke test of the N>1 path)
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    import eld_amd
    eld_amd.load_library()
    from eld_amd import _lib as L
    from eld_amd.model import ELDModel
    from eld_amd.noise import NoiseModel, NoiseParams, sample_noise, model_flags

    B, Hh, Ww = args.batch, args.height, args.width
    np.random.seed(2018)
    torch.manual_seed(2018)
    import io
    import contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        nm = NoiseModel(model=args.noise, include=4)               # SonyA7S2
    model = ELDModel()
    model.initialize(make_opt(local, args.precision))
    model.set_noise_model(nm)
    clean = synth_clean(B, Hh, Ww, dev, seed=1234 + rank)           # resident in HBM before the timed region
    total_steps = args.steps + args.warmup
    plists = [[nm._sample_params() for _ in range(B)] for _ in range(total_steps)]     # _sample_params semantics, host side

    batches = {}

    def batch(i):
        if i not in batches:
            ids = [(i * world * B) + rank + world * k for k in range(B)]      # global sample indices
            batches[i] = {'target': clean, 'params': plists[i % len(plists)], 'sample_ids': ids}
        return batches[i]

    def step(i):
        # Engine.train's loop body (eld_amd/engine.py): batch i (already synthesised on the side stream when the previous step prefetched it),
        # then batch i+1's synthesis is started beside this step's U-Net kernels -- one sampler launch per step either way
        model.set_input(batch(i), 'train')
        batches.pop(i, None)
        if args.prefetch:
            model.prefetch_input(batch(i + 1), 'train')
        model.optimize_parameters()
        return model.get_current_errors()['Pixel']           # loss.item(): the reference's per-iteration device sync (ELD_model.py:480)

    for i in range(args.warmup):
        step(i)
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.warmup, total_steps):
        loss = step(i)
    torch.cuda.synchronize()
    dt_local = time.perf_counter() - t0
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    per_rank, exchange = None, None
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
        allr = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(world)]
        torch.distributed.all_gather(allr, torch.tensor([dt_local / args.steps * 1e3], dtype=torch.float64, device=dev))
        per_rank = [round(float(v.item()), 3) for v in allr]
        # exposed cost of the gradient exchange: the same steps with the all-reduce switched off (replicas diverge: done last)
        model.exchange = False
        step(total_steps); torch.cuda.synchronize(); torch.distributed.barrier()
        t1 = time.perf_counter()
        for i in range(args.steps):
            step(total_steps + 1 + i)
        torch.cuda.synchronize()
        t_noex = torch.tensor([(time.perf_counter() - t1) / args.steps * 1e3], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t_noex, op=torch.distributed.ReduceOp.MAX)
        model.exchange = True
        nbytes = 4 * model.optimizer_G.grads.numel()
        exchange = {'bytes': nbytes, 'buckets': (model._buckets.n if model._buckets is not None else 1), 'ms_per_step_without_exchange': round(float(t_noex.item()), 3),
                    'exposed_ms': round(max(0.0, dt / args.steps * 1e3 - float(t_noex.item())), 3),
                    'ring_floor_ms': round(2.0 * (world - 1) / world * nbytes / 153e9 * 1e3, 3),
                    'note': 'exposed = step time with the bucketed RCCL all-reduce (overlapped with the backward) minus the same step without it; '
                            'ring_floor = 2(N-1)/N x bytes over one ~153 GB/s xGMI link'}

    pix_per_step = world * B * 4.0 * Hh * Ww
    res = {
        'metric': 'raw megapixels/sec (noise-synth + U-Net step)', 'value': round(pix_per_step * args.steps / dt / 1e6, 3),
        'unit': 'raw MPix/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f32' if args.precision == 'fp32' else 'bf16', 'data': 'synthetic',
        'config': {'workload': 'BASELINE.json configs[%d]: full ELD noise model (%s, SonyA7S2 params) on 4x%dx%d packed raw + U-Net %s '
                               'train step (fwd, L1, bwd, Adam), %d frames per GPU' % (1 if args.precision == 'fp32' else 2, args.noise, Hh, Ww, args.precision, B),
                   'images_per_gpu': B, 'global_batch': B * world, 'parallelism': 'dp%d' % world, 'final_loss': loss, 'loss_item_sync_per_step': True,
                   'fp32_products': (None if args.precision != 'fp32' else
                                     {0: 'v_mfma_f32_32x32x2_f32', 1: 'fp32 operands cut exactly into 3 bf16 pieces, 6 bf16 MFMA products per k-block, fp32 accumulate '
                                      '(each product exact to below fp32 rounding; eld_conv_fp32_algo 1)', 2: '2 fp16 pieces per operand (22-bit products), fp32 accumulate'}
                                     [eld_amd.load_library().eld_conv_fp32_algo(-1)])},
    }

    if per_rank is not None:
        res['per_rank_ms_per_step'] = per_rank
        res['allreduce'] = exchange
    if rank == 0:
        # ---- roofline of the dominant kernels, measured live with HIP events on the launch stream ----------------------
        traffic = load_traffic()
        res['roofline'], peak, x3 = unet_roofline(model, B, Hh, Ww, args.precision, dev, traffic)
        # one launch of the dominant kernel, timed live: conv7_1's forward (256 -> 128 channels at 1/4 resolution, the step's
        # median 3x3 layer) through the single-layer entry point -- conv_x3d_kernel<128,2,8> (or conv_igemm_kernel<float,0,64,2>)
        if args.precision == 'fp32' and Hh % 4 == 0 and Ww % 4 == 0:
            lib = eld_amd.load_library()
            h4, w4, ci, co = Hh // 4, Ww // 4, 256, 128
            xl = torch.randn(B, h4, w4, ci, device=dev)
            wl = torch.randn(co, ci, 3, 3, device=dev) * 0.02
            bl = torch.zeros(co, device=dev)
            ol = torch.empty(B, h4, w4, co, device=dev)
            wsl = torch.empty(lib.eld_layer_workspace_bytes(B, h4, w4, ci, co), dtype=torch.uint8, device=dev)

            def one():
                L.check(lib.eld_conv3x3_forward(L.dptr(xl), ci, None, 0, L.dptr(wl), L.dptr(bl), L.dptr(ol), B, h4, w4, co, 1, L.dptr(wsl), wsl.numel(),
                                                L.cur_stream()), 'eld_conv3x3_forward')
            one(); torch.cuda.synchronize()
            t_l = timed_events(one, 5)
            fl = 2.0 * B * h4 * w4 * co * ci * 9
            res['roofline']['per_launch'] = {'kernel': ('conv_x3d_kernel<128, 2, 8>' if x3 else 'conv_igemm_kernel<float, 0, 64, 2>') +
                                             ' (+ its 50 us weight-pack launch): conv7_1 forward, %d x %dx%d, 256 -> 128 channels' % (B, h4, w4),
                                             'algorithmic_gflop': round(fl / 1e9, 1), 'ms': round(t_l, 4), 'achieved': round(fl / (t_l * 1e-3) / 1e12, 2),
                                             'frac': round(fl / (t_l * 1e-3) / 1e12 / peak, 4)}
            del xl, wl, ol, wsl
        # the same step with eld_conv_fp32_algo(2) (two fp16 pieces per operand: 22-bit products, fp32 accumulation), reported
        # beside the default; `value` above stays on the exact three-piece split
        if args.precision == 'fp32' and world == 1 and x3 and not args.no_alt:
            lib2 = eld_amd.load_library()
            lib2.eld_conv_fp32_algo(2)
            try:
                step(total_steps); torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(3):
                    step(total_steps + 1 + i)
                torch.cuda.synchronize()
                dt2 = (time.perf_counter() - t0) / 3
                res['alt_fp16x2_products'] = {'value': round(pix_per_step / dt2 / 1e6, 3), 'unit': 'raw MPix/s', 'ms_per_step': round(dt2 * 1e3, 3),
                                              'note': 'eld_conv_fp32_algo(2): operands as 2 fp16 pieces (22 significant bits), 3 MFMA products, fp32 accumulate; '
                                                      'same parity tests and tolerances as the default; not the headline because a product is good to 2^-22, not 2^-24'}
            finally:
                lib2.eld_conv_fp32_algo(1)
        # the sampler's duration INSIDE the step (right behind the previous step's Adam: clocks and caches as the step leaves them), beside
        # the standalone launch timed below
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(4)]
        model.exchange = False                      # rank 0 is alone in this block: no collective may be issued here (N > 1: the replicas are done training)
        for i, (e0, e1) in enumerate(evs):          # no host sync inside: the host runs ahead, so e0 -> e1 spans only the queued synthesis work
            ids = [((total_steps + 8 + i) * world * B) + rank + world * k for k in range(B)]
            e0.record()
            model.set_input({'target': clean, 'params': plists[i % len(plists)], 'sample_ids': ids}, 'train')
            e1.record()
            model.optimize_parameters()
        torch.cuda.synchronize()
        t_in = [e0.elapsed_time(e1) for e0, e1 in evs[1:]]
        sampler_in_step_ms = min(t_in)
        # ... and what of it the step still waits for when the launch was prefetched on the synthesis stream during the previous step
        # (events on the launch stream around set_input: only the wait for the side stream's event is left there), beside the same steps
        # with the prefetch switched off (same box, same process)
        sampler_exposed_ms, serial_ms, prefetch_ms = None, None, None
        if args.prefetch:
            base = total_steps + 40
            evs2 = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(4)]
            model.set_input(batch(base), 'train'); model.prefetch_input(batch(base + 1), 'train'); model.optimize_parameters()
            for i, (e0, e1) in enumerate(evs2):
                e0.record()
                model.set_input(batch(base + 1 + i), 'train')
                e1.record()
                model.prefetch_input(batch(base + 2 + i), 'train')
                model.optimize_parameters()
            torch.cuda.synchronize()
            sampler_exposed_ms = min(e0.elapsed_time(e1) for e0, e1 in evs2[1:])
            if world == 1 and not args.no_alt:
                def timed_steps(first, n, pf):
                    keep, args.prefetch = args.prefetch, pf
                    try:
                        step(first); torch.cuda.synchronize()
                        t0_ = time.perf_counter()
                        for k in range(n):
                            step(first + 1 + k)
                        torch.cuda.synchronize()
                        return (time.perf_counter() - t0_) / n * 1e3
                    finally:
                        args.prefetch = keep
                serial_ms = timed_steps(total_steps + 60, 5, 0)
                prefetch_ms = timed_steps(total_steps + 70, 5, 1)
        model.exchange = True
        # BASELINE.json configs[2] (bf16 U-Net with MFMA convs, batch 8) beside the headline: the same step with the bf16 engine, its own
        # roofline against the 2.5 PF/s bf16 peak.  `value` above stays on configs[1] (fp32).
        if args.precision == 'fp32' and world == 1 and not args.no_alt:
            model_b = ELDModel()
            model_b.initialize(make_opt(local, 'bf16'))
            model_b.set_noise_model(nm)

            def step_b(i):
                ids = [(i * world * B) + rank + world * k for k in range(B)]
                model_b.set_input({'target': clean, 'params': plists[i % len(plists)], 'sample_ids': ids}, 'train')
                model_b.optimize_parameters()
                return model_b.get_current_errors()['Pixel']
            for i in range(2):
                step_b(i)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            nb16 = 5
            for i in range(nb16):
                loss_b = step_b(2 + i)
            torch.cuda.synchronize()
            dtb = (time.perf_counter() - t0) / nb16
            rb, _, _ = unet_roofline(model_b, B, Hh, Ww, 'bf16', dev, traffic)
            res['alt_bf16'] = {'value': round(pix_per_step / dtb / 1e6, 3), 'unit': 'raw MPix/s', 'ms_per_step': round(dtb * 1e3, 3), 'steps': nb16, 'warmup': 2,
                               'dtype': 'bf16', 'final_loss': loss_b, 'roofline': rb,
                               'config': {'workload': 'BASELINE.json configs[2]: the same noise model and frames, bf16 U-Net train step (bf16 activations / '
                                                      'gradients / packed weights on v_mfma_f32_32x32x16_bf16, fp32 accumulation, master weights, '
                                                      'parameter gradients and Adam), %d frames per GPU' % B}}
            del model_b
            torch.cuda.empty_cache()
        # the same fp32 step with the reference's OWN noise model 'Pg' (noise.py:158-166: the one model string whose arithmetic is pinned by
        # reference-minted vectors) beside the full model of the headline
        if args.precision == 'fp32' and world == 1 and not args.no_alt and args.noise != 'Pg':
            with contextlib.redirect_stdout(io.StringIO()):
                nm_pg = NoiseModel(model='Pg', include=4)
            model.set_noise_model(nm_pg)
            step(total_steps + 20); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(3):
                step(total_steps + 21 + i)
            torch.cuda.synchronize()
            dtp = (time.perf_counter() - t0) / 3
            model.set_noise_model(nm)
            res['alt_Pg_step'] = {'value': round(pix_per_step / dtp / 1e6, 3), 'unit': 'raw MPix/s', 'ms_per_step': round(dtp * 1e3, 3), 'steps': 3,
                                  'note': "the headline step with NoiseModel(model='Pg') (Poisson shot + Gaussian read, the reference's own model) instead of PGRU"}
        # BASELINE.json configs[4]: one setting of the evaluation sweep at sensor resolution, both precisions
        if world == 1 and not args.no_alt:
            res['alt_eval_sweep'] = {'fp32': eval_sweep_leg(dev, 'fp32'), 'bf16': eval_sweep_leg(dev, 'bf16')}
            torch.cuda.empty_cache()
        # sampler alone (HBM-bound: 8 B per raw pixel), batch of 8 resident images
        nb = 8
        yb = synth_clean(nb, Hh, Ww, dev, seed=99)
        zb = torch.empty_like(yb)
        pl = [NoiseParams(2.288, 6.451, 15583, 208.98, tl_lambda=-0.14285714, tl_scale=3.3, row_scale=0.9)] * nb
        fl = model_flags(args.noise) | L.CLIP

        def samp():
            sample_noise(yb, pl, fl, 2018, list(range(nb)), out=zb)
        samp(); torch.cuda.synchronize()
        t_s = timed_events(samp, 10)
        gbs = 8.0 * yb.numel() / (t_s * 1e-3) / 1e9
        res['roofline_sampler'] = {'bound': 'hbm', 'kernel': 'noise_kernel (%s+clip), %d images per launch, K=2.288 ratio=208.98' % (args.noise, nb),
                                   'achieved': round(gbs, 1), 'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': round(gbs / PEAK_HBM_GBS, 4),
                                   'traffic': (round(traffic['sampler_bytes_per_pixel'] * yb.numel()) if traffic and 'sampler_bytes_per_pixel' in traffic else None),
                                   'traffic_source': 'profiles/traffic.json (committed PMC pass; not re-measured in this run)',
                                   'algorithmic_bytes': 8 * yb.numel(), 'ms_per_launch': round(t_s, 4), 'mpix_s': round(yb.numel() / (t_s * 1e-3) / 1e6, 1),
                                   'in_step_ms': round(sampler_in_step_ms, 4),
                                   'exposed_ms': (round(sampler_exposed_ms, 4) if sampler_exposed_ms is not None else None),
                                   'exposed_note': 'what the launch stream still waits for when the launch was issued on the synthesis stream during the previous '
                                                   'step (ELDModel.prefetch_input, the default of Engine.train and of this bench); step time with / without that '
                                                   'prefetch, 5 steps each, same process: %s / %s ms' % (prefetch_ms and round(prefetch_ms, 3), serial_ms and round(serial_ms, 3)),
                                   'in_step_note': 'the same kernel on the step\'s own %d frames, HIP events around the synthesis call inside a full step '
                                                   '(clock and caches as the preceding Adam / MFMA kernels leave them); in-step frac %.4f' % (
                                                       B, 8.0 * B * 4 * Hh * Ww / (sampler_in_step_ms * 1e-3) / 1e9 / PEAK_HBM_GBS)}
        # The full model is bound by VALU issue, not bytes: the fraction of the roofline it actually sits on.  Issued lane-operations per pixel from
        # the committed SQ_INSTS_VALU pass (profiles/traffic.json, same source-hash gate as the byte counts) x this run's pixel rate, against 1024 SIMDs x 32 lanes
        # x 2.4 GHz; quarter-rate instructions (32 x 32 multiplies, transcendentals) count once here, so the SIMDs' busy fraction is higher still.
        if traffic and 'sampler_valu_lane_ops_per_pixel' in traffic:
            lops = float(traffic['sampler_valu_lane_ops_per_pixel'])
            ach = lops * yb.numel() / (t_s * 1e-3) / 1e12
            peak_v = 1024 * 32 * 2.4e9 / 1e12
            res['roofline_sampler']['valu'] = {'bound': 'valu', 'lane_ops_per_pixel': round(lops, 1), 'achieved': round(ach, 2), 'peak': round(peak_v, 2),
                                               'unit': 'T lane-ops/s', 'frac': round(ach / peak_v, 4),
                                               'source': 'profiles/traffic.json: rocprofv3 --pmc SQ_INSTS_VALU of this command on a library with the same source hash; '
                                                         'full-rate issue peak = 256 CUs x 4 SIMD-32 x 2.4 GHz'}
        # the other model strings of the reference (noise.py:158-166: 'Pg', 'pg', 'g'), same launch shape: achieved GB/s and fraction per model
        per_model = {}
        for ms_ in ('Pg', 'pg', 'g'):
            flm = model_flags(ms_) | L.CLIP

            def samp_m():
                sample_noise(yb, pl, flm, 2018, list(range(nb)), out=zb)
            samp_m(); torch.cuda.synchronize()
            tm = timed_events(samp_m, 10)
            gm = 8.0 * yb.numel() / (tm * 1e-3) / 1e9
            per_model[ms_] = {'ms_per_launch': round(tm, 4), 'achieved': round(gm, 1), 'frac': round(gm / PEAK_HBM_GBS, 4)}
        res['roofline_sampler']['models'] = per_model
        del yb, zb
        if world == 1 and not args.no_cpu_baseline:
            res['cpu_baseline'] = cpu_baseline(Hh, Ww)
        print(json.dumps(res))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
