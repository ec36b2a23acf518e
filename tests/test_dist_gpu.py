"""RCCL path of the data-parallel step (new functionality -- the reference is single-device, models/ELD_model.py:
187-190).  One process per GPU, torch.distributed backend "nccl" (= RCCL over xGMI on ROCm).  Needs >= 2 GPUs: on a 1-GPU box
every test here skips (the same logic runs on gloo in tests/test_dist_cpu.py and tests/test_model_gpu.py).

Checked for world = 2 and, when the node has them, 4 and 8 ranks:
  * replicas are identical after the rank-0 broadcast and stay identical after training steps;
  * N ranks x (B/N images each) with the bucketed all-reduce overlapped with the backward == one process stepping on the
    B-image global batch (loss and weights);
  * the sampler's rank-strided global sample ids give the same noisy batch for every world size.
"""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')

from test_model_gpu import batch, make_opt, new_model      # noqa: E402


def _ngpu():
    return torch.cuda.device_count() if torch.cuda.is_available() else 0


def _worker(rank, world, port, tmp, q, steps, gbatch):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      ELD_DIST_BACKEND='nccl', HSA_ENABLE_IPC_MODE_LEGACY='0')
    import contextlib
    import io
    import torch as T
    from eld_amd import dist as D
    D.init()
    assert T.distributed.get_backend() == 'nccl' and T.cuda.current_device() == rank
    T.manual_seed(2018 + 7 * rank)                  # replicas start different on purpose: rank 0's weights are broadcast
    import eld_amd.noise as noise
    from eld_amd.model import ELDModel
    with contextlib.redirect_stdout(io.StringIO()):
        nm = noise.NoiseModel(model='PGRU', include=4)
    m = ELDModel()
    m.initialize(make_opt(os.path.join(tmp, 'r%d' % rank), gpu_ids=[rank]))
    first = m.netG.flat_params.detach().cpu().numpy().copy()
    p = noise.NoiseParams(2.288, 6.451, 15583, 208.98, tl_lambda=-0.14285714, tl_scale=3.3, row_scale=0.9)
    noisy = None
    for it in range(steps):
        x, t = batch(shape=(gbatch, 4, 32, 48), seed=it)
        if it == 0:                                 # on-device synthesis with the default (rank-strided, global) sample ids
            m.set_input({'target': t[rank::world], 'params': [p] * (gbatch // world)}, 'train')
            noisy = m.input.cpu().numpy().copy()
        m.set_input({'input': x[rank::world], 'target': t[rank::world]}, 'train')
        m.optimize_parameters()
        loss = m.get_current_errors()['Pixel']
    q.put((rank, loss, first, m.netG.flat_params.detach().cpu().numpy().copy(), m._buckets is not None, noisy))
    T.distributed.barrier()
    T.distributed.destroy_process_group()


@pytest.mark.parametrize('world', [2, 4, 8])
def test_rccl_data_parallel_equals_global_batch(eld_lib, tmp_path, world):
    if _ngpu() < world:
        pytest.skip('needs %d GPUs, this box has %d' % (world, _ngpu()))
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    steps, gbatch = 2, 8
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, str(tmp_path), q, steps, gbatch)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    # single process, global batch, same initial weights as rank 0
    import contextlib
    import io
    import eld_amd.noise as noise
    with contextlib.redirect_stdout(io.StringIO()):
        noise.NoiseModel(model='PGRU', include=4)
    m = new_model(tmp_path)
    first = m.netG.flat_params.detach().cpu().numpy().copy()
    pz = noise.NoiseParams(2.288, 6.451, 15583, 208.98, tl_lambda=-0.14285714, tl_scale=3.3, row_scale=0.9)
    for it in range(steps):
        x, t = batch(shape=(gbatch, 4, 32, 48), seed=it)
        if it == 0:
            m.set_input({'target': t, 'params': [pz] * gbatch}, 'train')
            noisy = m.input.cpu().numpy().copy()
        m.set_input({'input': x, 'target': t}, 'train')
        m.optimize_parameters()
        loss = m.get_current_errors()['Pixel']
    ref = m.netG.flat_params.detach().cpu().numpy()
    for r in res:
        assert r[4]                                                    # bucketed, overlapped exchange ran
        assert np.array_equal(r[2], first)                             # broadcast: every replica starts from rank 0's weights
        assert np.array_equal(r[3], res[0][3])                         # ... and they stay identical
        assert abs(r[1] - loss) < 1e-6
        assert np.array_equal(r[5], noisy[r[0]::world])                # world-size-invariant noise (global sample ids)
    assert np.abs(res[0][3] - ref).max() < 3e-5


def test_bench_multi_gpu_line(eld_lib, tmp_path):
    """bench.py under torchrun with the nccl backend prints one JSON line with per-rank timings."""
    if _ngpu() < 2:
        pytest.skip('needs 2 GPUs, this box has %d' % _ngpu())
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', str(port), os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--batch', '1',
                        '--height', '256', '--width', '256', '--no-cpu-baseline'], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    assert line['n_gpus'] == 2 and len(line['per_rank_ms_per_step']) == 2 and line['allreduce']['bytes'] == 4 * 7760484
