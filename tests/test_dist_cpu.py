"""CPU tests of the N>1 path: world_size-2 gloo process group -- bucketed gradient all-reduce, replica
broadcast, image sharding by global sample index (world-size invariant)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from eld_amd import dist as D
    w, r, _ = D.init(backend='gloo')
    assert (w, r) == (world, rank) and D.world_size() == world and D.rank() == rank
    n = 3 * 1024 + 17                                   # several buckets + ragged tail
    g = torch.arange(n, dtype=torch.float32) * (rank + 1)
    ws = D.allreduce_sum_(g, bucket=1024)
    flat = torch.full((1000,), float(rank + 5))
    D.broadcast_(flat, 0)
    loss = D.allreduce_mean_scalar(torch.tensor([float(rank)]))
    q.put((rank, ws, g.numpy().copy(), flat.numpy().copy(), float(loss), D.shard_indices(11)))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize('world', [2, 4, 8])
def test_gloo_allreduce_broadcast_shard(world):
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    n = 3 * 1024 + 17
    expect = np.arange(n, dtype=np.float32) * float(world * (world + 1) // 2)       # (1 + 2 + ... + world) * arange
    shards = []
    for rank, ws, g, flat, loss, idx in res:
        assert ws == world
        assert np.array_equal(g, expect)
        assert np.all(flat == 5.0)                      # rank 0's replica everywhere
        assert loss == (world - 1) / 2.0
        shards.append(idx)
    assert sorted(i for sh in shards for i in sh) == list(range(11))          # 11 images do not divide any of the world sizes: a partition all the same
    assert max(len(sh) for sh in shards) - min(len(sh) for sh in shards) <= 1


def test_single_process_defaults():
    from eld_amd import dist as D
    assert D.world_size() == 1 and D.rank() == 0
    g = torch.ones(10)
    assert D.allreduce_sum_(g) == 1 and torch.equal(g, torch.ones(10))
    assert D.shard_indices(5) == [0, 1, 2, 3, 4]
    assert D.shard_indices(7, rank_=1, world=3) == [1, 4]
    # the union of the rank shards is the same set of global sample ids for every world size
    for w in (1, 2, 4, 8):
        assert sorted(i for r in range(w) for i in D.shard_indices(64, r, w)) == list(range(64))


def test_bucket_ranges_of_the_unet_gradient_buffer():
    """The overlapped exchange reduces the flat gradient buffer top-down in 8 MiB buckets (the order in which the engine's backward finishes them):
    4 buckets for the U-Net's 7,760,484 parameters, the last one ragged; GradBuckets walks exactly this table (eld_amd/dist.py)."""
    from eld_amd import dist as D
    r = D.bucket_ranges(7760484)
    assert r == [(6291456, 7760484), (4194304, 6291456), (2097152, 4194304), (0, 2097152)]
    for numel, bucket in ((10, 3), (9, 3), (1, 8), (0, 4)):
        rr = D.bucket_ranges(numel, bucket)
        assert sorted(i for lo, hi in rr for i in range(lo, hi)) == list(range(numel))          # a partition ...
        assert all(a[0] > b[0] for a, b in zip(rr, rr[1:]))                                       # ... walked from the top


def _bucket_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch.distributed as dist
    from eld_amd import dist as D
    D.init(backend='gloo')
    n = 4 * 1000 + 123
    g = torch.arange(n, dtype=torch.float32) + 1000.0 * rank
    # what GradBuckets.allreduce_sum_ does, minus the CUDA events: views of ONE flat tensor, top bucket first, all in flight, then wait
    handles = [dist.all_reduce(g[lo:hi], op=dist.ReduceOp.SUM, async_op=True) for lo, hi in D.bucket_ranges(n, 1000)]
    for h in handles:
        h.wait()
    q.put((rank, g.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world4_bucketed_views_of_one_flat_buffer():
    world, port = 4, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_bucket_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    n = 4 * 1000 + 123
    expect = np.arange(n, dtype=np.float32) * world + 1000.0 * sum(range(world))
    for _, g in res:
        assert np.array_equal(g, expect)


def test_train_dataset_shards_pairs_over_ranks(monkeypatch):
    """Data parallel through the unmodified train_syn.py (train_syn.py:73-80 builds one ELDTrainDataset + DataLoader per process): every rank's dataset
    is ITS shard of the pairs -- a partition of range(total) with equal lengths (the tail wraps), for totals that do not divide the world size too."""
    from eld_amd import data as Dt

    for total, world in ((10, 1), (10, 2), (11, 4), (13, 8), (5, 8)):
        tgt = [np.full((4, 2, 2), i / 64.0, np.float32) for i in range(total)]          # (the host path clips the input to [0, 1])
        inp = [np.full((4, 2, 2), i / 128.0, np.float32) for i in range(total)]
        seen, lens = [], []
        for rank in range(world):
            monkeypatch.setenv('WORLD_SIZE', str(world)); monkeypatch.setenv('RANK', str(rank))
            ds = Dt.ELDTrainDataset(tgt, [inp], augment=False)
            assert (ds.world, ds.rank) == (world, rank)
            lens.append(len(ds))
            for i in range(len(ds)):
                d = ds[i]
                assert 2.0 * float(d['input'].ravel()[0]) == float(d['target'].ravel()[0])          # the pair stays a pair
                seen.append(int(round(64.0 * float(d['target'].ravel()[0]))))
        assert len(set(lens)) == 1 and lens[0] == -(-total // world)
        assert set(seen) == set(range(total)) and len(seen) == lens[0] * world                      # everything once; only the wrapped tail repeats
        assert len(seen) - total < world
    monkeypatch.delenv('WORLD_SIZE'); monkeypatch.delenv('RANK')


class _FakeModel:
    """Stands in for ELDModel on a CPU box (the real one needs a GPU): records what Engine asks of it."""
    saves = []

    def initialize(self, opt):
        from eld_amd import dist as D
        self.opt, self.epoch, self.iterations, self.rank = opt, 0, 0, D.rank()
        self.optimizers = []
        self.evaluated = []

    def set_input(self, data, mode='train'):
        self.cur = data

    def optimize_parameters(self, **kw):
        pass

    def get_current_errors(self):
        return {'Pixel': float(self.cur['v'])}

    def update_learning_rate(self):
        pass

    def save(self, label=None):                  # ELDModel.save: rank 0 only (eld_amd/model.py)
        if self.rank == 0:
            _FakeModel.saves.append(label)

    def eval(self, data, savedir=None, **kw):
        self.evaluated.append(int(data['id']))
        return {'PSNR': float(data['id']), 'SSIM': 1.0}


def _engine_worker(rank, world, port, tmp, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), ELD_AMD_PREFETCH='0')
    import contextlib
    import io
    import types
    from eld_amd import dist as D
    from eld_amd import engine as E
    D.init(backend='gloo')
    E.models = types.SimpleNamespace(eld_model=_FakeModel)
    opt = types.SimpleNamespace(checkpoints_dir=tmp, name='dp', model='eld_model', no_log=False, save_epoch_freq=1)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        eng = E.Engine(opt)
        eng.train([{'v': 1.0}, {'v': 3.0}])
        meters = eng.eval([{'id': i} for i in range(7)], 'set', loss_key='PSNR')       # 7 images do not divide the world
    q.put((rank, buf.getvalue(), list(_FakeModel.saves), eng.model.evaluated, meters['PSNR'], meters.n['PSNR'], eng.best_val_loss))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize('world', [2, 4])
def test_rank0_logs_and_saves_and_eval_shards_images(world, tmp_path):
    """Engine under data parallelism (engine.py:31-99 run by every rank): rank 0 alone prints and writes checkpoints; the evaluation list is
    sharded by image ("replicas only") and every rank ends with the whole job's mean."""
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_engine_worker, args=(r, world, port, str(tmp_path), q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    all_ids = []
    for rank, out, saves, ids, psnr, n, best in res:
        if rank == 0:
            assert 'Epoch: 0' in out and 'Time Taken' in out
            assert saves[:2] == [None, 'latest'] and 'best_PSNR_set' in saves       # epoch checkpoint, latest, best-of-eval: rank 0 writes them
        else:
            assert out == '' and saves == []
        assert ids == list(range(rank, 7, world))
        all_ids += ids
        assert n == 7 and abs(psnr - 3.0) < 1e-12 and abs(best - 3.0) < 1e-12     # mean of 0..6 on EVERY rank
    assert sorted(all_ids) == list(range(7))
