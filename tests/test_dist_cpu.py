"""CPU tests of the N>1 path: world_size-2 gloo process group -- bucketed gradient all-reduce, replica
broadcast, image sharding by global sample index (world-size invariant)."""
import os
import socket

import numpy as np
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


def test_gloo_world2_allreduce_broadcast_shard():
    world = 2
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
    expect = np.arange(n, dtype=np.float32) * 3.0       # (1 + 2) * arange
    shards = []
    for rank, ws, g, flat, loss, idx in res:
        assert ws == world
        assert np.array_equal(g, expect)
        assert np.all(flat == 5.0)                      # rank 0's replica everywhere
        assert loss == 0.5
        shards.append(idx)
    assert sorted(shards[0] + shards[1]) == list(range(11)) and not set(shards[0]) & set(shards[1])


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
