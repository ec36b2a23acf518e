"""Data-parallel plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on
ROCm; "gloo" for the CPU tests).  The reference is single-device (models/ELD_model.py:187-190); sharding
by image over ranks is new functionality.

The only exchange step of the path is the gradient all-reduce: all 7,760,484 fp32 gradients live in ONE
flat buffer (eld_amd.unet), so the reduction is a handful of large contiguous RCCL calls -- xGMI is
point-to-point (7 links x ~153 GB/s per GPU) and a ring all-reduce is per-link bound, so few large
buckets beat per-tensor calls.  The engine's backward is one stream-ordered call that produces the flat
gradient buffer from its END towards its start and records one event per bucket as soon as the bucket is
final (include/eld_amd.h eld_unet_backward_buckets); GradBuckets all-reduces each bucket on a communication
stream behind its event, so the 31 MB exchange runs under the rest of the backward.  Gradient averaging
(sum / world) is folded into the fused Adam's grad_scale, so no extra pass over the gradients is made.
"""
import ctypes
import os

import torch
import torch.distributed as dist

BUCKET_FLOATS = 2 * 1024 * 1024      # 8 MiB buckets: 4 of them for the 7.76 M-parameter U-Net


def env_world():
    return int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0'))


def init(backend=None):
    """Initialise the default process group from the torchrun environment (no-op for world size 1)."""
    world, rank, local = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            backend = os.environ.get('ELD_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        if torch.cuda.is_available():
            torch.cuda.set_device(local % torch.cuda.device_count())
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return world, rank, local


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def ranks_share_device():
    """True when the job has more ranks than this node has GPUs (the one-GPU gloo smoke runs of the N > 1 path): two processes then time-slice ONE
    device, and a process that keeps a second stream busy beside its compute stream (ELDModel.prefetch_input) was measured 24x slower in that setting
    (424 against 17.7 ms per 512 x 512 step, profiles/r06_ab_notes.md) -- callers switch the lookahead off."""
    return world_size() > 1 and torch.cuda.is_available() and world_size() > torch.cuda.device_count()


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def allreduce_sum_(flat, bucket=BUCKET_FLOATS):
    """In-place SUM all-reduce of a flat gradient buffer in large contiguous buckets.  Returns the
    world size (the caller divides -- the fused Adam takes it as grad_scale = 1/world)."""
    w = world_size()
    if w == 1:
        return 1
    handles = []
    for o in range(0, flat.numel(), bucket):
        handles.append(dist.all_reduce(flat[o:o + bucket], op=dist.ReduceOp.SUM, async_op=True))
    for h in handles:
        h.wait()
    return w


def bucket_ranges(numel, bucket=BUCKET_FLOATS):
    """[(lo, hi)] of the buckets of a flat gradient buffer in REDUCTION order: top-down, the order in which the engine's backward makes them
    final (it produces the flat buffer from its end towards its start: head, decoder, encoder -- the reverse of named_parameters())."""
    numel, bucket = int(numel), int(bucket)
    starts = list(range(0, numel, bucket))
    return [(lo, min(lo + bucket, numel)) for lo in reversed(starts)]


class GradBuckets:
    """Bucket table of a flat CUDA gradient buffer for the overlapped exchange: ascending start offsets, one CUDA event per
    bucket (recorded by the engine's backward on the compute stream) and a communication stream."""

    def __init__(self, numel, device, bucket=BUCKET_FLOATS):
        self.numel = int(numel)
        self.bucket = int(bucket)
        self.starts = list(range(0, self.numel, int(bucket)))
        self.events = [torch.cuda.Event() for _ in self.starts]
        with torch.cuda.device(device):
            for e in self.events:
                e.record()                       # materialises the hipEvent_t behind the torch object
            self.stream = torch.cuda.Stream(device)
        n = len(self.starts)
        self.starts_c = (ctypes.c_int64 * n)(*self.starts)
        self.events_c = (ctypes.c_void_p * n)(*[e.cuda_event for e in self.events])
        self.n = n

    def allreduce_sum_(self, flat):
        """Call after the bucketed backward has been enqueued on the current stream.  Buckets are reduced top-down (the order
        in which they become final); the current stream then waits for all of them.  Returns the world size."""
        w = world_size()
        if w == 1:
            return 1
        handles = []
        for k, (lo, hi) in zip(range(self.n - 1, -1, -1), bucket_ranges(self.numel, self.bucket)):
            assert lo == self.starts[k]
            self.stream.wait_event(self.events[k])
            with torch.cuda.stream(self.stream):
                handles.append(dist.all_reduce(flat[lo:hi], op=dist.ReduceOp.SUM, async_op=True))
        for h in handles:
            h.wait()
        torch.cuda.current_stream().wait_stream(self.stream)
        return w


def broadcast_(flat, src=0):
    if world_size() > 1:
        dist.broadcast(flat, src=src)


def shard_indices(n, rank_=None, world=None):
    """Image shard of this rank: indices rank, rank+world, ... (global sample index = Philox sample id,
    so the union over ranks is identical for every world size)."""
    r = rank() if rank_ is None else rank_
    w = world_size() if world is None else world
    return list(range(r, n, w))


def shard_len(n, world=None):
    """Length of every rank's shard of n items: ceil(n / world) -- equal on all ranks (the tail wraps around to the first items, as
    torch.utils.data.DistributedSampler pads), because the gradient exchange averages per-rank MEANS (equal local batch sizes)."""
    w = world_size() if world is None else world
    return (int(n) + w - 1) // w


def shard_item(i, n, rank_=None, world=None):
    """Global index of item i of this rank's shard of n items (rank-strided; indices past n wrap to the start)."""
    r = rank() if rank_ is None else rank_
    w = world_size() if world is None else world
    g = int(i) * w + r
    return g % int(n) if n else g


def allreduce_meters(sums, counts):
    """Sum the (sum, count) pairs of running means over all ranks (evaluation sharded by image: SURVEY.md 8(e) "replicas only"); dicts of floats /
    ints with the same keys on every rank.  Returns (sums, counts) of the whole job; identity for world size 1."""
    if world_size() == 1:
        return dict(sums), dict(counts)
    keys = sorted(sums)
    t = torch.tensor([float(sums[k]) for k in keys] + [float(counts[k]) for k in keys], dtype=torch.float64)
    if dist.get_backend() == 'nccl':
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    v = t.cpu().tolist()
    n = len(keys)
    return {k: v[i] for i, k in enumerate(keys)}, {k: int(round(v[n + i])) for i, k in enumerate(keys)}


def allreduce_mean_scalar(x):
    if world_size() == 1:
        return x
    t = x.detach().clone()
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t / world_size()
