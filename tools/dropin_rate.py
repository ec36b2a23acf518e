#!/usr/bin/env python3
"""End-to-end iteration rate of the drop-in training path at the reference's own shape (train_syn.py defaults: one 4x512x512 patch per
step, 8 DataLoader workers): NoiseModel -> LMDBDataset -> SynDataset -> ELDTrainDataset -> DataLoader -> Engine.train over the eld_amd plugins,
LMDB records from the in-memory stand-in of eld_amd.launch (no /root/reference needed).  Prints iterations/s and raw MPix/s next to the
device-only step time of bench.py at the same shape.  Usage: dropin_rate.py [batch=1] [workers=8] [patches=256]"""
import contextlib
import io
import os
import sys
import time
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np          # noqa: E402
import torch                # noqa: E402


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    workers = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    patches = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    import tempfile
    tmp = tempfile.mkdtemp()
    os.chdir(tmp)
    import eld_amd.launch as Lm
    Lm.install_shims(synthetic_lmdb=True, patches=patches, patch_hw=(512, 512))
    import eld_amd.noise as noise
    from eld_amd import data as datasets
    from eld_amd.engine import Engine
    np.random.seed(2018)
    torch.manual_seed(2018)
    with contextlib.redirect_stdout(io.StringIO()):
        nm = noise.NoiseModel(model='PGRU', include=4)
    target = datasets.LMDBDataset('data/Train/SID_Sony_Raw.db')
    inp = datasets.SynDataset(datasets.LMDBDataset('data/Train/SID_Sony_Raw.db'), noise_maker=nm, num_burst=1)
    ds = datasets.ELDTrainDataset(target_dataset=target, input_datasets=[inp])
    loader = torch.utils.data.DataLoader(ds, batch_size=batch, shuffle=True, num_workers=workers, pin_memory=True,
                                         worker_init_fn=datasets.worker_init_fn if workers else None, persistent_workers=workers > 0)
    opt = types.SimpleNamespace(gpu_ids=[0], isTrain=True, checkpoints_dir=tmp, name='t', netG='unet', channels=4, stage_in='raw', stage_out='raw',
                                lr=1e-4, beta1=0.9, wd=0.0, loss='l1', resume=False, chop=False, no_log=True, save_epoch_freq=10 ** 6, model='eld_model', seed=2018)
    with contextlib.redirect_stdout(io.StringIO()):
        eng = Engine(opt)
        eng.train(loader)                                   # warm-up epoch (workers start, caches fill)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.train(loader)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    n = len(loader)
    print('drop-in path, batch %d x 4x512x512, %d workers: %d iterations in %.2f s = %.1f it/s = %.2f ms per iteration = %.1f raw MPix/s' % (
        batch, workers, n, dt, n / dt, dt / n * 1e3, n * batch * 4 * 512 * 512 / dt / 1e6))


if __name__ == '__main__':
    main()
