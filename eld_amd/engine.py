"""Epoch driver: mirror of the reference's Engine (engine.py:10-128) over eld_amd.model.ELDModel.

Same surface -- Engine(opt), train(loader), eval(loader, dataset_name, ...), set_learning_rate(lr),
epoch / iterations properties, checkpoint cadence -- minus the tensorboard / progress-bar plumbing of
util/util.py, which is outside the hot path.  In data-parallel runs every rank executes the same loop on
its shard of each batch; rank 0 logs and saves.
"""
import os
import time

import torch

from . import dist as D
from . import model as models


class AverageMeters:                     # util/util.py:146-173 (running means), minimal
    def __init__(self):
        self.sum, self.n = {}, {}

    def update(self, dic):
        for k, v in dic.items():
            self.sum[k] = self.sum.get(k, 0.0) + float(v)
            self.n[k] = self.n.get(k, 0) + 1

    def __getitem__(self, k):
        return self.sum[k] / max(self.n[k], 1)

    def __str__(self):
        return ' | '.join('%s: %.4f' % (k, self[k]) for k in self.sum)


class Engine(object):
    def __init__(self, opt):
        self.opt = opt
        self.best_val_loss = 1e6
        self.basedir = os.path.join(getattr(opt, 'checkpoints_dir', './checkpoints'), opt.name)      # engine.py:19-21
        if D.rank() == 0:
            os.makedirs(self.basedir, exist_ok=True)
        self.model = getattr(models, getattr(opt, 'model', 'eld_model'))()                          # engine.py:26
        self.model.initialize(opt)

    def train(self, train_loader, **kwargs):         # engine.py:31-72
        if D.rank() == 0:
            print('\nEpoch: %d' % self.epoch)
        avg_meters = AverageMeters()
        model = self.model
        t0 = time.time()
        # ELD_AMD_LOG_EVERY=k: read the loss back (one device sync) every k-th iteration only.  Default 1 = the reference's behaviour: it reads
        # loss.item() every iteration (engine.py:48-53, ELD_model.py:480).  An environment switch, not an option: the reference's CLI has no such flag.
        log_every = max(1, int(os.environ.get('ELD_AMD_LOG_EVERY', '1') or 1))
        # One batch of lookahead (ELD_AMD_PREFETCH=0 switches it off): batch i+1 is fetched from the loader and its on-device synthesis started on the
        # model's synthesis stream before iteration i's U-Net kernels are enqueued -- the reference overlaps synthesis with training the same way,
        # through its DataLoader workers (train_syn.py:78-80).  Batches, host draws and results are those of the plain loop, in the same order.
        prefetch = os.environ.get('ELD_AMD_PREFETCH', '1') != '0' and hasattr(model, 'prefetch_input')
        it = iter(train_loader)
        data = next(it, None)
        i = 0
        while data is not None:
            model.set_input(data, mode='train')
            nxt = next(it, None)
            if prefetch and nxt is not None:
                model.prefetch_input(nxt, mode='train')
            model.optimize_parameters(**kwargs)
            if i % log_every == 0:                   # the reference reads loss.item() every iteration (ELD_model.py:480)
                avg_meters.update(model.get_current_errors())
            self.iterations += 1
            data = nxt
            i += 1
        self.epoch += 1
        if not getattr(self.opt, 'no_log', False):
            if self.epoch % getattr(self.opt, 'save_epoch_freq', 100) == 0:
                model.save()
            model.save(label='latest')
            if D.rank() == 0:
                print('Time Taken: %d sec  [%s]' % (time.time() - t0, avg_meters))
        model.update_learning_rate()
        return avg_meters

    def eval(self, val_loader, dataset_name, savedir=None, loss_key=None, **kwargs):       # engine.py:75-99
        avg_meters = AverageMeters()
        world, rank = D.world_size(), D.rank()
        with torch.no_grad():
            for i, data in enumerate(val_loader):
                if world > 1 and i % world != rank:      # evaluation shards by image over the ranks ("replicas only": no data-path collective)
                    continue
                avg_meters.update(self.model.eval(data, savedir=savedir, **kwargs))
        if world > 1:                                    # ... and only the scalar sums travel: every rank ends with the whole job's means
            avg_meters.sum, avg_meters.n = D.allreduce_meters(avg_meters.sum, avg_meters.n)
        if loss_key is not None and avg_meters[loss_key] < self.best_val_loss:
            self.best_val_loss = avg_meters[loss_key]
            self.model.save(label='best_{}_{}'.format(loss_key, dataset_name))
        return avg_meters

    def test(self, test_loader, savedir=None, **kwargs):                                   # engine.py:101-107
        outs = []
        with torch.no_grad():
            for data in test_loader:
                outs.append(self.model.test(data, savedir=savedir, **kwargs))
        return outs

    def set_learning_rate(self, lr):                 # engine.py:109-112
        for optimizer in self.model.optimizers:
            if D.rank() == 0:
                print('[i] set learning rate to {}'.format(lr))
            for g in optimizer.param_groups:
                g['lr'] = lr

    @property
    def iterations(self):
        return self.model.iterations

    @iterations.setter
    def iterations(self, i):
        self.model.iterations = i

    @property
    def epoch(self):
        return self.model.epoch

    @epoch.setter
    def epoch(self, e):
        self.model.epoch = e
