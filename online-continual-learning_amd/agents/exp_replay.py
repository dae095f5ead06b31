"""agents/exp_replay.py:10-104 — Experience Replay inner loop (also hosts MIR / ASER through the plugins)."""
import os

import numpy as np
import torch

from .. import debug
from ..buffer import Buffer
from ..data import DeviceLoader
from ..utils import maybe_cuda, AverageMeter
from .base import ContinualLearner


class ExperienceReplay(ContinualLearner):
    def __init__(self, model, opt, params):
        super(ExperienceReplay, self).__init__(model, opt, params)
        self.buffer = Buffer(model, params)
        self.mem_size = params.mem_size
        self.eps_mem_batch = params.eps_mem_batch
        self.mem_iters = params.mem_iters

    def train_learner(self, x_train, y_train):
        self.before_train(x_train, y_train)
        # set up loader: device-resident task, same sampler / RNG draws as the reference's DataLoader
        train_loader = DeviceLoader(x_train, y_train, self.batch, shuffle=True, drop_last=True)
        # set up model
        self.model = self.model.train()

        # setup tracker
        losses_batch = AverageMeter()
        losses_mem = AverageMeter()
        acc_batch = AverageMeter()
        acc_mem = AverageMeter()
        aser = self.params.update == 'ASER' or self.params.retrieve == 'ASER'
        # one two-group pass instead of the batch pass + memory pass (see below)
        merge = (self.params.retrieve == 'random' and not aser and not self.params.trick['kd_trick'] and not self.params.trick['kd_trick_star']
                 and os.environ.get("OCL_ER_MERGE", "1") != "0")

        for ep in range(self.epoch):
            for i, batch_data in enumerate(train_loader):
                # batch update
                batch_x, batch_y = batch_data
                batch_y_host = train_loader.last_y_host
                for j in range(self.mem_iters):
                    if merge:
                        # Random retrieval reads neither the model nor its gradients, and nothing between the batch forward and
                        # the retrieval draws from an RNG: retrieving first leaves every RNG stream as the reference's order does.
                        # The batch pass and the memory pass (same weights, gradients summed by the two backward() calls) then
                        # run as ONE two-group pass with per-group BatchNorm statistics, like SCR's two views: half the launches
                        # of a step whose kernels sit at their latency floor.
                        mem_x, mem_y = self.buffer.retrieve(x=batch_x, y=batch_y)
                        if mem_x.size(0) == batch_x.size(0):
                            mem_x = maybe_cuda(mem_x, self.cuda)
                            mem_y = maybe_cuda(mem_y, self.cuda)
                            both = self.model.forward_views([batch_x, mem_x])
                            logits, mem_logits = both[:batch_x.size(0)], both[batch_x.size(0):]
                            loss = self.criterion(logits, batch_y)
                            loss_mem = self.criterion(mem_logits, mem_y)
                            if self.verbose:
                                _, pred_label = torch.max(logits, 1)
                                acc_batch.update((pred_label == batch_y).sum() / batch_y.size(0), batch_y.size(0))
                                losses_batch.update(loss, batch_y.size(0))
                                losses_mem.update(loss_mem, mem_y.size(0))
                                _, pred_label = torch.max(mem_logits, 1)
                                acc_mem.update((pred_label == mem_y).sum() / mem_y.size(0), mem_y.size(0))
                            if debug.on():
                                debug.emit("er_loss", loss=float(loss.detach()))
                                debug.emit("er_loss_mem", loss=float(loss_mem.detach()))
                            self.opt.zero_grad()
                            (loss + loss_mem).backward()
                            self.opt.step()
                            continue
                        pre_retrieved = (mem_x, mem_y)     # empty or short memory batch: the reference's two passes
                    else:
                        pre_retrieved = None
                    logits = self.model.forward(batch_x)
                    loss = self.criterion(logits, batch_y)
                    if self.params.trick['kd_trick']:
                        loss = 1 / (self.task_seen + 1) * loss + (1 - 1 / (self.task_seen + 1)) * \
                                   self.kd_manager.get_kd_loss(logits, batch_x)
                    if self.params.trick['kd_trick_star']:
                        loss = 1/((self.task_seen + 1) ** 0.5) * loss + \
                               (1 - 1/((self.task_seen + 1) ** 0.5)) * self.kd_manager.get_kd_loss(logits, batch_x)
                    if self.verbose:
                        # trackers only (the reference syncs with .item() every iteration; here only when printing)
                        _, pred_label = torch.max(logits, 1)
                        acc_batch.update((pred_label == batch_y).sum() / batch_y.size(0), batch_y.size(0))
                        losses_batch.update(loss, batch_y.size(0))
                    if debug.on():
                        debug.emit("er_loss", loss=float(loss.detach()))
                    # backward (in ASER mode the gradients of this pass and of the memory pass are discarded by the zero_grad()
                    # in front of the combined pass below -- only their forward's BatchNorm running-stat updates survive -- so
                    # the backward is skipped unless MIR retrieval reads this pass's gradient vector)
                    self.opt.zero_grad()
                    if not aser or self.params.retrieve == 'MIR':
                        loss.backward()

                    # mem update
                    mem_x, mem_y = pre_retrieved if pre_retrieved is not None else self.buffer.retrieve(x=batch_x, y=batch_y)
                    if mem_x.size(0) > 0:
                        mem_x = maybe_cuda(mem_x, self.cuda)
                        mem_y = maybe_cuda(mem_y, self.cuda)
                        mem_logits = self.model.forward(mem_x)
                        loss_mem = self.criterion(mem_logits, mem_y)
                        if self.params.trick['kd_trick']:
                            loss_mem = 1 / (self.task_seen + 1) * loss_mem + (1 - 1 / (self.task_seen + 1)) * \
                                       self.kd_manager.get_kd_loss(mem_logits, mem_x)
                        if self.params.trick['kd_trick_star']:
                            loss_mem = 1 / ((self.task_seen + 1) ** 0.5) * loss_mem + \
                                   (1 - 1 / ((self.task_seen + 1) ** 0.5)) * self.kd_manager.get_kd_loss(mem_logits,
                                                                                                         mem_x)
                        if self.verbose:
                            losses_mem.update(loss_mem, mem_y.size(0))
                            _, pred_label = torch.max(mem_logits, 1)
                            acc_mem.update((pred_label == mem_y).sum() / mem_y.size(0), mem_y.size(0))

                        if debug.on():
                            debug.emit("er_loss_mem", loss=float(loss_mem.detach()))
                        if not aser:
                            loss_mem.backward()

                    if aser:
                        # opt update: passes #1/#2 only leave their BatchNorm running-stat updates behind
                        self.opt.zero_grad()
                        combined_batch = torch.cat((mem_x, batch_x))
                        combined_labels = torch.cat((mem_y, batch_y))
                        if getattr(mem_y, 'host', None) is not None:
                            combined_labels.host = np.concatenate((np.asarray(mem_y.host), np.asarray(batch_y_host)))
                        combined_logits = self.model.forward(combined_batch)
                        loss_combined = self.criterion(combined_logits, combined_labels)
                        if debug.on():
                            debug.emit("er_loss_combined", loss=float(loss_combined.detach()))
                        loss_combined.backward()
                        self.opt.step()
                    else:
                        self.opt.step()

                # update mem
                self.buffer.update(batch_x, batch_y, y_host=batch_y_host)

                if i % 100 == 1 and self.verbose:
                    print(
                        '==>>> it: {}, avg. loss: {:.6f}, '
                        'running train acc: {:.3f}'
                            .format(i, losses_batch.avg(), acc_batch.avg())
                    )
                    print(
                        '==>>> it: {}, mem avg. loss: {:.6f}, '
                        'running mem acc: {:.3f}'
                            .format(i, losses_mem.avg(), acc_mem.avg())
                    )
        self.after_train()
