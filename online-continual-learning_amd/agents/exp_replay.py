"""Experience Replay inner loop on the engine (reference: agents/exp_replay.py:10-104; also the host of the MIR and ASER plugins).

Per stream batch the reference does: batch pass (forward, CE, backward) -> buffer.retrieve -> memory pass (forward, CE, backward
into the same gradients) -> [ASER mode: zero_grad, one pass over memory + batch] -> opt.step -> buffer.update.  This file keeps that
sequence of model / RNG / buffer effects and chooses a cheaper schedule where the result is provably the same:

* random retrieval (`_merged_step`): the memory batch is drawn first and batch + memory run as ONE two-group pass;
* ASER mode (`_two_pass_step`): the gradients of the first two passes are thrown away by the reference's zero_grad(), so their
  backward is not run (MIR retrieval still gets the batch pass's gradient vector).
"""
import contextlib
import os

import numpy as np
import torch

from .. import debug
from .. import ops
from ..buffer import Buffer
from ..data import DeviceLoader
from ..utils import maybe_cuda, AverageMeter
from ..loss import unit_gradient
from .base import ContinualLearner


class ExperienceReplay(ContinualLearner):
    def __init__(self, model, opt, params):
        super(ExperienceReplay, self).__init__(model, opt, params)
        self.buffer = Buffer(model, params)
        self.mem_size = params.mem_size
        self.eps_mem_batch = params.eps_mem_batch
        self.mem_iters = params.mem_iters

    # ---- pieces of a step ----------------------------------------------------------------------------------------------
    def _kd_mix(self, loss, logits, x):
        """CE blended with the distillation loss against last task's model (exp_replay.py:42-47 / :64-69)."""
        trick = self.params.trick
        if trick['kd_trick']:
            w = 1 / (self.task_seen + 1)
            loss = w * loss + (1 - w) * self.kd_manager.get_kd_loss(logits, x)
        if trick['kd_trick_star']:
            w = 1 / ((self.task_seen + 1) ** 0.5)
            loss = w * loss + (1 - w) * self.kd_manager.get_kd_loss(logits, x)
        return loss

    def _track(self, meters, logits, labels, loss):
        """Running loss / accuracy (only when printing: the reference's per-iteration .item() would stall the stream)."""
        if not self.verbose:
            return
        loss_meter, acc_meter = meters
        hits = (torch.max(logits, 1)[1] == labels).sum()
        acc_meter.update(hits / labels.size(0), labels.size(0))
        loss_meter.update(loss, labels.size(0))

    @staticmethod
    def _emit(tag, loss):
        if debug.on():
            debug.emit(tag, loss=float(loss.detach()))

    def _merged_step(self, batch_x, batch_y, mem_x, mem_y, meters):
        """Batch pass + memory pass as one pass with two BatchNorm groups (statistics and running-stat updates per group, in
        order); the two CE losses are back-propagated together, which is what two backward() calls into the same gradients sum to."""
        n = batch_x.size(0)
        trick = self.params.trick
        if (hasattr(self.model, "forward_views_taped") and not trick['labels_trick'] and not trick['separated_softmax']
                and self.params.agent not in ('SCR', 'SCP') and not getattr(self, "_force_autograd", False)):   # (_force_autograd: the A/B test)
            # plain cross-entropy (agents/base.py:113): the two losses write their dL/dlogits into the two row blocks of ONE buffer and
            # the engine's backward takes it as it is -- the same numbers autograd would assemble with two slice-backward fills, two
            # copies and an add (six tiny launches on the dependent chain of a 0.9 ms step)
            both, tape = self.model.forward_views_taped([batch_x, mem_x])
            dl = torch.empty_like(both)
            loss, _ = ops.cross_entropy(both[:n], batch_y, "mean", dl_out=dl[:n])
            loss_mem, _ = ops.cross_entropy(both[n:], mem_y, "mean", dl_out=dl[n:])
            self._track(meters[0], both[:n], batch_y, loss)
            self._track(meters[1], both[n:], mem_y, loss_mem)
            self._emit("er_loss", loss)
            self._emit("er_loss_mem", loss_mem)
            self.opt.zero_grad()
            self.model.backward_taped(tape, dl)
            self.opt.step()
            return
        both = self.model.forward_views([batch_x, mem_x])
        logits, mem_logits = both[:n], both[n:]
        loss, loss_mem = self.criterion(logits, batch_y), self.criterion(mem_logits, mem_y)
        self._track(meters[0], logits, batch_y, loss)
        self._track(meters[1], mem_logits, mem_y, loss_mem)
        self._emit("er_loss", loss)
        self._emit("er_loss_mem", loss_mem)
        self.opt.zero_grad()
        total = loss + loss_mem
        total.backward(unit_gradient(total))
        self.opt.step()

    def _passes_12_have_readers(self, aser):
        """ASER mode throws the gradients of the batch pass and the memory pass away (exp_replay.py:76 zero_grad): their logits and
        losses are only ever printed / traced / mixed with a distillation term.  Without such a reader (no verbose meter, no trace, no
        KD trick, no MIR retrieval reading the batch pass's gradient) the two passes are run for what they DO leave behind -- their
        BatchNorm running-statistic updates -- and nothing else: no head, no loss kernels, no tape (model.forward_stats_only).
        `_force_losses` (tests) makes the readers' path run regardless: tests/test_gpu_steps.py proves the two paths end in the same
        weights, statistics and memory bit for bit."""
        trick = self.params.trick
        return (not aser or self.params.retrieve == 'MIR' or self.verbose or debug.on() or trick['kd_trick'] or trick['kd_trick_star']
                or getattr(self, "_force_losses", False) or not hasattr(self.model, "forward_stats_only"))

    def _two_pass_step(self, batch_x, batch_y, batch_y_host, meters, aser, retrieved=None, logits=None):
        back12 = not aser or self.params.retrieve == 'MIR'   # MIR reads the batch pass's gradient
        need_loss = self._passes_12_have_readers(aser)
        if logits is None:      # (else: the pipelined ASER loop has issued this pass already)
            if need_loss:
                logits = self.model.forward(batch_x)
            else:
                self.model.forward_stats_only(batch_x)
        if need_loss:
            loss = self._kd_mix(self.criterion(logits, batch_y), logits, batch_x)
            self._track(meters[0], logits, batch_y, loss)
            self._emit("er_loss", loss)
        self.opt.zero_grad()
        if back12:
            loss.backward(unit_gradient(loss))

        mem_x, mem_y = retrieved if retrieved is not None else self.buffer.retrieve(x=batch_x, y=batch_y)
        if mem_x.size(0) > 0:
            mem_x, mem_y = maybe_cuda(mem_x, self.cuda), maybe_cuda(mem_y, self.cuda)
            if not need_loss:
                self.model.forward_stats_only(mem_x)
            else:
                mem_logits = self.model.forward(mem_x)
                loss_mem = self._kd_mix(self.criterion(mem_logits, mem_y), mem_logits, mem_x)
                self._track(meters[1], mem_logits, mem_y, loss_mem)
                self._emit("er_loss_mem", loss_mem)
                if not aser:
                    loss_mem.backward(unit_gradient(loss_mem))

        if aser:
            # exp_replay.py:76-84: the update comes from one more pass over memory + batch; passes 1 and 2 only leave their
            # BatchNorm running-statistic updates behind
            self.opt.zero_grad()
            combined_labels = torch.cat((mem_y, batch_y))
            if getattr(mem_y, 'host', None) is not None:
                combined_labels.host = np.concatenate((np.asarray(mem_y.host), np.asarray(batch_y_host)))
            # (torch.cat((mem_x, batch_x)) is not materialised: the engine reads the two pieces where they are)
            trick = self.params.trick
            if (hasattr(self.model, "forward_views_taped") and not trick['labels_trick'] and not trick['separated_softmax']
                    and not getattr(self, "_force_autograd", False) and os.environ.get("OCL_ASER_AUTOGRAD", "0") != "1"):   # (both: the A/B's autograd side)
                # plain cross-entropy (agents/base.py:113), as in _merged_step: the loss kernel's dL/dlogits goes straight into the engine's
                # backward -- the same numbers, without autograd's engine between the loss and the backward (a hand-over to its worker
                # thread with the GPU's queue already empty: ~0.1 ms of an ASER step, profiles/r6_aser_step_timeline.txt)
                out, tape = self.model.forward_views_taped([(mem_x, batch_x)])
                loss_combined, dl = ops.cross_entropy(out, combined_labels, "mean")
                self._emit("er_loss_combined", loss_combined)
                self.model.backward_taped(tape, dl)
            else:
                loss_combined = self.criterion(self.model.forward((mem_x, batch_x)), combined_labels)
                self._emit("er_loss_combined", loss_combined)
                loss_combined.backward(unit_gradient(loss_combined))
        self.opt.step()

    # ---- the loop ------------------------------------------------------------------------------------------------------
    def train_learner(self, x_train, y_train):
        # Between two optimiser steps every forward of the loop reads the same weights (ASER mode: five of them per iteration): inside
        # model.same_weights() the engine packs them once per step (resnet.py; writes by anybody else are detected, not assumed away)
        same = self.model.same_weights() if hasattr(self.model, "same_weights") else contextlib.nullcontext()
        with self.launch_stream(), same:
            self._train_learner(x_train, y_train)

    def _train_learner(self, x_train, y_train):
        self.before_train(x_train, y_train)
        # device-resident task behind the reference's DataLoader (same sampler, same RNG draws)
        train_loader = DeviceLoader(x_train, y_train, self.batch, shuffle=True, drop_last=True)
        self.model = self.model.train()
        meters = ((AverageMeter(), AverageMeter()), (AverageMeter(), AverageMeter()))   # (loss, acc) of batch / memory passes

        trick = self.params.trick
        aser = self.params.update == 'ASER' or self.params.retrieve == 'ASER'
        # Random retrieval reads neither the model nor its gradients and nothing between the batch forward and the retrieval draws
        # from an RNG, so retrieving first leaves every RNG stream as the reference's order does.
        merge = self.params.retrieve == 'random' and not aser and not trick['kd_trick'] and not trick['kd_trick_star']

        # ASER update: its host half (wait for the ranking, class-table bookkeeping, row moves) leaves the GPU idle, and the first
        # forward of the NEXT iteration (batch pass: reads the weights, which are final after opt.step, and updates the running
        # statistics, which the update's scoring kernels - already issued - read before it) does not depend on it.  That forward is
        # issued first, the update is finished behind it: same kernels, same data, same RNG draws, the host half hidden.
        upd = self.buffer.update_method
        pipeline = (aser and self.params.update == 'ASER' and self.params.retrieve != 'MIR' and self.cuda and not debug.on()
                    and not trick['kd_trick'] and not trick['kd_trick_star'] and hasattr(upd, "update_begin")
                    and os.environ.get("OCL_ASER_PIPELINE", "1") != "0")

        # Random retrieval + reservoir update never touch the model: like agents/scr.py, the data path (loader gather, retrieve gather,
        # reservoir scatter: ~20 tiny launches, 5 % of a 1 ms step) goes to its own stream, so that step i+1's data work runs next to
        # step i's backward instead of in front of step i+1's forward.  Same statements, same order, same RNG draws (OCL_DATA_STREAM=0:
        # everything on the main stream; tests/test_gpu_steps.py::test_er_data_stream_overlap_is_schedule_only: bit-identical end state).
        overlap = (merge and self.cuda and self.params.update == 'random' and not debug.on()
                   and os.environ.get("OCL_DATA_STREAM", "1") != "0")
        main = torch.cuda.current_stream() if self.cuda else None
        ds = ops.data_stream(x_train.device if torch.is_tensor(x_train) and x_train.is_cuda else torch.cuda.current_device()) if overlap else None
        if overlap:
            ds.wait_stream(main)

        def on_data():
            return torch.cuda.stream(ds) if overlap else contextlib.nullcontext()

        for ep in range(self.epoch):
            pending = None
            loader_it = iter(train_loader)
            i = -1
            while True:
                with on_data():
                    batch_data = next(loader_it, None)
                if batch_data is None:
                    break
                i += 1
                batch_x, batch_y = batch_data
                batch_y_host = train_loader.last_y_host
                pre = None
                if pipeline:
                    # (same weights as update_begin's feature pass, which read the stepped weights first)
                    if self._passes_12_have_readers(aser):
                        pre = self.model.forward(batch_x)
                    else:
                        self.model.forward_stats_only(batch_x)
                        pre = True          # "issued": nobody reads its logits
                    upd.update_finish(self.buffer, pending)
                    pending = None
                for j in range(self.mem_iters):
                    retrieved = None
                    if merge:
                        with on_data():
                            retrieved = self.buffer.retrieve(x=batch_x, y=batch_y)
                            retrieved = (maybe_cuda(retrieved[0], self.cuda), maybe_cuda(retrieved[1], self.cuda))
                        if overlap:
                            main.wait_stream(ds)
                            for t in (batch_x, batch_y) + retrieved:
                                t.record_stream(main)   # allocated on the data stream, consumed on the main one
                        if retrieved[0].size(0) == batch_x.size(0):
                            self._merged_step(batch_x, batch_y, retrieved[0], retrieved[1], meters)
                            continue
                    self._two_pass_step(batch_x, batch_y, batch_y_host, meters, aser, retrieved, logits=pre if j == 0 else None)

                if pipeline:
                    pending = upd.update_begin(self.buffer, batch_x, batch_y, y_host=batch_y_host)
                else:
                    with on_data():
                        self.buffer.update(batch_x, batch_y, y_host=batch_y_host)

                if i % 100 == 1 and self.verbose:
                    for tag, (loss_meter, acc_meter) in zip(("", "mem "), meters):
                        print('==>>> it: {}, {}avg. loss: {:.6f}, running {}acc: {:.3f}'.format(i, tag, loss_meter.avg(), "mem " if tag else "train ",
                                                                                             acc_meter.avg()))
            if pipeline:
                upd.update_finish(self.buffer, pending)
        if overlap:
            main.wait_stream(ds)
        self.after_train()
