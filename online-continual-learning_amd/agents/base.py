"""agents/base.py:14-227 — ContinualLearner: label bookkeeping, criterion (CE / SupCon), review trick, and
evaluate() with the nearest-class-mean classifier (NCM) or argmax, on the MI355X.

The labels trick and the separated softmax share one segmented cross-entropy kernel; the KD tricks use ocl_kd_fwd_bwd with a
parameter-snapshot teacher (kd_manager.py)."""
from abc import abstractmethod
import abc
import copy

import numpy as np
import torch

from .. import ops
from .. import debug
from ..loss import SupConLoss, cross_entropy_mean, cross_entropy_segmented_mean, unit_gradient
from ..kd_manager import KdManager
from ..utils import maybe_cuda, AverageMeter


class ContinualLearner(torch.nn.Module, metaclass=abc.ABCMeta):
    '''
    Abstract module which is inherited by each and every continual learning algorithm.
    '''

    def __init__(self, model, opt, params):
        super(ContinualLearner, self).__init__()
        self.params = params
        self.model = model
        self.opt = opt
        self.data = params.data
        self.cuda = params.cuda
        self.epoch = params.epoch
        self.batch = params.batch
        self.verbose = params.verbose
        self.old_labels = []
        self.new_labels = []
        self.task_seen = 0
        self.lbl_inv_map = {}
        self.class_task_map = {}
        self.kd_manager = KdManager()
        if getattr(params, 'error_analysis', False):
            raise NotImplementedError("error_analysis is outside the HIP hot path")

    _gc_frozen = False

    @classmethod
    def _freeze_long_lived_objects(cls):
        """Once per process, at the first training call: gc.freeze() moves everything alive (the imported packages' ~10^6 containers, the
        model, the replay memory's bookkeeping) to the collector's permanent generation.  Without it CPython's full (generation 2)
        collection walks all of them whenever its counters say so -- 105 - 133 ms in the middle of a training loop whose step is 3 ms
        (profiles/r6_aser_stall_probe.txt: the deterministic '+0.8 ms per step' of every fifth repeat of the ASER bench leg, at the same
        step of a fixed-seed run, was exactly one such collection).  Reference counting is untouched: frozen objects are still freed when
        their last reference goes, only cycle detection skips them; collections keep running over everything created afterwards.
        OCL_GC_FREEZE=0 turns it off."""
        if cls._gc_frozen:
            return
        cls._gc_frozen = True
        import gc
        import os
        if os.environ.get("OCL_GC_FREEZE", "1") != "0":
            gc.collect()
            gc.freeze()
        # (the young generations' collections do not show in the step: collector off / threshold 100000 change nothing, profiles/r6_gc_threshold_ab.txt)

    def before_train(self, x_train, y_train):
        """agents/base.py:43-50."""
        self._freeze_long_lived_objects()
        new_labels = list(set(y_train.tolist()))
        self.new_labels += new_labels
        for i, lbl in enumerate(new_labels):
            self.lbl_inv_map[lbl] = len(self.old_labels) + i

        for i in new_labels:
            self.class_task_map[i] = self.task_seen

    @abstractmethod
    def train_learner(self, x_train, y_train):
        pass

    def launch_stream(self):
        """Context for the training loop.  With launch-sequence replay requested (OCL_GRAPH=1, csrc/net.hip run_replayed) the loop runs
        on a stream of its own, ordered after and before the caller's stream: hipStreamBeginCapture is refused on the default stream,
        and the replay is what keeps a host-bound loop (ER + ASER: one synchronisation per step) from waiting on launch calls."""
        import contextlib
        import os

        @contextlib.contextmanager
        def ctx():
            # (the loop on a HIGH-priority stream of its own against the lowest-priority weight-gradient stream: no gain on SCR, ER + 0.02 ms --
            # profiles/r6_loop_prio_ab.txt; the switch is gone)
            if not (self.cuda and os.environ.get("OCL_GRAPH") == "1") or debug.on():
                yield
                return
            cur = torch.cuda.current_stream()
            if cur != torch.cuda.default_stream():
                yield
                return
            own = self.__dict__.get("_ocl_stream")
            if own is None:
                own = self.__dict__["_ocl_stream"] = torch.cuda.Stream()
            own.wait_stream(cur)
            with torch.cuda.stream(own):
                yield
            cur.wait_stream(own)
        return ctx()

    def after_train(self):
        """agents/base.py:56-91 (review trick branch :62-88)."""
        self.old_labels += self.new_labels
        self.new_labels_zombie = copy.deepcopy(self.new_labels)
        self.new_labels.clear()
        self.task_seen += 1
        if self.params.trick['review_trick'] and hasattr(self, 'buffer'):
            self.model.train()
            filled = self.buffer.current_index
            if filled > 0:
                from torch.utils.data import DataLoader
                from ..data import _IndexDataset
                bs = self.params.eps_mem_batch
                rv_loader = DataLoader(_IndexDataset(filled), batch_size=bs, shuffle=True, num_workers=0, drop_last=True)
                dev = self.buffer.buffer_img.device
                for ep in range(1):
                    for i, idx in enumerate(rv_loader):
                        idx = idx.to(dev)
                        batch_x = ops.gather_rows(self.buffer.buffer_img, idx)
                        batch_y = ops.gather_rows(self.buffer.buffer_label, idx)
                        if self.params.agent == 'SCR':
                            # the reference also runs one extra plain forward whose only effect is a BatchNorm
                            # running-stat update (base.py:77) before the two-view forwards (:78-80)
                            with torch.no_grad():
                                self.model.forward(batch_x)
                            feats = self.model.forward_views([batch_x, self.transform(batch_x)])
                            loss = self.criterion_views(feats, batch_y, 2)
                        else:
                            logits = self.model.forward(batch_x)
                            loss = self.criterion(logits, batch_y)
                        if debug.on():
                            debug.emit("review", indices=idx.cpu().numpy().copy(), loss=float(loss.detach()))
                        self.opt.zero_grad()
                        loss.backward(unit_gradient(loss))
                        self._step_scaled(0.1)   # grads / 10 (base.py:84-87)
        if self.params.trick['kd_trick'] or self.params.agent == 'LWF':   # base.py:90-91 (kd_trick_star alone never gets a teacher)
            self.kd_manager.update_teacher(self.model)

    def _step_scaled(self, scale):
        if hasattr(self.opt, "model"):
            self.opt.step(grad_scale=scale)
        else:
            for p in self.model.parameters():
                if p.requires_grad and p.grad is not None:
                    p.grad.data.mul_(scale)
            self.opt.step()

    def _host_labels(self, labels):
        h = getattr(labels, 'host', None)
        return np.asarray(h).astype(np.int64) if h is not None else labels.detach().cpu().numpy().astype(np.int64)

    def criterion(self, logits, labels):
        """agents/base.py:93-113.  The labels trick (:96-101) and the separated softmax (:102-108) are one kernel: a softmax
        over the logit columns of the row's label segment (ocl_ce_segmented_fwd_bwd); the segment table is built on the host
        from the labels' numpy mirror, as the reference builds `unq_lbls` / `lbl_inv_map`."""
        if self.params.trick['labels_trick']:
            y_host = self._host_labels(labels)
            seg = np.full(logits.shape[1], -1, dtype=np.int32)
            seg[np.unique(y_host)] = 0          # labels.unique().sort()[0]: the heads that appear in the batch
            return cross_entropy_segmented_mean(logits, labels, ops.upload(torch.from_numpy(seg), logits.device))
        elif self.params.trick['separated_softmax']:
            y_host = self._host_labels(labels)
            for lbl in y_host.tolist():
                self.lbl_inv_map[lbl]           # KeyError for a label outside old_labels + new_labels, as in the reference (:106)
            seg = np.full(logits.shape[1], -1, dtype=np.int32)
            seg[np.asarray(self.old_labels, dtype=np.int64)] = 0
            seg[np.asarray(self.new_labels, dtype=np.int64)] = 1
            return cross_entropy_segmented_mean(logits, labels, ops.upload(torch.from_numpy(seg), logits.device))
        # (the reference clones the labels first, agents/base.py:94, because its label tricks relabel in place; nothing here writes to them: no copy launch)
        if self.params.agent in ['SCR', 'SCP']:
            SC = SupConLoss(temperature=self.params.temp)
            return SC(logits, labels)
        else:
            return cross_entropy_mean(logits, labels)

    def criterion_views(self, feat_view_major, labels, n_views):
        """SupConLoss on the engine's native view-major [n_views*bsz, dim] layout (no permute / cat)."""
        SC = SupConLoss(temperature=self.params.temp)
        return SC.forward_view_major(feat_view_major, labels, n_views)

    def forward(self, x):
        return self.model.forward(x)

    def evaluate(self, test_loaders):
        """agents/base.py:118-227.  The reference extracts exemplar features one image at a time (:130-134);
        eval-mode BatchNorm makes the batched extraction below numerically equivalent per sample."""
        self.model.eval()
        acc_array = np.zeros(len(test_loaders))
        use_ncm = self.params.trick['ncm_trick'] or self.params.agent in ['ICARL', 'SCR', 'SCP']
        if use_ncm:
            buffer_filled = self.buffer.current_index
            dev = self.buffer.buffer_img.device
            labels = self.buffer.buffer_label[:buffer_filled].contiguous()
            # every buffered label must be a seen class (the reference would raise KeyError, base.py:126)
            host_labels = set(self.buffer.label_host[:buffer_filled].tolist())
            missing = host_labels - set(self.old_labels)
            if missing:
                raise KeyError(sorted(missing)[0])
            class_ids = torch.tensor(self.old_labels, dtype=torch.long, device=dev)
            with torch.no_grad():
                if buffer_filled > 0:
                    feats = self.model.features_batched(self.buffer.buffer_img[:buffer_filled])
                else:
                    feats = torch.zeros((0, self.model.feature_dim), device=dev)
                means, counts = ops.ncm_class_means(feats, labels, class_ids)
                counts_h = counts.cpu().numpy()
                for ci in range(len(self.old_labels)):   # dict order of cls_exemplar = old_labels order
                    if counts_h[ci] == 0:
                        # no exemplar: random mean (base.py:135-137), drawn on the CPU generator
                        mu_y = torch.normal(0, 1, size=(1, feats.shape[1])).squeeze()
                        mu_y = mu_y / mu_y.norm()
                        means[ci] = mu_y.to(dev)
        with torch.no_grad():
            for task, test_loader in enumerate(test_loaders):
                acc = AverageMeter()
                correct = []
                sizes = []
                seen_idx, seen_pred = [], []
                for i, (batch_x, batch_y) in enumerate(test_loader):
                    batch_x = maybe_cuda(batch_x, self.cuda)
                    batch_y = maybe_cuda(batch_y, self.cuda)
                    if use_ncm:
                        feature = self.model.features_batched(batch_x)  # (batch_size, feature_size)
                        pred = ops.ncm_predict(feature, means)
                        pred_label = ops.gather_rows(class_ids, pred)
                    else:
                        logits = self.model.forward(batch_x)
                        _, pred_label = torch.max(logits, 1)
                    correct.append((pred_label == batch_y).sum())
                    sizes.append(batch_y.size(0))
                    if debug.on():
                        seen_idx.append(np.asarray(test_loader.last_index_host).copy())
                        seen_pred.append(pred_label.cpu().numpy())
                if debug.on():
                    debug.emit("evaluate", task=task, index=np.concatenate(seen_idx) if seen_idx else np.zeros(0, dtype=np.int64),
                               pred=np.concatenate(seen_pred) if seen_pred else np.zeros(0, dtype=np.int64))
                if correct:
                    correct_h = torch.stack(correct).cpu().tolist()   # one sync per task loader
                    for c, n in zip(correct_h, sizes):
                        acc.update(c / n, n)
                acc_array[task] = acc.avg()
        print(acc_array)
        return acc_array
