"""agents/scr.py:11-69 — Supervised Contrastive Replay inner loop.

Per iteration: retrieve -> concat -> augment -> two views through SupConResNet -> SupCon loss -> SGD ->
reservoir update.  The two views run as one batched pass with per-view BatchNorm statistics (the reference
calls model.forward twice, scr.py:55)."""
import contextlib
import math
import os

import torch

from .. import ops
from .. import debug
from ..buffer import Buffer
from ..data import DeviceLoader
from ..setup_elements import input_size_match
from ..utils import maybe_cuda, AverageMeter
from ..loss import unit_gradient
from .base import ContinualLearner


class ScrAugment(object):
    """Stand-in for the kornia pipeline of scr.py:18-24: RandomResizedCrop(scale=(0.2,1)) -> RandomHorizontalFlip ->
    ColorJitter(0.4,0.4,0.4,0.1,p=0.8) -> RandomGrayscale(p=0.2).  kornia 0.4.1's RNG parameterisation is
    unpinned: parameters are drawn here on the torch CPU generator and applied by one HIP kernel."""

    def __init__(self, size, scale=(0.2, 1.0), ratio=(3. / 4., 4. / 3.), jitter=(0.4, 0.4, 0.4, 0.1), p_jitter=0.8,
                 p_gray=0.2):
        self.h, self.w = size
        self.scale, self.ratio, self.jitter = scale, ratio, jitter
        self.p_jitter, self.p_gray = p_jitter, p_gray

    TRIES = 10

    def draw(self, n):
        """All random numbers of one call: ONE torch.rand on the CPU generator, [n, 30]."""
        return torch.rand(n, 2 * self.TRIES + 10)

    def fallback_crop(self):
        """Centre crop used when none of the 10 attempts fits: whole image, aspect ratio clamped into `ratio` (torchvision get_params)."""
        h, w = float(self.h), float(self.w)
        in_ratio = w / h
        if in_ratio < self.ratio[0]:
            return w, round(w / self.ratio[0])
        if in_ratio > self.ratio[1]:
            return round(h * self.ratio[1]), h
        return w, h

    def config(self):
        """The 12 numbers ocl_scr_augment_uniform takes (include/ocl_hip.h)."""
        fw, fh = self.fallback_crop()
        return [self.scale[0], self.scale[1], self.ratio[0], self.ratio[1], self.jitter[0], self.jitter[1], self.jitter[2],
                self.jitter[3], self.p_jitter, self.p_gray, fw, fh]

    def sample_params(self, n):
        return self.params_from_uniform(self.draw(n))

    def params_from_uniform(self, u):
        """Host statement of the parameter arithmetic (the device kernel `aug_params_kernel` is tested against it).
        Crop boxes the way torchvision / kornia draw them (RandomResizedCrop.get_params; kornia 0.4.1 random_crop_size_generator):
        up to 10 attempts of (area ~ U(scale) * H * W, log-ratio ~ U(log ratio)), integer width / height, the first attempt that
        fits inside the image wins; if none fits, the centre crop with the aspect ratio clamped into `ratio`.  The position is
        uniform over the placements that keep the box inside the image.  All draws come from the torch CPU generator."""
        h, w = float(self.h), float(self.w)
        tries = self.TRIES
        n = u.shape[0]
        area = (self.scale[0] + (self.scale[1] - self.scale[0]) * u[:, :tries]) * h * w
        logr = math.log(self.ratio[0]) + (math.log(self.ratio[1]) - math.log(self.ratio[0])) * u[:, tries:2 * tries]
        r = torch.exp(logr)
        cw_try = torch.round(torch.sqrt(area * r))
        ch_try = torch.round(torch.sqrt(area / r))
        fits = (cw_try > 0) & (cw_try <= w) & (ch_try > 0) & (ch_try <= h)
        first = torch.where(fits.any(1), fits.float().argmax(1), torch.zeros(n, dtype=torch.long))
        cw = cw_try.gather(1, first[:, None]).squeeze(1)
        ch = ch_try.gather(1, first[:, None]).squeeze(1)
        none = ~fits.any(1)
        if none.any():
            fw, fh = self.fallback_crop()
            cw = torch.where(none, torch.full_like(cw, float(fw)), cw)
            ch = torch.where(none, torch.full_like(ch, float(fh)), ch)
        e = u[:, 2 * tries:]
        y0 = torch.floor(e[:, 0] * (h - ch + 1)).clamp(max=h - 1)
        x0 = torch.floor(e[:, 1] * (w - cw + 1)).clamp(max=w - 1)
        y0 = torch.where(none, torch.floor((h - ch) / 2), y0)
        x0 = torch.where(none, torch.floor((w - cw) / 2), x0)
        b, c, s, hue = self.jitter
        p = torch.empty(n, ops.AUG_NPARAM)
        p[:, 0], p[:, 1], p[:, 2], p[:, 3] = y0, x0, ch, cw
        p[:, 4] = (e[:, 2] < 0.5).float()
        p[:, 5] = (e[:, 3] < self.p_jitter).float()
        p[:, 6] = 1.0 - b + 2 * b * e[:, 4]
        p[:, 7] = 1.0 - c + 2 * c * e[:, 5]
        p[:, 8] = 1.0 - s + 2 * s * e[:, 6]
        p[:, 9] = -hue + 2 * hue * e[:, 7]
        p[:, 10] = torch.floor(e[:, 8] * 24).clamp(0, 23)
        p[:, 11] = (e[:, 9] < self.p_gray).float()
        return p

    def apply_parts(self, parts):
        """The augmented view of torch.cat(parts) without the concatenation: ONE draw for all rows (the generator sees exactly what
        __call__(torch.cat(parts)) would make it see), one kernel per piece writing its rows of one output tensor."""
        n = sum(p.shape[0] for p in parts)
        u = ops.upload(self.draw(n), parts[0].device)
        out = torch.empty((n,) + tuple(parts[0].shape[1:]), dtype=torch.float32, device=parts[0].device)
        cfg, off = self.config(), 0
        for p in parts:
            k = p.shape[0]
            if k:
                ops.scr_augment_uniform(p, u[off:off + k], cfg, out=out[off:off + k])
            off += k
        return out

    def __call__(self, x):
        if isinstance(x, (tuple, list)):   # the pieces of a batch: one draw, one output tensor, no concatenated input
            return self.apply_parts(x)
        # the draws on the host generator (one call, as before); everything derived from them on the device: the ~40 small CPU
        # tensor ops of params_from_uniform cost 0.25 ms of host time per step
        u = ops.upload(self.draw(x.shape[0]), x.device)
        if debug.on():
            out, params = ops.scr_augment_uniform(x, u, self.config(), want_params=True)
            debug.emit("scr_augment", params=params.cpu().numpy())
            return out
        return ops.scr_augment_uniform(x, u, self.config())


class SupContrastReplay(ContinualLearner):
    def __init__(self, model, opt, params):
        super(SupContrastReplay, self).__init__(model, opt, params)
        self.buffer = Buffer(model, params)
        self.mem_size = params.mem_size
        self.eps_mem_batch = params.eps_mem_batch
        self.mem_iters = params.mem_iters
        self.transform = ScrAugment(size=(input_size_match[self.params.data][1], input_size_match[self.params.data][2]),
                                    scale=(0.2, 1.))

    def train_learner(self, x_train, y_train):
        with self.launch_stream():
            self._train_learner(x_train, y_train)

    def _train_learner(self, x_train, y_train):
        self.before_train(x_train, y_train)
        # set up loader
        train_loader = DeviceLoader(x_train, y_train, self.batch, shuffle=True, drop_last=True)
        # set up model
        self.model = self.model.train()

        # setup tracker
        losses = AverageMeter()
        acc_batch = AverageMeter()

        # Data path on its own stream (random retrieve / reservoir update only: they never touch the model).  Same statements, same
        # order, same RNG draws as the reference's loop; only the stream they are issued on differs, so that the gather / concat /
        # augment / scatter launches of step i+1 (~20 tiny launches, ~0.15 ms back to back) run next to step i's backward instead
        # of in front of step i+1's forward.
        overlap = (self.cuda and self.params.retrieve == 'random' and self.params.update == 'random' and not debug.on()
                   and os.environ.get("OCL_DATA_STREAM", "1") != "0")
        main = torch.cuda.current_stream() if self.cuda else None
        ds = ops.data_stream(x_train.device if torch.is_tensor(x_train) and x_train.is_cuda else torch.cuda.current_device()) if overlap else None
        if overlap:
            ds.wait_stream(main)

        def on_data():
            return torch.cuda.stream(ds) if overlap else contextlib.nullcontext()

        for ep in range(self.epoch):
            loader_it = iter(train_loader)
            i = -1
            while True:
                with on_data():
                    batch_data = next(loader_it, None)
                if batch_data is None:
                    break
                i += 1
                # batch update
                batch_x, batch_y = batch_data
                batch_y_host = train_loader.last_y_host

                for j in range(self.mem_iters):
                    with on_data():
                        mem_x, mem_y = self.buffer.retrieve(x=batch_x, y=batch_y)

                    if mem_x.size(0) > 0:
                        with on_data():
                            mem_x = maybe_cuda(mem_x, self.cuda)
                            mem_y = maybe_cuda(mem_y, self.cuda)
                            combined_labels = torch.cat((mem_y, batch_y))
                            if isinstance(self.transform, ScrAugment) and not debug.on():
                                # torch.cat((mem_x, batch_x)) is not materialised: the augmentation and the engine's layout conversion
                                # read the two pieces where they are (ScrAugment takes the pieces as a tuple)
                                first_view = (mem_x, batch_x)
                                aug = self.transform(first_view)
                            else:
                                first_view = (torch.cat((mem_x, batch_x)),)
                                aug = self.transform(first_view[0])
                            second_view = tuple(aug) if isinstance(aug, (tuple, list)) else (aug,)
                        if overlap:
                            main.wait_stream(ds)
                            for t in first_view + second_view + (combined_labels,):
                                t.record_stream(main)   # allocated on the data stream, consumed on the main one
                        features = self.model.forward_views([first_view, second_view])
                        loss = self.criterion_views(features, combined_labels, 2)
                        if self.verbose:
                            losses.update(loss, batch_y.size(0))
                        if debug.on():
                            debug.emit("scr_loss", loss=float(loss.detach()))
                        self.opt.zero_grad()
                        loss.backward(unit_gradient(loss))
                        self.opt.step()

                # update mem
                with on_data():
                    self.buffer.update(batch_x, batch_y, y_host=batch_y_host)
                if i % 100 == 1 and self.verbose:
                        print(
                            '==>>> it: {}, avg. loss: {:.6f}, '
                                .format(i, losses.avg(), acc_batch.avg())
                        )
        if overlap:
            main.wait_stream(ds)
        self.after_train()
