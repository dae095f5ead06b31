"""Reduced_ResNet18 / SupConResNet with the reference's module and parameter names (models/resnet.py:14-37,
69-116,140-168), executed by the HIP engine (csrc/net.hip) instead of ATen.

The nn.Conv2d / nn.BatchNorm2d / nn.Linear objects below are *parameter containers only*: they give the
same construction order (hence the same torch-RNG draws and initial weights as the reference for a given
seed) and the same state_dict keys, but their forward() is never called.  On first use the parameters,
gradients and BatchNorm buffers are re-pointed into flat device arrays that the engine reads/writes in
place; the flat gradient is exactly the vector `get_grad_vector` builds (utils/buffer/buffer_utils.py:58-71).
"""
import ctypes as C

import contextlib

import torch
import torch.nn as nn

from . import ffi


def conv3x3(in_planes, out_planes, stride=1):
    return nn.Conv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=1, bias=False)


class BasicBlock(nn.Module):
    """Container mirroring models/resnet.py:14-37."""
    expansion = 1

    def __init__(self, in_planes, planes, stride=1):
        super().__init__()
        self.conv1 = conv3x3(in_planes, planes, stride)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = conv3x3(planes, planes)
        self.bn2 = nn.BatchNorm2d(planes)
        self.shortcut = nn.Sequential()
        if stride != 1 or in_planes != self.expansion * planes:
            self.shortcut = nn.Sequential(
                nn.Conv2d(in_planes, self.expansion * planes, kernel_size=1, stride=stride, bias=False),
                nn.BatchNorm2d(self.expansion * planes))

    def forward(self, x):
        raise RuntimeError("BasicBlock is a parameter container; the HIP engine runs the whole network")


class _NetFunction(torch.autograd.Function):
    """One autograd node for the whole network.  backward() writes the parameter gradients straight into the
    flat gradient buffer (overwrite or accumulate, as loss.backward() would) and returns no tensors for them."""

    @staticmethod
    def forward(ctx, x, anchor, owner, groups, frozen=False):
        out, slot, gen = owner._engine_train_forward(x, groups, save=True, frozen=frozen)
        ctx.owner, ctx.slot, ctx.gen = owner, slot, gen
        return out

    @staticmethod
    def backward(ctx, dout):
        ctx.owner._engine_backward(ctx.slot, ctx.gen, dout)
        return None, None, None, None, None


class _EngineMixin:
    """Binding between an nn.Module parameter container and an ocl_net engine object."""

    # per-model sizing knobs (set before first use)
    max_batch = None
    n_slots = 2

    def _engine_desc(self):
        raise NotImplementedError

    def _trunk(self):
        raise NotImplementedError

    def _init_engine_state(self):
        self._net = None
        self._flat = None
        self._gflat = None
        self._grads_fresh = True   # True: next backward overwrites (grads are logically zero / None)
        self._weights_dirty = True   # the parameter array was written (optimiser step, load, re-bind) since the engine last packed it
        self._same_weights = False   # inside `with model.same_weights():` -- see there
        self._packed_version = None
        self._slot_rr = 0
        self._slot_gen = {}
        self._anchor = None

    # ---- train()/eval(): the engine only looks at the root module's flag; nn.Module.train() walks all ~60 container
    # modules through __setattr__ (~1 ms per toggle, and ASER toggles six times per step: utils/utils.py:45-90).  The
    # flags of the children are still kept in step (state is observable), through their __dict__.
    def train(self, mode=True):
        if not isinstance(mode, bool):
            raise ValueError("training mode is expected to be boolean")
        mods = self.__dict__.get("_all_modules_cache")
        if mods is None:
            mods = list(self.modules())
            self.__dict__["_all_modules_cache"] = mods
        for m in mods:
            m.__dict__["training"] = mode
        return self

    def eval(self):
        return self.train(False)

    # ---- lazy binding --------------------------------------------------------------------------------------
    def _ensure_bound(self, device=None):
        if self._net is not None:
            return
        p0 = next(self.parameters())
        if not p0.is_cuda:
            raise RuntimeError("the model must be on the MI355X before use (call .cuda()); there is no CPU path")
        dev = p0.device
        ffi.init(dev.index if dev.index is not None else torch.cuda.current_device())
        L = ffi.lib()
        desc = self._engine_desc()
        h = ffi.vp(0)
        ffi.check(L.ocl_net_create(C.byref(desc), C.byref(h)), "net_create")
        self._net = h
        self._desc = desc
        self._n_tapes = desc.n_slots - 1   # the engine's last slot is the scratch slot of forwards that keep no tape
        n_params = L.ocl_net_param_count(h)
        # --- flat parameters / gradients: re-point every nn.Parameter into the flat arrays
        named = dict(self.named_parameters())
        flat = torch.empty(n_params, dtype=torch.float32, device=dev)
        gflat = torch.zeros(n_params, dtype=torch.float32, device=dev)
        nt = L.ocl_net_num_tensors(h)
        if nt != len(named):
            raise RuntimeError("engine/module parameter count mismatch: %d vs %d" % (nt, len(named)))
        name_buf = C.create_string_buffer(64)
        off = ffi.i64(0)
        ndim = ffi.i32(0)
        shape = (ffi.i64 * 4)()
        self._views = []
        order = [n for n, _ in self.named_parameters()]
        for i in range(nt):
            ffi.check(L.ocl_net_tensor_info(h, i, name_buf, C.byref(off), C.byref(ndim), shape), "tensor_info")
            name = name_buf.value.decode()
            if order[i] != name:
                raise RuntimeError("parameter order mismatch at %d: module has %s, engine has %s" % (i, order[i], name))
            p = named[name]
            shp = tuple(shape[k] for k in range(ndim.value))
            if tuple(p.shape) != shp:
                raise RuntimeError("parameter %s: module shape %s, engine shape %s" % (name, tuple(p.shape), shp))
            view = flat[off.value: off.value + p.numel()].view(shp)
            view.copy_(p.data)
            p.data = view
            gview = gflat[off.value: off.value + p.numel()].view(shp)
            self._views.append((p, gview))
        # --- BatchNorm buffers
        n_stats = L.ocl_net_bn_stat_count(h)
        nbn = L.ocl_net_num_bn(h)
        running = torch.empty(n_stats, dtype=torch.float32, device=dev)
        nbt = torch.zeros(nbn, dtype=torch.int64, device=dev)
        modules = dict(self.named_modules())
        ch = ffi.i32(0)
        for i in range(nbn):
            ffi.check(L.ocl_net_bn_info(h, i, name_buf, C.byref(off), C.byref(ch)), "bn_info")
            bn = modules[name_buf.value.decode()]
            c = ch.value
            rm = running[off.value: off.value + c]
            rv = running[off.value + c: off.value + 2 * c]
            rm.copy_(bn.running_mean)
            rv.copy_(bn.running_var)
            nbt[i] = bn.num_batches_tracked.to(dev)
            bn._buffers["running_mean"] = rm
            bn._buffers["running_var"] = rv
            bn._buffers["num_batches_tracked"] = nbt[i]
        ws_bytes = L.ocl_net_workspace_bytes(h)
        ws = torch.empty(ws_bytes + 256, dtype=torch.uint8, device=dev)
        shift = (-ws.data_ptr()) % 256
        self._ws = ws
        self._flat, self._gflat, self._running, self._nbt = flat, gflat, running, nbt
        ffi.check(L.ocl_net_bind(h, ffi.ptr(flat), ffi.ptr(gflat), ffi.ptr(running), ffi.ptr(nbt),
                                 ffi.vp(ws.data_ptr() + shift), ws_bytes), "net_bind")
        self._anchor = torch.zeros(1, dtype=torch.float32, device=dev, requires_grad=True)
        self.feature_dim = L.ocl_net_feature_dim(h)
        self.out_dim = L.ocl_net_out_dim(h)

    def __del__(self):
        try:
            if getattr(self, "_net", None) is not None:
                ffi.lib().ocl_net_destroy(self._net)
                self._net = None
        except Exception:
            pass

    # ---- gradient bookkeeping ------------------------------------------------------------------------------
    def flat_params(self):
        self._ensure_bound()
        return self._flat

    def flat_grads(self):
        self._ensure_bound()
        return self._gflat

    def mark_weights_written(self):
        """The bound parameter array was written through the engine (FusedSGD.step): the next forward re-packs."""
        self._weights_dirty = True

    @contextlib.contextmanager
    def same_weights(self):
        """Inside the block, a forward that follows another forward of the block with no write to the parameters in between carries
        OCL_FWD_SAME_WEIGHTS: the engine reuses the weight packs it made for the earlier one (agents/exp_replay.py: the ASER update's
        feature pass, the batch pass, the retrieval's feature pass, the memory pass and the combined pass between two optimiser steps).
        "No write" is checked, not assumed: FusedSGD.step() reports itself, and every torch in-place write to a parameter or to the
        flat array (load_state_dict, copy_, mul_ ...) moves a version counter that is compared here.  What the counters cannot see is
        a write through `p.data` -- do not do that inside such a block (nothing in this package does)."""
        prev, self._same_weights = self._same_weights, True
        try:
            yield
        finally:
            self._same_weights = prev

    def _weights_version(self):
        v = self._flat._version
        for p, _ in self._views:
            v += p._version
        return v

    def _weights_flag(self, params_override):
        """OCL_FWD_SAME_WEIGHTS for the forward being issued, and the bookkeeping behind it: `_weights_dirty` (a step since the last
        pack) and `_packed_version` (the parameters' version counters at the last forward of a same_weights block; None = unknown)."""
        if params_override is not None:
            return 0          # (the engine notices by itself that its arena holds another array's packs afterwards)
        if not self._same_weights:
            self._packed_version, self._weights_dirty = None, False
            return 0
        v = self._weights_version()
        # (a pass of the block that does pack writes the data-gradient packs too: a taped pass on the same weights follows in the block)
        flag = ffi.FWD_SAME_WEIGHTS if (not self._weights_dirty and self._packed_version == v) else ffi.FWD_PACK_ALL
        self._packed_version, self._weights_dirty = v, False
        return flag

    def mark_grads_zero(self):
        """zero_grad() without touching memory: the next backward overwrites."""
        self._grads_fresh = True

    def _attach_grads(self):
        fresh = self._grads_fresh
        for p, gview in self._views:
            if p.grad is None:
                fresh = True   # torch's zero_grad(set_to_none=True): None means zero
            if p.grad is not gview:
                p.grad = gview
        return fresh

    # ---- engine calls --------------------------------------------------------------------------------------
    def _check_input(self, x):
        if not x.is_cuda or x.dtype != torch.float32 or x.dim() != 4 or x.shape[1] != 3:
            raise RuntimeError("expected a float32 [n,3,H,W] tensor on the GPU, got %s %s on %s" % (x.dtype, tuple(x.shape), x.device))
        if x.shape[2] != self._desc.in_h or x.shape[3] != self._desc.in_w:
            raise RuntimeError("engine built for %dx%d inputs, got %dx%d" % (self._desc.in_h, self._desc.in_w, x.shape[2], x.shape[3]))
        return x.contiguous()

    def _segments(self, x):
        """x: one [n,3,H,W] tensor or a sequence of them (the pieces of a batch the reference would torch.cat: memory rows, stream batch,
        augmented views).  Returns (tensors kept alive, total n, pointer array, size array) for ocl_net_forward_segments: the engine's
        layout conversion reads the pieces where they are, no concatenated copy is made."""
        parts = [self._check_input(t) for t in (x if isinstance(x, (list, tuple)) else (x,)) if t.shape[0] > 0]
        if not parts or len(parts) > 8:
            raise RuntimeError("forward: 1 .. 8 non-empty input tensors, got %d" % len(parts))
        n = sum(t.shape[0] for t in parts)
        if n > self._desc.max_batch:
            raise RuntimeError("batch %d exceeds the engine's max_batch %d" % (n, self._desc.max_batch))
        ptrs = (ffi.vp * len(parts))(*[t.data_ptr() for t in parts])
        sizes = (ffi.i32 * len(parts))(*[t.shape[0] for t in parts])
        return parts, n, ptrs, sizes

    def _tape_slot(self):
        """Next activation tape (round robin over the model's n_slots); a generation counter detects overwritten tapes."""
        slot = self._slot_rr
        self._slot_rr = (self._slot_rr + 1) % self._n_tapes
        gen = self._slot_gen.get(slot, 0) + 1
        self._slot_gen[slot] = gen
        return slot, gen

    def _engine_train_forward(self, x, groups, save, params_override=None, update_running=True, want_feat=False, frozen=False):
        self._ensure_bound()
        parts, n, ptrs, sizes = self._segments(x)
        dev = parts[0].device
        # forwards that keep no tape (MIR's scoring passes, the KD teacher, no_grad passes) run in a scratch slot of their own, so
        # that any number of them may sit between a taped forward and its backward
        slot, gen = self._tape_slot() if save else (self._n_tapes, 0)
        out = torch.empty((n, self.out_dim), dtype=torch.float32, device=dev)
        feat = torch.empty((n, self.feature_dim), dtype=torch.float32, device=dev) if want_feat else None
        flags = ffi.FWD_TRAIN | (ffi.FWD_SAVE_TAPE if save else 0) | (ffi.FWD_UPDATE_RUNNING if update_running else 0)
        if frozen:   # model.eval() under autograd: running statistics, activations kept for backward
            flags = ffi.FWD_SAVE_TAPE | ffi.FWD_FROZEN_BN
        flags |= self._weights_flag(params_override)
        ffi.check(ffi.lib().ocl_net_forward_segments(self._net, ptrs, sizes, len(parts), groups, flags, ffi.ptr(params_override), ffi.ptr(feat),
                                                     ffi.ptr(out), slot, ffi.stream()), "net_forward(train)")
        if want_feat:
            return out, feat
        return out, slot, gen

    def forward_stats_only(self, x):
        """A train-mode pass run for its BatchNorm running-statistic updates alone (the batch and memory passes of ER + ASER, whose
        logits and gradients the reference throws away, agents/exp_replay.py:49-76): no head, no pooled features, no tape, no output."""
        self._ensure_bound()
        if not self.training:
            raise RuntimeError("forward_stats_only is a train-mode pass")
        parts, n, ptrs, sizes = self._segments(x)
        flags = ffi.FWD_TRAIN | ffi.FWD_UPDATE_RUNNING | self._weights_flag(None)
        ffi.check(ffi.lib().ocl_net_forward_segments(self._net, ptrs, sizes, len(parts), 1, flags, None, None, None, self._n_tapes, ffi.stream()),
                  "net_forward(stats only)")

    def _engine_eval_forward(self, x, want_out=True, want_feat=False, params_override=None):
        self._ensure_bound()
        parts, n, ptrs, sizes = self._segments(x)
        dev = parts[0].device
        out = torch.empty((n, self.out_dim), dtype=torch.float32, device=dev) if want_out else None
        feat = torch.empty((n, self.feature_dim), dtype=torch.float32, device=dev) if want_feat else None
        ffi.check(ffi.lib().ocl_net_forward_segments(self._net, ptrs, sizes, len(parts), 1, self._weights_flag(params_override), ffi.ptr(params_override),
                                                     ffi.ptr(feat), ffi.ptr(out), self._n_tapes, ffi.stream()), "net_forward(eval)")
        return out, feat

    def _engine_backward(self, slot, gen, dout):
        if self._slot_gen.get(slot) != gen:
            raise RuntimeError("backward through a forward whose activations were overwritten: more than %d forward "
                               "passes are alive at once (raise model.n_slots before first use)" % self._n_tapes)
        fresh = self._attach_grads()
        dout = dout.contiguous()
        if dout.dtype != torch.float32:
            raise RuntimeError("gradient must be float32")
        ffi.check(ffi.lib().ocl_net_backward(self._net, slot, ffi.ptr(dout), 0 if fresh else 1, ffi.stream()), "net_backward")
        self._grads_fresh = False

    # ---- public forward paths ------------------------------------------------------------------------------
    def _forward_out(self, x, groups=1):
        self._ensure_bound()
        if self.training:
            if torch.is_grad_enabled():
                return _NetFunction.apply(x, self._anchor, self, groups)
            out, _, _ = self._engine_train_forward(x, groups, save=False)
            return out
        if torch.is_grad_enabled() and groups == 1:
            # eval mode under autograd (utils/buffer/gss_greedy_update.py:16,77-79): BatchNorm uses its running statistics and the
            # pass is differentiable w.r.t. every parameter
            return _NetFunction.apply(x, self._anchor, self, 1, True)
        out, _ = self._engine_eval_forward(x, want_out=True)
        return out

    def forward_views(self, views):
        """SCR: the reference calls model.forward once per view (agents/scr.py:55), i.e. BatchNorm statistics are
        per view.  Here all views run as ONE batched pass with per-group statistics; returns [n_views*bsz, out]
        view-major.  A view may itself be a sequence of tensors (memory rows, stream batch): the pieces are read where they are
        (ocl_net_forward_segments), nothing is concatenated."""
        parts, sizes = [], []
        for v in views:
            vs = list(v) if isinstance(v, (list, tuple)) else [v]
            parts += vs
            sizes.append(sum(t.shape[0] for t in vs))
        if len(set(sizes)) != 1:
            raise RuntimeError("forward_views: the views must hold the same number of images, got %s" % sizes)
        return self._forward_out(tuple(parts), groups=len(views))

    def forward_views_taped(self, views):
        """forward_views for a caller that computes dL/dout itself: (out, tape) without an autograd node; backward_taped(tape, dout)
        runs the backward of exactly this pass into the flat gradient (overwrite / accumulate as loss.backward() would).  Saves the
        slice / add / fill launches autograd spends on `both[:n]`, `both[n:]` (agents/exp_replay.py: the merged ER step)."""
        self._ensure_bound()
        if not self.training:
            raise RuntimeError("forward_views_taped is a train-mode pass")
        parts, sizes = [], []
        for v in views:
            vs = list(v) if isinstance(v, (list, tuple)) else [v]
            parts += vs
            sizes.append(sum(t.shape[0] for t in vs))
        if len(set(sizes)) != 1:
            raise RuntimeError("forward_views_taped: the views must hold the same number of images, got %s" % sizes)
        out, slot, gen = self._engine_train_forward(tuple(parts), len(views), save=True)
        return out, (slot, gen)

    def backward_taped(self, tape, dout):
        self._engine_backward(tape[0], tape[1], dout)

    def forward_with_params(self, x, flat_params):
        """no-grad forward of a virtual model (MIR's theta - lr*grad, mir_retrieve.py:21,25) in the current mode,
        without touching this model's BatchNorm running statistics (the reference updates the deepcopy's)."""
        self._ensure_bound()
        if self.training:
            out, _, _ = self._engine_train_forward(x, 1, save=False, params_override=flat_params, update_running=False)
            return out
        out, _ = self._engine_eval_forward(x, want_out=True, params_override=flat_params)
        return out

    def _features(self, x):
        self._ensure_bound()
        if self.training:
            if torch.is_grad_enabled():
                raise RuntimeError("features() with autograd in train mode is not on the replay hot path; "
                                   "wrap in torch.no_grad() or call model.eval()")
            _, feat = self._engine_train_forward(x, 1, save=False, want_feat=True)
            return feat
        _, feat = self._engine_eval_forward(x, want_out=False, want_feat=True)
        return feat

    def features_batched(self, x, chunk=None):
        """features() over an arbitrarily large batch in engine-sized chunks (eval-mode BN is per-sample, so
        chunking does not change results: mini_batch_deep_features uses 64, evaluate() uses 1).  x may be a sequence of tensors
        (the pieces of the reference's torch.cat((eval_x, cand_x)), utils/buffer/aser_utils.py:73) when they fit one pass."""
        self._ensure_bound()
        chunk = chunk or self._desc.max_batch
        if isinstance(x, (list, tuple)):
            x = [t for t in x if t.shape[0] > 0]
            if sum(t.shape[0] for t in x) <= chunk and 1 <= len(x) <= 8:
                return self._features(tuple(x))
            x = torch.cat(list(x), 0)
        outs = [self._features(x[i:i + chunk]) for i in range(0, x.shape[0], chunk)]
        return outs[0] if len(outs) == 1 else torch.cat(outs, 0)


class ResNet(nn.Module, _EngineMixin):
    """models/resnet.py:69-109 with BasicBlock; `Reduced_ResNet18` = ResNet([2,2,2,2], nclasses, nf=20)."""

    def __init__(self, num_blocks, num_classes, nf, bias, in_hw=(32, 32)):
        super().__init__()
        if list(num_blocks) != [2, 2, 2, 2] or not bias:
            raise NotImplementedError("the HIP engine implements Reduced_ResNet18 (BasicBlock [2,2,2,2], bias=True)")
        self.in_planes = nf
        self.nf = nf
        self.in_hw = tuple(in_hw)
        self.conv1 = conv3x3(3, nf * 1)
        self.bn1 = nn.BatchNorm2d(nf * 1)
        self.layer1 = self._make_layer(nf * 1, num_blocks[0], stride=1)
        self.layer2 = self._make_layer(nf * 2, num_blocks[1], stride=2)
        self.layer3 = self._make_layer(nf * 4, num_blocks[2], stride=2)
        self.layer4 = self._make_layer(nf * 8, num_blocks[3], stride=2)
        self.linear = nn.Linear(nf * 8, num_classes, bias=bias)
        self._init_engine_state()

    def _make_layer(self, planes, num_blocks, stride):
        strides = [stride] + [1] * (num_blocks - 1)
        layers = []
        for s in strides:
            layers.append(BasicBlock(self.in_planes, planes, s))
            self.in_planes = planes
        return nn.Sequential(*layers)

    def _engine_desc(self):
        h, w = self.in_hw
        mb = self.max_batch or (512 if h * w <= 32 * 32 else 256)
        return ffi.NetDesc(h, w, self.nf, self.linear.out_features, 0, 0, mb, self.n_slots + 1)   # + the scratch slot

    def features(self, x):
        '''Features before FC layers'''
        return self._features(x)

    def logits(self, x):
        '''Apply the last FC linear mapping to get logits'''
        from . import ops
        return ops.gemm_small(x, self.linear.weight.data, bias=self.linear.bias.data, trans_b=True)

    def forward(self, x):
        return self._forward_out(x)


def Reduced_ResNet18(nclasses, nf=20, bias=True, in_hw=(32, 32)):
    """Reduced ResNet18 as in GEM MIR (nf=20), models/resnet.py:112-116."""
    return ResNet([2, 2, 2, 2], nclasses, nf, bias, in_hw=in_hw)


class SupConResNet(nn.Module, _EngineMixin):
    """backbone + projection head (models/resnet.py:140-168)"""

    def __init__(self, dim_in=160, head='mlp', feat_dim=128, in_hw=(32, 32)):
        super().__init__()
        self.encoder = Reduced_ResNet18(100, in_hw=in_hw)
        self.head_kind = head
        self.feat_dim = feat_dim
        if head == 'linear':
            self.head = nn.Linear(dim_in, feat_dim)
        elif head == 'mlp':
            self.head = nn.Sequential(nn.Linear(dim_in, dim_in), nn.ReLU(inplace=True), nn.Linear(dim_in, feat_dim))
        elif head == 'None':
            self.head = None
        else:
            raise NotImplementedError('head not supported: {}'.format(head))
        self._init_engine_state()

    def _engine_desc(self):
        h, w = self.encoder.in_hw
        kind = {'mlp': 1, 'linear': 2, 'None': 3}[self.head_kind]
        mb = self.max_batch or (512 if h * w <= 32 * 32 else 256)
        return ffi.NetDesc(h, w, self.encoder.nf, 100, kind, self.feat_dim, mb, self.n_slots + 1)   # + the scratch slot

    def forward(self, x):
        return self._forward_out(x)

    def features(self, x):
        return self._features(x)


# nn.Module precedes the mixin in the MRO: install the fast train()/eval() explicitly
for _cls in (ResNet, SupConResNet):
    _cls.train = _EngineMixin.train
    _cls.eval = _EngineMixin.eval
