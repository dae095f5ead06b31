"""utils/buffer/buffer_utils.py: random_retrieve (:9-26), ClassBalancedRandomSampling (:74-160), get_grad_vector
(:58-71) — host bookkeeping kept in Python (numpy global RNG, torch CPU RNG, CPython set/dict order are part of
the reference's observable behaviour), data movement done by the HIP gather kernels."""
from collections import defaultdict

import numpy as np
import torch

from .. import ops
from .. import debug


try:
    from .. import _hostc
except ImportError:          # not built (csrc/Makefile builds it next to libocl_hip.so): the Python statement of the loop is used
    _hostc = None
_hostc_ok = None


def _hostc_usable():
    """The C helper restates torch's CPU generator; it is trusted only after its permutations have been compared with the
    installed torch's on a scratch copy of the generator state (once per process)."""
    global _hostc_ok
    if _hostc_ok is None:
        ok = _hostc is not None
        if ok:
            saved = torch.get_rng_state()
            try:
                sizes = [0, 1, 2, 3, 7, 50, 51, 700, 1, 0, 13]
                want = [torch.randperm(n).tolist() for n in sizes]
                after = torch.get_rng_state()
                scratch = saved.clone()
                got = [_hostc.randperm_check(scratch.numpy(), n) for n in sizes]
                ok = got == want and torch.equal(scratch, after)
            finally:
                torch.set_rng_state(saved)
            if not ok:
                import warnings
                warnings.warn("ocl_amd: _hostc's restatement of torch.randperm disagrees with this torch build; "
                              "ClassBalancedRandomSampling uses the Python loop")
        _hostc_ok = ok
    return _hostc_ok


_setdiff_ok = None
_EMULATE = __import__("os").environ.get("OCL_CBRS_EMULATE", "1") != "0"   # (0: the draw with exclusions builds its sets, A/B)


def _setdiff_emulation_ok():
    """The C helper can produce the iteration order of `slots - excluded` without building that set (a simulation of CPython's set table,
    csrc/hostc.c: emu_difference).  It is trusted only after it has reproduced list(a - b) on a few hundred sets of this interpreter --
    fresh and churned, across the resize thresholds -- once per process; otherwise the real set operation is used (as before round 6)."""
    global _setdiff_ok
    if _setdiff_ok is None:
        ok = _hostc is not None and hasattr(_hostc, "setdiff_check")
        if ok:
            import random
            rnd = random.Random(20251)
            seen = 0
            for trial in range(400):
                n = rnd.choice([0, 1, 4, 5, 18, 19, 20, 49, 50, 76, 77, 90, 200, 306, 307, 330])
                a = set(rnd.sample(range(5000), n))
                for _ in range(rnd.choice([0, 10, 120])):     # remove / add: dummies, tables larger than a fresh set's
                    if a and rnd.random() < 0.5:
                        a.discard(rnd.choice(tuple(a)))
                    else:
                        a.add(rnd.randrange(5000))
                b = set(rnd.sample(range(5000), rnd.choice([1, 30, 100, 150])))
                if a:
                    b |= set(rnd.sample(tuple(a), min(len(a), rnd.randrange(0, 4))))
                got = _hostc.setdiff_check(a, b)
                if got is None:
                    continue
                seen += 1
                if got != list(a - b):
                    ok = False
                    break
            ok = ok and seen >= 100
            if not ok:
                import warnings
                warnings.warn("ocl_amd: _hostc's simulation of CPython's set difference disagrees with this interpreter; the class-balanced "
                              "draw with exclusions builds the sets")
        _setdiff_ok = ok
    return _setdiff_ok


def _host_labels(y, y_host=None):
    if y_host is not None:
        return np.asarray(y_host).astype(np.int64)
    return y.detach().cpu().numpy().astype(np.int64)


def random_retrieve(buffer, num_retrieve, excl_indices=None, return_indices=False):
    """buffer_utils.py:9-26: uniform sample without replacement of the filled slots (numpy global RNG)."""
    filled_indices = np.arange(buffer.current_index)
    if excl_indices is not None:
        excl_indices = list(excl_indices)
    else:
        excl_indices = []
    # setdiff1d returns the sorted unique survivors: with nothing excluded that is `filled_indices` itself (two sorts saved)
    valid_indices = np.setdiff1d(filled_indices, np.array(excl_indices)) if len(excl_indices) else filled_indices
    num_retrieve = min(num_retrieve, valid_indices.shape[0])
    indices = torch.from_numpy(np.random.choice(valid_indices, num_retrieve, replace=False)).long()

    debug.emit("random_retrieve", indices=indices.numpy().copy())
    x, y = ops.gather_pair(buffer.buffer_img, buffer.buffer_label, indices)   # (index upload + both gathers: one call, one launch)
    y.host = buffer.label_host[indices.numpy()] if num_retrieve else np.zeros(0, dtype=np.int64)

    if return_indices:
        return x, y, indices
    else:
        return x, y


def match_retrieve(buffer, cur_y, exclud_idx=None):
    """buffer_utils.py:29-49: for every item of the batch one buffered sample of the same class, drawn per class with Python's
    global `random.sample` from the tracker's slot set (CPython set order is part of the behaviour); two empty tensors when some
    class of the batch has too few slots.  Labels come from the numpy mirror the loader attaches (`cur_y.host`)."""
    import random
    from collections import Counter
    ys = _host_labels(cur_y, getattr(cur_y, "host", None)).tolist()
    per_class = Counter(ys)
    positions = defaultdict(list)
    for pos, label in enumerate(ys):
        positions[label].append(pos)
    chosen = [None] * len(ys)
    for label in per_class:
        slots = buffer.buffer_tracker.class_index_cache[label]
        if exclud_idx is not None:
            slots = slots - set(exclud_idx.tolist())
        if not slots or len(slots) < per_class[label]:
            print('match retrieve attempt fail')
            return torch.tensor([]), torch.tensor([])
        for pos, slot in zip(positions[label], random.sample(list(slots), per_class[label])):
            chosen[pos] = slot
    indices = torch.tensor(chosen)
    debug.emit("match_retrieve", indices=indices.numpy().copy())
    idx_dev = ops.upload(indices, buffer.buffer_img.device)
    x = ops.gather_rows(buffer.buffer_img, idx_dev)
    y = ops.gather_rows(buffer.buffer_label, idx_dev)
    y.host = buffer.label_host[indices.numpy()]
    return x, y


def get_grad_vector(model):
    """buffer_utils.py:58-71: the flat gradient vector (zeros where a parameter has no gradient).  The engine's
    flat gradient array already has that layout; before any backward it is logically zero."""
    g = model.flat_grads()
    if model._grads_fresh:
        return torch.zeros_like(g)
    return g


class ClassBalancedRandomSampling:
    """Class-balanced draws from the memory (reference: utils/buffer/buffer_utils.py:74-160).  Two class-level tables, shared by the
    ASER plugins and reset by their constructors exactly as in the reference: `class_index_cache` {label: set of slots} and
    `class_num_cache` (slots per label).  Which sample of a class gets drawn depends on torch's CPU generator (one randperm per
    non-empty class) AND on CPython's iteration order of the freshly built difference set, so both are kept as they are; only the
    selected rows are gathered on the GPU."""
    class_index_cache = None
    class_num_cache = None
    _scratch = None
    # bookkeeping for the C helper's per-class memo of `slots - set()` iteration orders (csrc/hostc.c): a version per label, bumped by
    # update_cache whenever a slot enters or leaves the class; a token that changes whenever the dict itself is another object
    _versions = None
    _tracked = None
    _token = 0
    # The C helper can compare a checksum of every live set with the one it memoised (it catches in-place mutations behind update_cache's
    # back, at the price of a walk over the set's whole hash table: most of a draw once the sets have been churned, and growing with the
    # step count -- profiles/r6_aser_drift_probe.txt).  Default: every draw.  The ASER plugins, whose every mutation goes through
    # update_cache, set verify_every = 64: one verified draw in 64.
    verify_every = 1
    _draws = 0

    @classmethod
    def draw(cls, n_smp_cls, excl_indices=None):
        """The slot indices of one class-balanced draw (host tensor): the statement of the reference's loop.  One randperm per
        non-empty class on torch's CPU generator; which slot a permutation index means is the iteration order of the NEW set
        `slots - excluded`."""
        excluded = excl_indices if excl_indices is not None else set()
        picks = []
        for slots in cls.class_index_cache.values():          # dict insertion order = order in which classes first appeared
            if not slots:
                continue
            eligible = slots - excluded
            shuffle = torch.randperm(len(eligible))           # drawn even when nothing is eligible (len 0), as the reference does
            members = list(eligible)
            picks.extend(members[j] for j in shuffle.tolist()[:n_smp_cls])
        # one tensor conversion for the whole draw (the reference builds and concatenates one tensor per class: ~1 ms per call)
        return torch.tensor(picks, dtype=torch.long)

    @classmethod
    def draw_fast(cls, n_smp_cls, excl_indices=None):
        """Same draw through the C helper (csrc/hostc.c): same CPython set operations, the generator's word stream restated in
        C on the bytes of torch.get_rng_state().  ~0.1 ms instead of ~0.65 ms for 100 classes, and the two draws of an ASER
        retrieval sit on the step's critical path (the GPU waits for them)."""
        if not _hostc_usable():
            return cls.draw(n_smp_cls, excl_indices)
        cache = cls.class_index_cache
        if cache is not cls._tracked:     # another dict (plugin re-initialised, rebuilt from the labels, set by a test): forget the memo
            cls._tracked = cache          # (the reference keeps the tracked dict alive, so its identity cannot be reused)
            ClassBalancedRandomSampling._token += 1
            cls._versions = np.zeros(4096, dtype=np.int64)
        state = torch.get_rng_state()
        room = max(1, len(cache) * max(0, int(n_smp_cls)))
        if cls._scratch is None or cls._scratch.shape[0] < room:
            cls._scratch = np.empty(room, dtype=np.int64)
        entry_state = state.clone()       # (the C call advances `state` in place)
        try:
            ClassBalancedRandomSampling._draws += 1
            verify = 1 if cls.verify_every <= 1 or ClassBalancedRandomSampling._draws % cls.verify_every == 0 else 0
            emulate = 1 if (excl_indices and _EMULATE and _setdiff_emulation_ok()) else 0
            n = _hostc.cbrs_sample(cache, excl_indices, int(n_smp_cls), state.numpy(), cls._scratch, cls._versions, cls._token, verify, emulate)
        except Exception:
            # the C helper gave up midway (a class set holding a non-integer, out of memory): the generator has not been touched yet
            # -- it is set only below -- so the Python loop replays the draw from the state this call started with
            torch.set_rng_state(entry_state)
            cls.invalidate()
            return cls.draw(n_smp_cls, excl_indices)
        torch.set_rng_state(state)
        return torch.from_numpy(cls._scratch[:n].copy())

    @classmethod
    def invalidate(cls):
        """Forget the C helper's memoised iteration orders (call after mutating a class set other than through update_cache; the
        helper also compares a checksum of every set's (element, hash-table slot) pairs -- contents and layout, hence iteration
        order -- so this is belt and braces)."""
        cls._tracked = None

    @classmethod
    def sample(cls, buffer_x, buffer_y, n_smp_cls, excl_indices=None, device="cpu", label_host=None):
        """Up to n_smp_cls slots of every class present, excluding `excl_indices` -> (x, y, slot indices [host])."""
        sample_ind = cls.draw_fast(n_smp_cls, excl_indices)

        x, y = ops.gather_pair(buffer_x, buffer_y, sample_ind)
        if label_host is not None:
            y.host = label_host[sample_ind.numpy()]
        return x, y, sample_ind

    @classmethod
    def update_cache(cls, buffer_y_host, num_class, new_y=None, ind=None, device="cpu"):
        """new_y / ind given (ASER update in use): slots `ind` are about to hold labels `new_y` -- move them between the class sets
        (`buffer_y_host`, the numpy label mirror, still holds the labels being replaced).  Otherwise (ASER retrieval alone):
        rebuild the index from the whole label array; the counts are not touched by this path in the reference either."""
        if cls.class_index_cache is None:
            cls.class_index_cache = defaultdict(set)
            cls.class_num_cache = torch.zeros(num_class, dtype=torch.long)
        if new_y is None:
            rebuilt = defaultdict(set)
            for slot, label in enumerate(buffer_y_host):
                rebuilt[int(label)].add(slot)
            cls.class_index_cache = rebuilt
            return
        versions = cls._versions if cls._tracked is cls.class_index_cache else None
        for slot, label in zip(ind, new_y):
            slot, label = int(slot), int(label)
            previous = int(buffer_y_host[slot])
            if previous in cls.class_index_cache and slot in cls.class_index_cache[previous]:
                cls.class_index_cache[previous].remove(slot)
                cls.class_num_cache[previous] -= 1
                if versions is not None and 0 <= previous < 4096:
                    versions[previous] += 1
            cls.class_index_cache[label].add(slot)
            cls.class_num_cache[label] += 1
            if versions is not None and 0 <= label < 4096:
                versions[label] += 1

class BufferClassTracker(object):
    """buffer_utils.py:163-203: per-buffer class -> set-of-slots index and per-class counts, maintained by the reservoir update
    (utils/buffer/reservoir_update.py:25-26,56-57) and read by match_retrieve.  `buffer_y` is the numpy label mirror."""

    def __init__(self, num_class, device="cpu"):
        self.class_index_cache = defaultdict(set)
        self.class_num_cache = np.zeros(num_class)

    def update_cache(self, buffer_y, new_y=None, ind=None):
        for slot, new_label in zip(ind, new_y):
            slot, new_label, old_label = int(slot), int(new_label), int(buffer_y[slot])
            if old_label in self.class_index_cache and slot in self.class_index_cache[old_label]:
                self.class_index_cache[old_label].remove(slot)
                self.class_num_cache[old_label] -= 1
            self.class_index_cache[new_label].add(slot)
            self.class_num_cache[new_label] += 1

    def check_tracker(self):
        print(self.class_num_cache.sum())
        print(len([k for i in self.class_index_cache.values() for k in i]))
