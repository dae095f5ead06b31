"""TEST INFRASTRUCTURE ONLY — CPU restatement of the replay hot path of RaptorMai/online-continual-learning.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this module, and only as the
checker / timed baseline; the product package (online-continual-learning_amd) never does.

Pinned against the reference itself: oracle/make_golden.py imports /root/reference in the build container, runs
the reference's own functions / agents on seeded inputs and writes tests/golden/*.npz; tests/test_oracle_golden.py
(CPU, `-m "not gpu"`) checks every function below against those vectors.  Integer / index results are compared
exactly, fp32 results to the tolerances stated in the tests.

Third-party arithmetic the reference delegates to (not under /root/reference): PyTorch ATen (reference pin 1.7.1,
container 2.10.0+rocm7.0 — conv2d, batch_norm, linear, cross_entropy, SGD, sort, RNG) and kornia==0.4.1
(augmentation; absent -> PARITY UNPINNED for the augmentation only, see `identity_aug`).

Every function cites the reference file:line it restates.
"""
import itertools
import math
from collections import OrderedDict, defaultdict

import numpy as np
import torch
import torch.nn.functional as F

# ======================================================================================================
# kNN Shapley value  — utils/buffer/aser_utils.py:7-61 (closed form), :94-116 (ranking), utils/utils.py:93-95
# ======================================================================================================


LAST_RETRIEVE_AUX = None


def sq_dist_matrix(eval_f, cand_f):
    """utils/utils.py:93-95 applied pairwise (aser_utils.py:108-114): sum((u-v)^2) over features, fp32."""
    e = np.asarray(eval_f, dtype=np.float32)[:, None, :]
    c = np.asarray(cand_f, dtype=np.float32)[None, :, :]
    d = (e - c)
    return (d * d).sum(-1, dtype=np.float32)


def knn_sv(eval_f, eval_y, cand_f, cand_y, k, order=None):
    """aser_utils.py:29-59.  Returns (sv [n_eval,n_cand] float32, order [n_eval,n_cand]).
    Arithmetic mirrors the reference on CPU: factor = numer/denom in fp32 (:43-49), product in fp32 (:51), reverse
    cumulative sum accumulated in fp64 and rounded to fp32 per element (torch CPU cumsum), scatter (:55-59).
    Ties in distance are broken by ascending candidate index (the reference's argsort is unstable: any order of
    exactly-tied candidates is a valid reference output)."""
    eval_y = np.asarray(eval_y)
    cand_y = np.asarray(cand_y)
    n_eval, n_cand = len(eval_y), len(cand_y)
    if order is None:
        order = np.argsort(sq_dist_matrix(eval_f, cand_f), axis=1, kind="stable")
    sv = np.zeros((n_eval, n_cand), dtype=np.float32)
    pos = np.arange(n_cand, dtype=np.float32) + 1.0
    denom = pos.copy()
    denom[:n_cand - 1] *= np.float32(k)
    numer = pos.copy()
    numer[k:n_cand - 1] = np.float32(k)
    numer[n_cand - 1] = 1.0
    factor = (numer / denom).astype(np.float32)
    for i in range(n_eval):
        ind = (cand_y[order[i]] == eval_y[i]).astype(np.float32)
        nxt = np.zeros_like(ind)
        nxt[:-1] = ind[1:]
        term = ((ind - nxt) * factor).astype(np.float32)
        acc = 0.0
        out = np.zeros(n_cand, dtype=np.float32)
        for j in range(n_cand - 1, -1, -1):
            acc += float(term[j])
            out[j] = np.float32(acc)
        sv[i, order[i]] = out
    return sv, order


def knn_shapley_bruteforce(dist_row, match_row, k):
    """Independent known-answer: exact Shapley value by enumerating all permutations for ONE evaluation point with
    utility v(S) = (1/k) * sum over the min(k,|S|) nearest members of S of 1[label matches]."""
    n = len(dist_row)
    phi = np.zeros(n, dtype=np.float64)

    def util(members):
        if not members:
            return 0.0
        near = sorted(members, key=lambda j: (dist_row[j], j))[:k]
        return sum(match_row[j] for j in near) / float(k)

    for perm in itertools.permutations(range(n)):
        s = []
        prev = 0.0
        for j in perm:
            s.append(j)
            cur = util(s)
            phi[j] += cur - prev
            prev = cur
    return phi / math.factorial(n)


def argsort_desc_stable(v):
    """Descending argsort with ties in ascending index order — the deterministic member of the reference's tie
    equivalence class (torch's CPU argsort(descending=True) is unstable: SURVEY §4)."""
    return np.argsort(-np.asarray(v), kind="stable")


def argsort_desc_torch(v):
    """torch CPU argsort(descending=True): what the reference executes (aser_retrieve.py:88, aser_update.py:88,
    mir_retrieve.py:29); only used by oracle/make_golden.py to separate logic errors from tie-order effects."""
    return torch.from_numpy(np.ascontiguousarray(v)).argsort(descending=True).numpy()


ARGSORT_DESC = argsort_desc_stable


def aser_score(sv_adv, sv_coop, aser_type):
    """aser_retrieve.py:77-86."""
    if aser_type == "neg_sv":
        return (sv_adv.sum(0) * -1).astype(np.float32)
    if aser_type == "asv":
        return (sv_coop.max(0) - sv_adv.min(0)).astype(np.float32)
    return (sv_coop.mean(0) - sv_adv.mean(0)).astype(np.float32)


# ======================================================================================================
# losses — utils/loss.py:19-96, agents/base.py:95,113, utils/buffer/mir_retrieve.py:26-28
# ======================================================================================================


def supcon_loss(features_bvd, labels, temperature):
    """utils/loss.py:42-96 with contrast_mode='all'; torch CPU fp32 (autograd gives the reference gradient)."""
    b, v = features_bvd.shape[0], features_bvd.shape[1]
    z = torch.cat([features_bvd[:, i] for i in range(v)], dim=0)            # :56
    sim = (z @ z.t()) / temperature                                          # :67-69
    sim = sim - sim.max(dim=1, keepdim=True)[0].detach()                     # :71-72
    a = b * v
    notself = 1.0 - torch.eye(a, dtype=z.dtype, device=z.device)             # :77-82
    lab = labels.view(-1, 1)
    pos = (lab == lab.t()).to(z.dtype).repeat(v, v) * notself                # :51,75,83
    logprob = sim - torch.log((torch.exp(sim) * notself).sum(1, keepdim=True))  # :86-87
    per_anchor = -(pos * logprob).sum(1) / pos.sum(1)                        # :90,93
    return per_anchor.view(v, b).mean()                                      # :94


def ce_mean(logits, y):
    return F.cross_entropy(logits, y, reduction="mean")


def ce_labels_trick(logits, labels):
    """agents/base.py:96-101: cross-entropy over the heads that appear in the batch only."""
    labels = labels.clone()
    unq = labels.unique().sort()[0]
    for i, lbl in enumerate(unq):
        labels[labels == lbl] = i
    return F.cross_entropy(logits[:, unq], labels, reduction="mean")


def ce_separated_softmax(logits, labels, old_labels, new_labels, lbl_inv_map):
    """agents/base.py:102-108: old and new classes normalised separately, NLL over the concatenation."""
    old_ss = F.log_softmax(logits[:, old_labels], dim=1)
    new_ss = F.log_softmax(logits[:, new_labels], dim=1)
    ss = torch.cat([old_ss, new_ss], dim=1)
    idx = torch.tensor([lbl_inv_map[int(l)] for l in labels.tolist()], dtype=torch.long)
    return F.nll_loss(ss, idx)


def loss_fn_kd(scores, target_scores, T=2.0):
    """utils/kd_manager.py:6-11."""
    log_scores_norm = F.log_softmax(scores / T, dim=1)
    targets_norm = F.softmax(target_scores / T, dim=1)
    return (-1 * targets_norm * log_scores_norm).sum(dim=1).mean() * T ** 2


def mir_scores(logits_pre, logits_post, y):
    """mir_retrieve.py:26-28."""
    return F.cross_entropy(logits_post, y, reduction="none") - F.cross_entropy(logits_pre, y, reduction="none")


# ======================================================================================================
# Reduced-ResNet18 / SupConResNet — models/resnet.py:14-37,69-116,140-168 as a functional over a state dict
# ======================================================================================================


class _MaskedRelu(torch.autograd.Function):
    """ReLU whose BACKWARD uses a given 0/1 mask instead of (output > 0).  Used by the parity tests to teacher-force the
    activation pattern of the implementation under test: a pre-activation within fp32 round-off of zero may legitimately
    land on either side, and the gradient is discontinuous there."""

    @staticmethod
    def forward(ctx, x, mask):
        ctx.save_for_backward(mask)
        return x.clamp(min=0)

    @staticmethod
    def backward(ctx, g):
        (mask,) = ctx.saved_tensors
        return g * mask, None


class OracleNet(object):
    """Functional restatement: `state` maps the reference's state_dict keys to CPU tensors (parameters may
    require grad).  head: None -> logits = linear(features) (ResNet.forward); 'mlp' | 'linear' | 'None' ->
    SupConResNet.forward (features -> head -> F.normalize)."""

    def __init__(self, state, head=None, training=True):
        self.s = state
        self.head = head
        self.training = training
        self.pre = "encoder." if head is not None else ""
        self.rec = None   # set to a dict to record the raw conv output feeding every BatchNorm (layer-wise parity tests)
        self.tape = None  # set to a dict to keep LIVE intermediates with retain_grad (stage-wise backward diagnostics)
        self.mask_override = None  # dict key -> float 0/1 NCHW mask used for the ReLU backward (keys "z:stem", "a1:<block>", "z:<block>")
        self.pre_act = None   # set to a dict to record ReLU inputs (to judge how ambiguous a mask mismatch is)

    def _relu(self, x, key):
        if self.pre_act is not None:
            self.pre_act[key] = x.detach()
        if self.mask_override is not None and key in self.mask_override:
            return _MaskedRelu.apply(x, self.mask_override[key])
        return F.relu(x)

    def _bn(self, x, name):
        s = self.s
        if self.rec is not None:
            self.rec[name] = x.detach()
        if self.tape is not None and x.requires_grad:
            x.retain_grad()
            self.tape["y:" + name] = x
        if self.training:
            s[name + ".num_batches_tracked"] += 1
        return F.batch_norm(x, s[name + ".running_mean"], s[name + ".running_var"], s[name + ".weight"], s[name + ".bias"],
                            self.training, 0.1, 1e-5)

    def _block(self, x, p, stride):
        s = self.s
        out = self._relu(self._bn(F.conv2d(x, s[p + ".conv1.weight"], None, stride, 1), p + ".bn1"), "a1:" + p)
        if self.tape is not None and out.requires_grad:
            out.retain_grad()
            self.tape["a1:" + p] = out
        out = self._bn(F.conv2d(out, s[p + ".conv2.weight"], None, 1, 1), p + ".bn2")
        if (p + ".shortcut.0.weight") in s:
            sc = self._bn(F.conv2d(x, s[p + ".shortcut.0.weight"], None, stride, 0), p + ".shortcut.1")
        else:
            sc = x
        res = self._relu(out + sc, "z:" + p)
        if self.tape is not None and res.requires_grad:
            res.retain_grad()
            self.tape["z:" + p] = res
        return res

    def features(self, x):
        s, pre = self.s, self.pre
        out = self._relu(self._bn(F.conv2d(x, s[pre + "conv1.weight"], None, 1, 1), pre + "bn1"), "z:stem")
        if self.tape is not None and out.requires_grad:
            out.retain_grad()
            self.tape["z:stem"] = out
        for layer in range(1, 5):
            for b in range(2):
                out = self._block(out, "%slayer%d.%d" % (pre, layer, b), 2 if (b == 0 and layer > 1) else 1)
        out = F.avg_pool2d(out, 4)
        return out.view(out.size(0), -1)

    def forward(self, x):
        s = self.s
        f = self.features(x)
        if self.head is None:
            return F.linear(f, s["linear.weight"], s["linear.bias"])
        if self.head == "mlp":
            f = F.linear(F.relu(F.linear(f, s["head.0.weight"], s["head.0.bias"])), s["head.2.weight"], s["head.2.bias"])
        elif self.head == "linear":
            f = F.linear(f, s["head.weight"], s["head.bias"])
        return F.normalize(f, dim=1)

    def param_names(self):
        return [k for k in self.s if not (k.endswith("running_mean") or k.endswith("running_var") or k.endswith("num_batches_tracked"))]


def clone_state(state_dict, requires_grad=True):
    """Detached CPU copy of a module's state_dict, parameters as autograd leaves."""
    out = OrderedDict()
    for k, v in state_dict.items():
        t = v.detach().to("cpu").clone()
        if requires_grad and t.is_floating_point() and not (k.endswith("running_mean") or k.endswith("running_var")):
            t.requires_grad_(True)
        out[k] = t
    return out


def flat_grad(state, names):
    """utils/buffer/buffer_utils.py:58-71: gradients in parameter order, zeros where None."""
    return torch.cat([(state[n].grad if state[n].grad is not None else torch.zeros_like(state[n])).reshape(-1) for n in names])


def sgd_step(state, names, lr, wd=0.0):
    """torch.optim.SGD (momentum 0), utils/setup_elements.py:73-75; parameters without grad are skipped."""
    with torch.no_grad():
        for n in names:
            p = state[n]
            if p.grad is None:
                continue
            g = p.grad
            if wd != 0:
                g = g + wd * p
            p.add_(g, alpha=-lr)


def zero_grad(state, names):
    for n in names:
        state[n].grad = None


# ======================================================================================================
# replay buffer + plugins — utils/buffer/*.py (host logic, numpy / torch CPU RNG)
# ======================================================================================================


class OracleBuffer(object):
    """utils/buffer/buffer.py:9-34: img [mem,C,H,W] f32, label [mem] i64, current_index, n_seen_so_far."""

    def __init__(self, mem_size, shape):
        self.img = torch.zeros((mem_size,) + tuple(shape), dtype=torch.float32)
        self.label = torch.zeros(mem_size, dtype=torch.int64)
        self.current_index = 0
        self.n_seen_so_far = 0
        self.mem_size = mem_size


def random_retrieve_indices(buf, num, excl=None):
    """utils/buffer/buffer_utils.py:9-17 (numpy global RNG)."""
    filled = np.arange(buf.current_index)
    valid = np.setdiff1d(filled, np.array(list(excl) if excl is not None else []))
    num = min(num, valid.shape[0])
    return np.random.choice(valid, num, replace=False).astype(np.int64)


def reservoir_update(buf, x, y, tracker=None):
    """utils/buffer/reservoir_update.py:8-61 (torch CPU RNG).  Returns the list of written slots.  tracker: the
    BufferClassTracker hooks of :25-26 (after the fill, only when the whole batch fitted) and :56-57 (before the overwrite)."""
    n = x.shape[0]
    room = max(0, buf.mem_size - buf.current_index)
    if room:
        take = min(room, n)
        buf.img[buf.current_index:buf.current_index + take] = x[:take]
        buf.label[buf.current_index:buf.current_index + take] = y[:take]
        buf.current_index += take
        buf.n_seen_so_far += take
        if take == n:
            filled = list(range(buf.current_index - take, buf.current_index))
            if tracker is not None:
                tracker.update(buf.label, y[:take], filled)
            return filled
    x, y = x[room:], y[room:]
    draw = torch.FloatTensor(x.shape[0]).uniform_(0, buf.n_seen_so_far).long()
    keep = (draw < buf.mem_size).nonzero().squeeze(-1)
    slots = draw[keep]
    buf.n_seen_so_far += x.shape[0]
    if slots.numel() == 0:
        return []
    last_writer = {}
    for s, src in zip(slots.tolist(), keep.tolist()):
        last_writer[s] = src
    ks, vs = list(last_writer.keys()), list(last_writer.values())
    if tracker is not None:
        tracker.update(buf.label, y[vs], ks)
    buf.img[ks] = x[vs]
    buf.label[ks] = y[vs]
    return ks


class ClassTracker(object):
    """utils/buffer/buffer_utils.py:163-203 (BufferClassTracker): class -> set of slots, class counts (numpy float)."""

    def __init__(self, num_class):
        self.index = defaultdict(set)
        self.count = np.zeros(num_class)

    def update(self, labels, new_y, ind):
        orig = labels[ind]
        for i, ny, oy in zip(ind, new_y, orig):
            oy, ny = oy.item(), ny.item()
            if oy in self.index and i in self.index[oy]:
                self.index[oy].remove(i)
                self.count[oy] -= 1
            self.index[ny].add(i)
            self.count[ny] += 1


def match_retrieve_indices(tracker, cur_y, exclude=None):
    """utils/buffer/buffer_utils.py:29-49 (match_retrieve): one buffered sample of the same class per item of the batch, drawn
    with Python's `random.sample` from the class's slot set (CPython set order); empty when a class is short."""
    import random
    from collections import Counter
    ys = cur_y.tolist()
    counter = Counter(ys)
    where = defaultdict(list)
    for pos, val in enumerate(ys):
        where[val].append(pos)
    select = [None] * len(ys)
    for c in counter:
        members = tracker.index[c]
        if exclude is not None:
            members = members - set(exclude.tolist())
        if not members or len(members) < counter[c]:
            return np.zeros(0, dtype=np.int64)
        got = random.sample(list(members), counter[c])
        for pos, val in zip(where[c], got):
            select[pos] = val
    return np.asarray(select, dtype=np.int64)


def mem_match_indices(buf, tracker, num_retrieve, warmup):
    """utils/buffer/mem_match.py:11-21: random candidates, then one label-matched partner per candidate from the slots outside the
    draw; the candidates are redrawn until every one of them finds a partner.  -> (candidate slots, partner slots), both empty
    before the warm-up or when the memory is empty.  (Like the reference, this never returns when the memory cannot hold a
    partner for every candidate.)"""
    empty = np.zeros(0, dtype=np.int64)
    if not buf.n_seen_so_far > num_retrieve * warmup:
        return empty, empty
    while True:
        cand = random_retrieve_indices(buf, num_retrieve)
        if cand.shape[0] == 0:
            return cand, empty
        partners = match_retrieve_indices(tracker, buf.label[cand], exclude=cand)
        if partners.shape[0] > 0:
            return cand, partners


class ClassCache(object):
    """utils/buffer/buffer_utils.py:74-160 (ClassBalancedRandomSampling): class -> set of slots, class counts."""

    def __init__(self):
        self.index = None
        self.count = None

    def update(self, labels, num_class, new_y=None, ind=None):
        if self.index is None:
            self.index = defaultdict(set)
            self.count = torch.zeros(num_class, dtype=torch.long)
        if new_y is not None:
            for i, ny in zip([int(v) for v in ind], [int(v) for v in new_y]):
                oy = int(labels[i])
                if oy in self.index and i in self.index[oy]:
                    self.index[oy].remove(i)
                    self.count[oy] -= 1
                self.index[ny].add(i)
                self.count[ny] += 1
        else:
            fresh = defaultdict(set)
            for i, c in enumerate(labels.tolist()):
                fresh[c].add(i)
            self.index = fresh

    def sample(self, n_per_class, excl=None):
        """:81-121 — one torch.randperm per non-empty class (CPU generator); CPython set order decides ties."""
        excl = excl or set()
        picked = torch.tensor([], dtype=torch.long)
        for members in self.index.values():
            if members:
                valid = members - excl
                perm = torch.randperm(len(valid))
                picked = torch.cat((picked, torch.tensor(list(valid), dtype=torch.long)[perm][:n_per_class]))
        return picked


def minority_indices(cache, cur_y, mem_size, num_class):
    """utils/buffer/aser_utils.py:147-153."""
    thr = torch.tensor(1).float().uniform_(0, 1 / num_class).item()
    prop = cache.count.float() / mem_size
    return (prop[cur_y] < thr).nonzero(as_tuple=True)[0]


def deep_features(net, x):
    """utils/utils.py:45-90: eval-mode, no-grad features in chunks of 64."""
    was = net.training
    net.training = False
    with torch.no_grad():
        out = torch.cat([net.features(x[i:i + 64]) for i in range(0, x.shape[0], 64)], 0) if x.shape[0] else torch.zeros(0)
    net.training = was
    return out.reshape(x.shape[0], -1)


def aser_retrieve(net, buf, cache, cur_x, cur_y, params, is_aser_upt=True):
    """utils/buffer/aser_retrieve.py:21-92. Returns (ret_idx_into_buffer, cand_ind, sv)."""
    n_smp = int(params["n_smp_cls"])
    if buf.n_seen_so_far <= params["mem_size"]:
        idx = random_retrieve_indices(buf, params["eps_mem_batch"])
        return torch.from_numpy(idx), None, None
    if not is_aser_upt:
        cache.update(buf.label, params["n_classes"])
    cand = cache.sample(n_smp)
    f = deep_features(net, torch.cat((cur_x, buf.img[cand])))
    sv_adv, _ = knn_sv(f[:cur_x.shape[0]].numpy(), cur_y.numpy(), f[cur_x.shape[0]:].numpy(), buf.label[cand].numpy(), params["k"])
    if params["aser_type"] != "neg_sv":
        coop = cache.sample(n_smp, excl=set(cand.tolist()))
        f2 = deep_features(net, torch.cat((buf.img[coop], buf.img[cand])))
        sv_coop, _ = knn_sv(f2[:coop.shape[0]].numpy(), buf.label[coop].numpy(), f2[coop.shape[0]:].numpy(), buf.label[cand].numpy(),
                            params["k"])
        sv = aser_score(sv_adv, sv_coop, params["aser_type"])
    else:
        sv = aser_score(sv_adv, None, "neg_sv")
    order = ARGSORT_DESC(sv)
    global LAST_RETRIEVE_AUX   # features behind the scores, for the near-tie-aware parity checks in tests/
    LAST_RETRIEVE_AUX = dict(adv=(f[:cur_x.shape[0]].numpy(), cur_y.numpy(), f[cur_x.shape[0]:].numpy(), buf.label[cand].numpy()))
    if params["aser_type"] != "neg_sv":
        LAST_RETRIEVE_AUX["coop"] = (f2[:coop.shape[0]].numpy(), buf.label[coop].numpy(), f2[coop.shape[0]:].numpy(), buf.label[cand].numpy())
    return cand[order[:params["eps_mem_batch"]]], cand, sv


def aser_update(net, buf, cache, x, y, params):
    """utils/buffer/aser_update.py:22-112. Returns (ind_buffer, ind_cur) of the replacement (or None while filling)."""
    room = params["mem_size"] - buf.current_index
    if room:
        xf, yf = x[:room], y[:room]
        cache.update(buf.label, params["n_classes"], new_y=yf, ind=range(buf.current_index, buf.current_index + xf.shape[0]))
        reservoir_update(buf, xf, yf)
    if buf.current_index != params["mem_size"]:
        return None
    cur_x, cur_y = x[room:], y[room:]
    minority = minority_indices(cache, cur_y, params["mem_size"], params["n_classes"])
    ev = cache.sample(int(params["n_smp_cls"]))
    eval_x = torch.cat((buf.img[ev], cur_x[minority]))
    eval_y = torch.cat((buf.label[ev], cur_y[minority]))
    cand_ind = torch.from_numpy(random_retrieve_indices(buf, int(params["n_smp_cls"] * params["n_classes"]), set(ev.tolist())))
    cand_x = torch.cat((buf.img[cand_ind], cur_x))
    cand_y = torch.cat((buf.label[cand_ind], cur_y))
    f = deep_features(net, torch.cat((eval_x, cand_x)))
    sv, _ = knn_sv(f[:eval_x.shape[0]].numpy(), eval_y.numpy(), f[eval_x.shape[0]:].numpy(), cand_y.numpy(), params["k"])
    tot = sv.sum(0)
    n_cur, n_cand = cur_x.shape[0], cand_x.shape[0]
    n_buf = n_cand - n_cur
    order = torch.from_numpy(np.ascontiguousarray(ARGSORT_DESC(tot)))
    large, small = order[:n_buf], order[n_buf:]
    ind_cur = large[(large >= n_buf).nonzero(as_tuple=True)[0]] - n_buf
    ind_buffer = cand_ind[small[(small < n_buf).nonzero(as_tuple=True)[0]]]
    buf.n_seen_so_far += n_cur
    cache.update(buf.label, params["n_classes"], new_y=cur_y[ind_cur], ind=ind_buffer)
    buf.img[ind_buffer] = cur_x[ind_cur]
    buf.label[ind_buffer] = cur_y[ind_cur]
    return dict(ind_buffer=ind_buffer.numpy(), ind_cur=ind_cur.numpy(), sv=tot, eval_indices=ev.numpy(), cand_ind=cand_ind.numpy(),
                order=order.numpy(), n_minority=int(minority.numel()),
                aux=(f[:eval_x.shape[0]].numpy(), eval_y.numpy(), f[eval_x.shape[0]:].numpy(), cand_y.numpy()))


# ======================================================================================================
# NCM classifier — agents/base.py:121-142,159-176
# ======================================================================================================


def ncm_means(feats, labels, class_ids):
    """Per-class mean of L2-normalised features, re-normalised. Classes without exemplars -> NaN row (caller fills)."""
    f = feats / feats.norm(dim=1, keepdim=True)
    out = torch.full((len(class_ids), feats.shape[1]), float("nan"))
    for i, c in enumerate(class_ids):
        m = labels == c
        if m.any():
            mu = f[m].mean(0)
            out[i] = mu / mu.norm()
    return out


def ncm_predict(feats, means):
    f = feats / feats.norm(dim=1, keepdim=True)
    d = ((f[:, None, :] - means[None, :, :]) ** 2).sum(-1)
    return d.min(1)[1]


# ======================================================================================================
# whole steps (teacher-forced), used for parity tests and as the timed CPU baseline
# ======================================================================================================


def identity_aug(x):
    return x


def _retrieve_indices(buf, params, retrieve, batch_y, tracker, warmup):
    """random (utils/buffer/random_retrieve.py:3-9) or match (utils/buffer/sc_retrieve.py:4-15) retrieval -> buffer slots."""
    if retrieve == "match":
        if buf.n_seen_so_far > params["eps_mem_batch"] * warmup:
            return match_retrieve_indices(tracker, batch_y)
        return np.zeros(0, dtype=np.int64)
    return random_retrieve_indices(buf, params["eps_mem_batch"])


def scr_step(state, names, buf, batch_x, batch_y, params, aug=identity_aug, retrieve="random", tracker=None, warmup=4):
    """agents/scr.py:40-63 for ONE iteration. Returns (loss or None, retrieved indices, written slots)."""
    net = OracleNet(state, head=params.get("head", "mlp"), training=True)
    idx = _retrieve_indices(buf, params, retrieve, batch_y, tracker, warmup)
    loss = None
    if idx.shape[0] > 0:
        mem_x, mem_y = buf.img[idx], buf.label[idx]
        cx = torch.cat((mem_x, batch_x))
        cy = torch.cat((mem_y, batch_y))
        feats = torch.cat([net.forward(cx).unsqueeze(1), net.forward(aug(cx)).unsqueeze(1)], dim=1)   # two forwards (:55)
        loss = supcon_loss(feats, cy, params["temp"])
        zero_grad(state, names)
        loss.backward()
        sgd_step(state, names, params["lr"])
    slots = reservoir_update(buf, batch_x, batch_y, tracker=tracker)
    return (None if loss is None else float(loss.detach())), idx, slots


def er_step(state, names, buf, batch_x, batch_y, params, retrieve="random", gss=None, tracker=None, warmup=4, kd=None):
    """agents/exp_replay.py:34-92 for ONE iteration with random / MIR / match retrieval and reservoir or GSS update.
    kd(loss, logits, x) -> loss: the KD tricks' blend of the cross-entropy with the distillation loss (exp_replay.py:42-47, :64-69)."""
    net = OracleNet(state, head=None, training=True)
    logits = net.forward(batch_x)
    loss = ce_mean(logits, batch_y)
    if kd is not None:
        loss = kd(loss, logits, batch_x)
    zero_grad(state, names)
    loss.backward()
    info = {"loss": float(loss.detach())}
    if retrieve == "MIR":
        sub = random_retrieve_indices(buf, params["subsample"])
        g = flat_grad(state, names)
        info["sub"] = sub
        info["grad"] = g.clone()
        if sub.shape[0] > 0:
            virt = OrderedDict()
            o = 0
            for k, v in state.items():
                if k in names:
                    n_el = v.numel()
                    virt[k] = (v.detach() - params["lr"] * g[o:o + n_el].view_as(v))
                    o += n_el
                else:
                    virt[k] = v.detach().clone()      # the deepcopy's BN buffers
            sub_x, sub_y = buf.img[sub], buf.label[sub]
            with torch.no_grad():
                pre = net.forward(sub_x)
                post = OracleNet(virt, head=None, training=True).forward(sub_x)
                sc = mir_scores(pre, post, sub_y)
            top = ARGSORT_DESC(sc.numpy())[:params["eps_mem_batch"]]
            idx = sub[top]
            info["scores"] = sc.numpy()
        else:
            idx = sub
    else:
        idx = _retrieve_indices(buf, params, retrieve, batch_y, tracker, warmup)
    info["idx"] = idx
    if idx.shape[0] > 0:
        mem_logits = net.forward(buf.img[idx])
        loss_mem = ce_mean(mem_logits, buf.label[idx])
        if kd is not None:
            loss_mem = kd(loss_mem, mem_logits, buf.img[idx])
        loss_mem.backward()
        info["loss_mem"] = float(loss_mem.detach())
    sgd_step(state, names, params["lr"])
    if gss is not None:
        info["gss"] = gss_update(gss, state, names, buf, batch_x, batch_y)
    else:
        info["slots"] = reservoir_update(buf, batch_x, batch_y, tracker=tracker)
    return info


def aser_er_step(state, names, buf, cache, batch_x, batch_y, params):
    """agents/exp_replay.py:34-92 with --retrieve ASER --update ASER (combined-batch branch :79-87)."""
    net = OracleNet(state, head=None, training=True)
    info = {}
    loss = ce_mean(net.forward(batch_x), batch_y)
    zero_grad(state, names)
    loss.backward()
    ret_idx, cand, sv = aser_retrieve(net, buf, cache, batch_x, batch_y, params)
    info["ret_idx"] = ret_idx.numpy() if torch.is_tensor(ret_idx) else ret_idx
    info["cand"] = None if cand is None else cand.numpy()
    info["sv"] = sv
    info["ret_aux"] = None if cand is None else LAST_RETRIEVE_AUX
    mem_x, mem_y = buf.img[ret_idx], buf.label[ret_idx]
    if mem_x.shape[0] > 0:
        ce_mean(net.forward(mem_x), mem_y).backward()
    zero_grad(state, names)
    cx, cy = torch.cat((mem_x, batch_x)), torch.cat((mem_y, batch_y))
    lc = ce_mean(net.forward(cx), cy)
    lc.backward()
    sgd_step(state, names, params["lr"])
    info["loss"] = float(lc.detach())
    info["upd"] = aser_update(net, buf, cache, batch_x, batch_y, params)
    return info


def mir_scores_for_gradient(state, names, grad_vector, sub_x, sub_y, lr):
    """utils/buffer/mir_retrieve.py:19-28 for a GIVEN gradient vector: per-sample CE at theta - lr*g minus per-sample CE at theta,
    both forwards in train mode under no_grad on copies of `state` (the deepcopy's BatchNorm buffers, :35)."""
    pre_state = OrderedDict((k, v.detach().clone()) for k, v in state.items())
    virt = OrderedDict()
    o = 0
    for k, v in state.items():
        if k in names:
            n_el = v.numel()
            virt[k] = v.detach() - lr * grad_vector[o:o + n_el].view_as(v)
            o += n_el
        else:
            virt[k] = v.detach().clone()
    with torch.no_grad():
        pre = OracleNet(pre_state, head=None, training=True).forward(sub_x)
        post = OracleNet(virt, head=None, training=True).forward(sub_x)
        return mir_scores(pre, post, sub_y).numpy()


def review_epoch(state, names, buf, params, agent, aug=identity_aug):
    """agents/base.py:62-88 (review trick, run by after_train): one epoch over the filled part of the buffer in shuffled
    mini-batches of eps_mem_batch (drop_last), each step with the gradients divided by 10.  For SCR the reference runs one plain
    forward first (:77; only its BatchNorm running-statistic update survives) and then the two view forwards (:78-80).
    Returns the per-batch (indices, loss)."""
    n = buf.current_index
    out = []
    if n == 0:
        return out
    head = params.get("head") if agent == "SCR" else None
    net = OracleNet(state, head=head, training=True)
    loader = torch.utils.data.DataLoader(_Idx(n), batch_size=params["eps_mem_batch"], shuffle=True, num_workers=0, drop_last=True)
    for idx in loader:
        bx, by = buf.img[idx], buf.label[idx]
        logits = net.forward(bx)
        if agent == "SCR":
            feats = torch.cat([net.forward(bx).unsqueeze(1), net.forward(aug(bx)).unsqueeze(1)], dim=1)
            loss = supcon_loss(feats, by, params["temp"])
        else:
            loss = ce_mean(logits, by)
        zero_grad(state, names)
        loss.backward()
        with torch.no_grad():
            for k in names:
                if state[k].grad is not None:
                    state[k].grad.copy_(state[k].grad.clone() / 10.)
        sgd_step(state, names, params["lr"])
        out.append((idx.numpy().copy(), float(loss.detach())))
    return out


# ======================================================================================================
# GSS-Greedy update — utils/buffer/gss_greedy_update.py:6-122
# ======================================================================================================


class GssState(object):
    """:7-13: gss_mem_strength gradient vectors of gss_batch_size samples each, one score per slot."""

    def __init__(self, cfg):
        self.mem_strength = cfg.get("gss_mem_strength", 10)
        self.batch_size = cfg.get("gss_batch_size", 10)
        self.score = torch.zeros(cfg["mem_size"])


def cosine_similarity(x1, x2, eps=1e-8):
    """utils/buffer/buffer_utils.py:51-56."""
    w1 = x1.norm(p=2, dim=1, keepdim=True)
    w2 = x2.norm(p=2, dim=1, keepdim=True)
    return torch.mm(x1, x2.t()) / (w1 * w2.t()).clamp(min=eps)


def eval_mode_grad(state, names, x, y):
    """Flat gradient of the mean CE of an EVAL-mode forward (the plugin calls model.eval() first, :16): BatchNorm is the affine
    map of its running statistics.  get_grad_vector layout (buffer_utils.py:58-71)."""
    net = OracleNet(state, head=None, training=False)
    zero_grad(state, names)
    F.cross_entropy(net.forward(x), y).backward()
    return flat_grad(state, names).detach().clone()


def gss_rand_mem_grads(gss, state, names, buf):
    """:82-104."""
    bs = min(gss.batch_size, buf.current_index)
    n_sub = min(gss.mem_strength, buf.current_index // bs)
    perm = torch.randperm(buf.current_index)
    rows = []
    for i in range(n_sub):
        idx = perm[i * bs:i * bs + bs]
        rows.append(eval_mode_grad(state, names, buf.img[idx], buf.label[idx]))
    return torch.stack(rows)


def gss_each_sample_sim(state, names, mem_grads, x, y):
    """:106-122."""
    out = torch.zeros(x.shape[0])
    for i in range(x.shape[0]):
        g = eval_mode_grad(state, names, x[i:i + 1], y[i:i + 1]).unsqueeze(0)
        out[i] = max(cosine_similarity(mem_grads, g))
    return out


def gss_update(gss, state, names, buf, x, y):
    """:15-64.  Returns a record of what happened (for the parity tests)."""
    info = {}
    room = buf.mem_size - buf.current_index
    if room <= 0:
        mem_grads = gss_rand_mem_grads(gss, state, names, buf)
        batch_grad = eval_mode_grad(state, names, x, y).unsqueeze(0)
        batch_sim = max(cosine_similarity(mem_grads, batch_grad))
        info["batch_sim"] = float(batch_sim)
        if batch_sim < 0:
            score = gss.score[:buf.current_index]
            sim = (score - torch.min(score)) / ((torch.max(score) - torch.min(score)) + 0.01)
            index = torch.multinomial(sim, x.shape[0], replacement=False)
            item_sim = gss_each_sample_sim(state, names, mem_grads, x, y)
            scaled = ((item_sim + 1) / 2).unsqueeze(1)
            repl = ((gss.score[index] + 1) / 2).unsqueeze(1)
            outcome = torch.multinomial(torch.cat((scaled, repl), dim=1), 1, replacement=False)
            sub = outcome.squeeze(1).bool()
            buf.img[index[sub]] = x[sub].clone()
            buf.label[index[sub]] = y[sub].clone()
            gss.score[index[sub]] = item_sim[sub].clone()
            info.update(index=index.numpy().copy(), item_sim=item_sim.numpy().copy(), sub=sub.numpy().copy())
    else:
        take = min(room, x.shape[0])
        x, y = x[:take], y[:take]
        if buf.current_index == 0:
            cos = torch.zeros(x.shape[0]) + 0.1
        else:
            mem_grads = gss_rand_mem_grads(gss, state, names, buf)
            cos = gss_each_sample_sim(state, names, mem_grads, x, y)
        buf.img[buf.current_index:buf.current_index + take] = x
        buf.label[buf.current_index:buf.current_index + take] = y
        gss.score[buf.current_index:buf.current_index + take] = cos
        buf.current_index += take
        info.update(fill=take, item_sim=cos.numpy().copy())
    return info


# ======================================================================================================
# seeded initialisation + whole-task driver (experiment/run.py:38-51 for ONE run), used by the CPU parity tests
# ======================================================================================================


def init_state(agent, data, head="mlp"):
    """utils/setup_elements.py:46-68: builds the torch.nn layers in the reference's construction order (same RNG
    draws => same initial weights for a given torch seed) and returns their tensors under the reference's
    state_dict keys."""
    import torch.nn as nn
    n_cls = {"cifar100": 100, "cifar10": 10, "mini_imagenet": 100}[data]
    st = OrderedDict()

    def add_bn(prefix, c):
        bn = nn.BatchNorm2d(c)
        for k, v in bn.state_dict().items():
            st[prefix + "." + k] = v

    def encoder(pre, ncls):
        st[pre + "conv1.weight"] = nn.Conv2d(3, 20, 3, 1, 1, bias=False).weight.data
        add_bn(pre + "bn1", 20)
        inp = 20
        for layer in range(1, 5):
            planes = 20 * 2 ** (layer - 1)
            for b in range(2):
                stride = 2 if (b == 0 and layer > 1) else 1
                p = "%slayer%d.%d" % (pre, layer, b)
                st[p + ".conv1.weight"] = nn.Conv2d(inp, planes, 3, stride, 1, bias=False).weight.data
                add_bn(p + ".bn1", planes)
                st[p + ".conv2.weight"] = nn.Conv2d(planes, planes, 3, 1, 1, bias=False).weight.data
                add_bn(p + ".bn2", planes)
                if stride != 1 or inp != planes:
                    st[p + ".shortcut.0.weight"] = nn.Conv2d(inp, planes, 1, stride, bias=False).weight.data
                    add_bn(p + ".shortcut.1", planes)
                inp = planes
        lin = nn.Linear(160, ncls)
        st[pre + "linear.weight"], st[pre + "linear.bias"] = lin.weight.data, lin.bias.data

    if agent in ("SCR", "SCP"):
        encoder("encoder.", 100)
        dim_in = 640 if data == "mini_imagenet" else 160
        if head == "mlp":
            l0, l2 = nn.Linear(dim_in, dim_in), nn.Linear(dim_in, 128)
            st["head.0.weight"], st["head.0.bias"], st["head.2.weight"], st["head.2.bias"] = l0.weight.data, l0.bias.data, l2.weight.data, l2.bias.data
        elif head == "linear":
            l0 = nn.Linear(dim_in, 128)
            st["head.weight"], st["head.bias"] = l0.weight.data, l0.bias.data
    else:
        encoder("", n_cls)
        if data == "mini_imagenet":   # setup_elements.py:63-66: the classifier is re-created (a second RNG draw)
            lin = nn.Linear(640, n_cls)
            st["linear.weight"], st["linear.bias"] = lin.weight.data, lin.bias.data
    return clone_state(st)


class _Idx(torch.utils.data.Dataset):
    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        return i


def to_tensor(x_u8):
    """torchvision ToTensor on uint8 HWC (utils/setup_elements.py:29-43): CHW float32 / 255."""
    return torch.from_numpy(np.ascontiguousarray(x_u8)).permute(0, 3, 1, 2).contiguous().float().div(255)


class OracleAgent(object):
    """ONE run of agents/exp_replay.py / agents/scr.py + agents/base.py on CPU, built from the step functions above.
    `cfg` is an oracle.synth.STEP_CASES-style dict; cfg["trick"] may switch on review_trick / ncm_trick (agents/base.py:62-88,
    :121)."""

    def __init__(self, cfg, aug=identity_aug):
        self.cfg = dict(cfg)
        self.agent = cfg["agent"]
        self.data = cfg["data"]
        self.head = cfg.get("head", "mlp") if self.agent == "SCR" else None
        self.state = init_state(self.agent, self.data, cfg.get("head", "mlp"))
        self.names = [k for k in self.state if self.state[k].requires_grad]
        hw = {"cifar10": 32, "cifar100": 32, "mini_imagenet": 84}[self.data]
        self.buf = OracleBuffer(cfg["mem_size"], (3, hw, hw))
        self.cache = ClassCache()
        self.n_classes = {"cifar100": 100, "cifar10": 10, "mini_imagenet": 100}[self.data]
        self.p = dict(eps_mem_batch=cfg["eps_mem_batch"], temp=cfg.get("temp", 0.07), lr=cfg.get("lr", 0.1), head=self.head,
                      subsample=cfg.get("subsample", 50), k=cfg.get("k", 3), n_smp_cls=cfg.get("n_smp_cls", 1.5),
                      aser_type=cfg.get("aser_type", "asvm"), mem_size=cfg["mem_size"], n_classes=self.n_classes)
        self.trick = dict(cfg.get("trick", {}))
        self.old_labels = []
        self.aug = aug
        self.batch = cfg.get("batch", 10)
        self.log = []
        self.review_log = []
        self.gss = GssState(cfg) if cfg.get("update") == "GSS" else None
        self.tracker = ClassTracker(self.n_classes) if cfg.get("buffer_tracker") else None
        self.task_seen = 0          # base.py:37, incremented by after_train (:61)
        self.teacher = None         # kd_manager.teacher_model: a deep copy of the model, taken (and left) in train mode (kd_manager.py:18-19)

    def _kd_mix(self, loss, logits, x):
        """exp_replay.py:42-47 (and :64-69 for the memory pass): kd_trick blends with weight 1 / (t + 1), kd_trick_star with
        1 / sqrt(t + 1); the distillation term is 0 while there is no teacher (kd_manager.py:27-28) -- kd_trick_star alone never gets
        one (base.py:90), so it only scales the cross-entropy."""
        def kd_loss():
            if self.teacher is None:
                return 0
            with torch.no_grad():
                t_logits = OracleNet(self.teacher, head=None, training=True).forward(x)   # the copy's own (train-mode) forward
            return loss_fn_kd(logits, t_logits)
        if self.trick.get("kd_trick"):
            loss = 1 / (self.task_seen + 1) * loss + (1 - 1 / (self.task_seen + 1)) * kd_loss()
        if self.trick.get("kd_trick_star"):
            loss = 1 / ((self.task_seen + 1) ** 0.5) * loss + (1 - 1 / ((self.task_seen + 1) ** 0.5)) * kd_loss()
        return loss

    def train_learner(self, x_u8, y):
        new = list(set(y.tolist()))                                            # base.py:43-44
        xs = to_tensor(x_u8)
        ys = torch.from_numpy(np.asarray(y)).long()
        loader = torch.utils.data.DataLoader(_Idx(len(ys)), batch_size=self.batch, shuffle=True, drop_last=True)
        for idx in loader:
            bx, by = xs[idx], ys[idx]
            if self.agent == "SCR":
                self.log.append(scr_step(self.state, self.names, self.buf, bx, by, self.p, self.aug, retrieve=self.cfg.get("retrieve", "random"),
                                         tracker=self.tracker, warmup=self.cfg.get("warmup", 4)))
            elif self.cfg["retrieve"] == "ASER" or self.cfg["update"] == "ASER":
                self.log.append(aser_er_step(self.state, self.names, self.buf, self.cache, bx, by, self.p))
            else:
                kd = self._kd_mix if (self.trick.get("kd_trick") or self.trick.get("kd_trick_star")) else None
                self.log.append(er_step(self.state, self.names, self.buf, bx, by, self.p, self.cfg["retrieve"], gss=self.gss,
                                        tracker=self.tracker, warmup=self.cfg.get("warmup", 4), kd=kd))
        self.after_train(new)

    def after_train(self, new):
        """base.py:56-91: label bookkeeping, then (review trick) one epoch over the buffer at batch eps_mem_batch with the
        gradients divided by 10."""
        self.old_labels += new                                                 # base.py:58
        self.task_seen += 1                                                    # base.py:61
        if self.trick.get("review_trick"):
            self.review_log.append(review_epoch(self.state, self.names, self.buf, self.p, self.agent, self.aug))
        if self.trick.get("kd_trick"):                                         # base.py:90-91
            self.teacher = clone_state(self.state, requires_grad=False)

    def evaluate(self, tests, test_batch=128, detail=None):
        """base.py:118-227 (NCM for SCR / ncm_trick, argmax otherwise); test loaders shuffle (2 RNG draws each).  `detail`
        (a list) receives per task a dict(index=sample order, pred=predicted labels[, dist=NCM distance matrix])."""
        net = OracleNet(self.state, head=self.head, training=False)
        acc = np.zeros(len(tests))
        ncm = self.agent == "SCR" or self.trick.get("ncm_trick", False)
        with torch.no_grad():
            if ncm:
                n = self.buf.current_index
                for c in self.buf.label[:n].tolist():
                    if c not in self.old_labels:
                        raise KeyError(c)                                      # cls_exemplar[y.item()], base.py:126
                feats = deep_features(net, self.buf.img[:n]) if n else torch.zeros(0, 160)
                means = ncm_means(feats, self.buf.label[:n], self.old_labels)
                for i in range(len(self.old_labels)):
                    if torch.isnan(means[i]).any():
                        mu = torch.normal(0, 1, size=(1, means.shape[1])).squeeze()
                        means[i] = mu / mu.norm()
            for t, (x_u8, y) in enumerate(tests):
                xs, ys = to_tensor(x_u8), torch.from_numpy(np.asarray(y)).long()
                loader = torch.utils.data.DataLoader(_Idx(len(ys)), batch_size=test_batch, shuffle=True)
                tot, cnt = 0.0, 0
                rec = dict(index=[], pred=[], dist=[])
                for idx in loader:
                    bx, by = xs[idx], ys[idx]
                    if ncm:
                        f = net.features(bx)
                        f = f / f.norm(dim=1, keepdim=True)
                        d = ((f[:, None, :] - means[None, :, :]) ** 2).sum(-1)
                        pred = torch.tensor(self.old_labels)[d.min(1)[1]]
                        rec["dist"].append(d.numpy())
                    else:
                        pred = net.forward(bx).max(1)[1]
                    rec["index"].append(idx.numpy())
                    rec["pred"].append(pred.numpy())
                    c = (pred == by).sum().item() / by.size(0)
                    tot += c * by.size(0)
                    cnt += by.size(0)
                acc[t] = float(tot) / cnt
                if detail is not None:
                    detail.append({k: np.concatenate(v) for k, v in rec.items() if v})
        return acc

    def state_dict(self):
        return OrderedDict((k, v.detach()) for k, v in self.state.items())
