"""utils/buffer/aser_utils.py: compute_knn_sv (:7-61), deep_features (:64-91), add_minority_class_input (:119-157).
Distance matrix, per-row sort, indicator recursion and scatter are ONE HIP kernel (ocl_knn_sv); the deep features
come from the engine's eval-mode forward."""
import torch

from .. import ops
from ..utils import maybe_cuda, mini_batch_deep_features, nonzero_indices
from .buffer_utils import ClassBalancedRandomSampling, _host_labels


def deep_features(model, eval_x, n_eval, cand_x, n_cand):
    """aser_utils.py:64-91."""
    # eval_x / cand_x may each be a tensor or a sequence of tensors (the pieces of the reference's torch.cat calls, e.g. class-balanced
    # memory samples + minority batch items): the engine reads the pieces where they are, nothing is concatenated
    def pieces(t):
        return [maybe_cuda(p) for p in (t if isinstance(t, (list, tuple)) else (t,))]
    if cand_x is None:
        num = n_eval
        total_x = pieces(eval_x)
    else:
        num = n_eval + n_cand
        total_x = pieces(eval_x) + pieces(cand_x)
    deep_features_ = mini_batch_deep_features(model, total_x, num)
    eval_df = deep_features_[0:n_eval]
    cand_df = deep_features_[n_eval:]
    return eval_df, cand_df


def compute_knn_sv(model, eval_x, eval_y, cand_x, cand_y, k, device="cpu", want_order=False):
    """aser_utils.py:7-61: KNN Shapley value matrix [n_eval, n_cand] of candidates w.r.t. evaluation data.
    want_order (parity tests): also return the per-row ascending-distance candidate order the kernel used."""
    def rows(t):
        return sum(p.size(0) for p in t) if isinstance(t, (list, tuple)) else t.size(0)
    n_eval = rows(eval_x)
    n_cand = rows(cand_x)
    eval_df, cand_df = deep_features(model, eval_x, n_eval, cand_x, n_cand)
    return ops.knn_sv(eval_df.contiguous(), eval_y, cand_df.contiguous(), cand_y, k, want_order=want_order)


def features_begin(model, eval_a_x, cand_x):
    """The eval-mode feature pass over eval_a + candidates, issued now (compute_knn_sv_pair(begun=...) takes it from here)."""
    na, nc = eval_a_x.size(0), cand_x.size(0)
    return mini_batch_deep_features(model, [maybe_cuda(eval_a_x), maybe_cuda(cand_x)], na + nc)


def compute_knn_sv_pair(model, eval_a_x, eval_a_y, eval_b_x, eval_b_y, cand_x, cand_y, k, want_order=False, begun=None):
    """Two compute_knn_sv calls over the SAME candidates (aser_retrieve.py:56-76: adversarial and cooperative Shapley values)
    with ONE eval-mode feature pass over eval_a + eval_b + candidates: eval-mode features are per-sample, so the candidates'
    features (computed twice by the reference) are the same in both calls."""
    na, nb, nc = eval_a_x.size(0), eval_b_x.size(0), cand_x.size(0)
    if begun is not None:   # features of eval_a + candidates are on their way (features_begin): eval_b's in a pass of their own
        fb = mini_batch_deep_features(model, [maybe_cuda(eval_b_x)], nb) if nb > 0 else begun.new_zeros((0, begun.shape[1]))   # (no class has a second sample)
        fab, fc = torch.cat((begun[0:na], fb)), begun[na:].contiguous()
    else:
        f = mini_batch_deep_features(model, [maybe_cuda(eval_a_x), maybe_cuda(eval_b_x), maybe_cuda(cand_x)], na + nb + nc)
        fab, fc = f[0:na + nb].contiguous(), f[na + nb:].contiguous()
    # one workgroup per evaluation row, rows independent: both evaluation sets are ONE launch over the stacked rows
    sv = ops.knn_sv(fab, torch.cat((eval_a_y, eval_b_y)), fc, cand_y, k, want_order=want_order)
    if want_order:
        sv, order = sv
        return (sv[:na], order[:na]), (sv[na:], order[na:])
    return sv[:na], sv[na:]


def add_minority_class_input(cur_x, cur_y, mem_size, num_class, cur_y_host=None):
    """aser_utils.py:119-157.  Threshold ~ U(0, 1/num_class) on the torch CPU generator; class counts come from
    ClassBalancedRandomSampling.class_num_cache (host)."""
    # Select input instances from minority classes that will be concatenated to pre-selected data
    threshold = torch.tensor(1).float().uniform_(0, 1 / num_class).item()

    # If number of buffered samples from certain class is lower than random threshold,
    #   that class is minority class
    cls_proportion = ClassBalancedRandomSampling.class_num_cache.float() / mem_size
    cur_y_cpu = torch.from_numpy(_host_labels(cur_y, cur_y_host))
    minority_ind = nonzero_indices(cls_proportion[cur_y_cpu] < threshold)

    minority_batch_x, minority_batch_y = ops.gather_pair(cur_x.contiguous(), cur_y.contiguous(), minority_ind)
    return minority_batch_x, minority_batch_y
