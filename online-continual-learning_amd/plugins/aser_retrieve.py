"""ASER retrieval -- the `retrieve_methods['ASER']` plugin (reference: utils/buffer/aser_retrieve.py:8-92).

Until the stream has seen more samples than the memory holds, retrieval is uniform.  Afterwards class-balanced memory samples are
the candidates; each is scored by its kNN Shapley value w.r.t. the incoming batch (adversarial: high value = close to current
samples of its own class) and w.r.t. a second class-balanced memory sample (cooperative), combined per `aser_type`
("asv": max coop - min adv, "asvm": mean coop - mean adv, "neg_sv": -sum adv), and the best `eps_mem_batch` are returned.
The candidates' features are extracted once (the reference extracts them twice): one eval-mode pass over batch + candidates, issued
before the host draws the cooperative samples, and one over those; scoring, ranking and the final gather stay on the GPU, nothing is
copied back to the host."""
import torch

from .. import debug
from .. import ops
from ..setup_elements import n_classes
from ..utils import maybe_cuda
import os

from .aser_utils import compute_knn_sv, compute_knn_sv_pair, features_begin
from .buffer_utils import ClassBalancedRandomSampling, random_retrieve


def _split_features():
    return os.environ.get("OCL_ASER_SPLIT", "1") != "0"


class ASER_retrieve(object):
    def __init__(self, params, **kwargs):
        self.num_retrieve = params.eps_mem_batch
        self.device = "cuda" if torch.cuda.is_available() else "cpu"
        self.k = params.k
        self.mem_size = params.mem_size
        self.aser_type = params.aser_type
        self.n_smp_cls = int(params.n_smp_cls)
        self.out_dim = n_classes[params.data]
        self.is_aser_upt = params.update == "ASER"
        ClassBalancedRandomSampling.class_index_cache = None     # class-level state, reset per plugin instance (:19)
        # (every mutation of the class sets goes through update_cache: the C helper's memo is verified one draw in 64, buffer_utils.py)
        ClassBalancedRandomSampling.verify_every = int(__import__("os").environ.get("OCL_CBRS_VERIFY_EVERY", "64"))

    def retrieve(self, buffer, **kwargs):
        if buffer.n_seen_so_far <= self.mem_size:                 # memory not yet cycled once: uniform retrieval (:24-26)
            return random_retrieve(buffer, self.num_retrieve)
        return self._by_shapley_value(buffer, maybe_cuda(kwargs['x']), maybe_cuda(kwargs['y']))

    def _by_shapley_value(self, buffer, cur_x, cur_y):
        sampler = ClassBalancedRandomSampling
        if not self.is_aser_upt:   # without the ASER update nobody maintains the class cache: rebuild it from the labels (:42-43)
            sampler.update_cache(buffer.label_host, self.out_dim)
        cand_x, cand_y, cand_slots = sampler.sample(buffer.buffer_img, buffer.buffer_label, self.n_smp_cls, device=self.device)
        trace = debug.on()
        order_adv = order_coop = None
        if self.aser_type == "neg_sv":
            adv = compute_knn_sv(buffer.model, cur_x, cur_y, cand_x, cand_y, self.k, device=self.device, want_order=trace)
            if trace:
                adv, order_adv = adv
            coop = None
        else:
            # the cooperative evaluation set is drawn before any scoring (no RNG draw lies between, so the streams are those of the
            # reference's order :56-76) and both matrices come out of one feature pass
            # Round 6: the feature pass over batch + candidates is ISSUED before the cooperative draw: that draw excludes the candidates'
            # slots -- a real set difference per class on the host, ~0.2 ms with the GPU's queue empty (the step's one synchronisation lies
            # just behind) -- and now runs beside ~0.3 ms of GPU work; the cooperative samples' features follow in a second, smaller pass
            # (eval-mode features are per sample).  OCL_ASER_SPLIT=0: one pass over all three pieces, after both draws.
            pending = features_begin(buffer.model, cur_x, cand_x) if _split_features() else None
            coop_x, coop_y, _ = sampler.sample(buffer.buffer_img, buffer.buffer_label, self.n_smp_cls, excl_indices=set(cand_slots.tolist()),
                                               device=self.device)
            adv, coop = compute_knn_sv_pair(buffer.model, cur_x, cur_y, coop_x, coop_y, cand_x, cand_y, self.k, want_order=trace, begun=pending)
            if trace:
                (adv, order_adv), (coop, order_coop) = adv, coop
        score = ops.aser_score(adv, coop, self.aser_type)
        best = ops.argsort_desc(score)[:self.num_retrieve].contiguous()
        if trace:
            debug.emit("aser_retrieve", cand_ind=cand_slots.numpy().copy(), sv=score.cpu().numpy(), ret=cand_slots[best.cpu()].numpy(),
                       order_adv=order_adv.cpu().numpy(), order_coop=None if order_coop is None else order_coop.cpu().numpy())
        return ops.gather_pair(cand_x, cand_y, best)
