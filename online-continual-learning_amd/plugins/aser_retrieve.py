"""utils/buffer/aser_retrieve.py:8-92 — ASER retrieval: class-balanced candidates, adversarial + cooperative kNN
Shapley values, top-N.  Host RNG / set bookkeeping as in the reference; scoring and selection on the GPU with no
device->host synchronisation."""
import torch

from .. import ops
from .. import debug
from ..setup_elements import n_classes
from ..utils import maybe_cuda
from .aser_utils import compute_knn_sv, compute_knn_sv_pair
from .buffer_utils import ClassBalancedRandomSampling, random_retrieve


class ASER_retrieve(object):
    def __init__(self, params, **kwargs):
        super().__init__()
        self.num_retrieve = params.eps_mem_batch
        self.device = "cuda" if torch.cuda.is_available() else "cpu"
        self.k = params.k
        self.mem_size = params.mem_size
        self.aser_type = params.aser_type
        self.n_smp_cls = int(params.n_smp_cls)
        self.out_dim = n_classes[params.data]
        self.is_aser_upt = params.update == "ASER"
        ClassBalancedRandomSampling.class_index_cache = None

    def retrieve(self, buffer, **kwargs):
        model = buffer.model

        if buffer.n_seen_so_far <= self.mem_size:
            # Use random retrieval until buffer is filled
            ret_x, ret_y = random_retrieve(buffer, self.num_retrieve)
        else:
            # Use ASER retrieval if buffer is filled
            cur_x, cur_y = kwargs['x'], kwargs['y']
            ret_x, ret_y = self._retrieve_by_knn_sv(model, buffer, cur_x, cur_y, self.num_retrieve)
        return ret_x, ret_y

    def _retrieve_by_knn_sv(self, model, buffer, cur_x, cur_y, num_retrieve):
        """aser_retrieve.py:34-92."""
        buffer_x, buffer_y = buffer.buffer_img, buffer.buffer_label
        cur_x = maybe_cuda(cur_x)
        cur_y = maybe_cuda(cur_y)

        # Reset and update ClassBalancedRandomSampling cache if ASER update is not enabled
        if not self.is_aser_upt:
            ClassBalancedRandomSampling.update_cache(buffer.label_host, self.out_dim)

        # Get candidate data for retrieval (i.e., cand <- class balanced subsamples from memory)
        cand_x, cand_y, cand_ind = \
            ClassBalancedRandomSampling.sample(buffer_x, buffer_y, self.n_smp_cls, device=self.device)

        # Type 1 - Adversarial SV: eval <- current input
        eval_adv_x, eval_adv_y = cur_x, cur_y
        dbg = debug.on()
        order_adv = order_coop = None

        if self.aser_type != "neg_sv":
            # Type 2 - Cooperative SV: eval <- class balanced subsamples from memory excluding the candidates.  Sampled before
            # the adversarial values are computed (nothing in between draws from an RNG), so that both Shapley matrices come
            # from ONE feature pass over current input + cooperative samples + candidates.
            excl_indices = set(cand_ind.tolist())
            eval_coop_x, eval_coop_y, _ = \
                ClassBalancedRandomSampling.sample(buffer_x, buffer_y, self.n_smp_cls,
                                                   excl_indices=excl_indices, device=self.device)
            sv_matrix_adv, sv_matrix_coop = compute_knn_sv_pair(model, eval_adv_x, eval_adv_y, eval_coop_x, eval_coop_y, cand_x, cand_y,
                                                                self.k, want_order=dbg)
            if dbg:
                sv_matrix_adv, order_adv = sv_matrix_adv
                sv_matrix_coop, order_coop = sv_matrix_coop
            sv = ops.aser_score(sv_matrix_adv, sv_matrix_coop, self.aser_type)
        else:
            sv_matrix_adv = compute_knn_sv(model, eval_adv_x, eval_adv_y, cand_x, cand_y, self.k, device=self.device, want_order=dbg)
            if dbg:
                sv_matrix_adv, order_adv = sv_matrix_adv
            sv = ops.aser_score(sv_matrix_adv, None, "neg_sv")

        ret_ind = ops.argsort_desc(sv)[:num_retrieve].contiguous()
        if debug.on():
            debug.emit("aser_retrieve", cand_ind=cand_ind.numpy().copy(), sv=sv.cpu().numpy(), ret=cand_ind[ret_ind.cpu()].numpy(),
                       order_adv=order_adv.cpu().numpy(), order_coop=None if order_coop is None else order_coop.cpu().numpy())

        ret_x = ops.gather_rows(cand_x, ret_ind)
        ret_y = ops.gather_rows(cand_y, ret_ind)
        return ret_x, ret_y
