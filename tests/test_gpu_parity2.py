"""GPU: the parity holes named by the round-1 review, closed.

 * evaluate() end to end (agents/base.py:118-176) teacher-forced against the oracle: accuracy arrays equal, every prediction
   equal or a distance near-tie, the exemplar-less class draw and the loaders' RNG draws in step;
 * the review trick (agents/base.py:62-88) co-simulated batch by batch;
 * ER + ASER at BASELINE size (mem 5000, 100 classes, k = 3) against the oracle for five steps, tie-aware;
 * MIR with the FULL 10 000-slot 84x84 memory and the engine's own gradient: the interference scores are the oracle's for the
   engine's virtual step (1e-4), and the engine's gradient is the oracle's (bounded separately);
 * the per-step SGD update: 1e-2 free-running, and with the ReLU activation pattern teacher-forced the gradients at the very
   states where the free-running error is largest agree to 2e-4 (what is left is ReLU sign flips at ~0);
 * free-running trajectories: the HIP end state sits inside the spread the oracle itself shows under fp32-rounding-level
   perturbations (thread counts, 1-ulp initial-weight perturbation);
 * GSS-Greedy update (eval-mode gradients on the engine) co-simulated; match retrieval / tracker via the golden runs;
 * the sharded driver: run.single_run on a 2-task stream.
"""
import copy
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import gold, ROOT
from oracle import ocl_oracle as O
from oracle.synth import STEP_CASES, make_stream, seed_all, digest_state, class_images
from test_gpu_steps import (build_agent, cosim, make_params, _buffers_equal, _rng_get, _rng_set, _rng_equal, _prefill_fn, _sv_given_order,
                            _flat, TRICK)

pytestmark = pytest.mark.gpu


def _load_oracle_into(agent, model, oa):
    """Teacher forcing: weights, BatchNorm buffers, replay memory, counters and label bookkeeping of the oracle agent."""
    model.load_state_dict(oa.state_dict())
    n = oa.buf.current_index
    b = agent.buffer
    dev = b.buffer_img.device
    b.buffer_img[:n] = oa.buf.img[:n].to(dev)
    b.buffer_label[:n] = oa.buf.label[:n].to(dev)
    b.label_host[:n] = oa.buf.label[:n].numpy()
    b.current_index, b.n_seen_so_far = oa.buf.current_index, oa.buf.n_seen_so_far
    agent.old_labels = list(oa.old_labels)
    agent.new_labels = []
    agent.task_seen = 1


# ---------------------------------------------------------------------------------------------------------------------
# evaluate()
# ---------------------------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("name", ["scr_tiny", "scr_c100", "er_review", "er_c10"])
def test_evaluate_teacher_forced_vs_oracle(cuda, name):
    """Both sides hold the oracle's end-of-task state and the same host RNG state, then call evaluate(): accuracy arrays equal
    (they are ratios of integer counts), predictions equal sample by sample except where the oracle's own two best class
    distances / logits are within 1e-5 of each other, host RNG left in the same state (2 draws per shuffled test loader + the
    torch.normal draw of an exemplar-less class, agents/base.py:135-137)."""
    from ocl_amd import debug
    from ocl_amd.data import setup_test_loader
    cfg = STEP_CASES[name]
    params, model, agent = build_agent(cfg)
    seed_all(cfg["seed"])
    oa = O.OracleAgent(dict(cfg, trick=dict(cfg.get("trick", {}), review_trick=False)))
    tasks, tests = make_stream(cfg)
    loaders = setup_test_loader(tests, params)
    for t, (x, y) in enumerate(tasks):
        oa.train_learner(x, y)
        _load_oracle_into(agent, model, oa)
        st = _rng_get()
        detail = []
        acc_o = oa.evaluate(tests, detail=detail)
        st_o = _rng_get()
        _rng_set(st)
        debug.LOG = []
        try:
            acc_h = agent.evaluate(loaders)
            ev = [e for tag, e in debug.LOG if tag == "evaluate"]
        finally:
            debug.LOG = None
        assert _rng_equal(st_o, _rng_get()), "evaluate() consumed the host RNG differently (task %d)" % t
        n_diff = 0
        for k, (e, d) in enumerate(zip(ev, detail)):
            assert np.array_equal(e["index"], d["index"]), "test loader order differs"
            diff = np.nonzero(e["pred"] != d["pred"])[0]
            for i in diff:      # a different prediction is only acceptable on a near-tie of the oracle's own scores
                assert "dist" in d, "argmax predictions differ"
                row = np.sort(d["dist"][i])
                assert row[1] - row[0] < 1e-5 * max(1.0, row[1]), (name, t, k, int(i), row[:3])
            n_diff += len(diff)
        print(name, "task", t, "acc", acc_h, acc_o, "near-tie prediction differences:", n_diff)
        if n_diff == 0:
            assert np.array_equal(acc_h, acc_o)
        else:
            assert np.abs(acc_h - acc_o).max() <= n_diff / min(len(y) for _, y in tests)
    if name == "scr_tiny":   # 3 slots, 4 seen classes: the random-mean branch ran
        assert len(set(oa.buf.label[:oa.buf.current_index].tolist())) < len(oa.old_labels)


# ---------------------------------------------------------------------------------------------------------------------
# review trick
# ---------------------------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("name", ["er_review", "scr_review"])
def test_review_trick_cosim(cuda, name):
    """after_train()'s review pass (agents/base.py:62-88) from the oracle's end-of-task state: same shuffled buffer batches
    (DataLoader RNG draws), per-batch loss within 1e-4, the accumulated update of the whole pass within 1e-2 of the oracle's,
    BatchNorm running statistics within 1e-4, num_batches_tracked equal (SCR: three forwards per batch, :77-80)."""
    from ocl_amd import debug
    cfg = STEP_CASES[name]
    params, model, agent = build_agent(cfg)
    assert params.trick["review_trick"]
    seed_all(cfg["seed"])
    oa = O.OracleAgent(dict(cfg, trick=dict(cfg["trick"], review_trick=False)))   # the review pass is driven explicitly below
    tasks, _ = make_stream(cfg)
    x, y = tasks[0]
    oa.train_learner(x, y)
    _load_oracle_into(agent, model, oa)
    agent.old_labels, agent.new_labels, agent.task_seen = [], list(oa.old_labels), 0     # the state after_train() starts from
    w0 = _flat(oa.state, oa.names)
    st = _rng_get()
    rec = O.review_epoch(oa.state, oa.names, oa.buf, oa.p, oa.agent, oa.aug)
    st_o = _rng_get()
    _rng_set(st)
    debug.LOG = []
    try:
        agent.after_train()
        ev = [e for tag, e in debug.LOG if tag == "review"]
    finally:
        debug.LOG = None
    assert _rng_equal(st_o, _rng_get())
    assert len(rec) == len(ev) == oa.buf.current_index // cfg["eps_mem_batch"] >= 2
    for j, ((idx, loss), e) in enumerate(zip(rec, ev)):
        assert np.array_equal(idx, e["indices"])
        # first batch: identical weights on both sides (1e-4); later batches run on free-running weights
        assert abs(loss - e["loss"]) < (1e-4 if j == 0 else 1e-3) * max(1.0, abs(loss)), (j, loss, e["loss"])
    dw_o = _flat(oa.state, oa.names) - w0
    dw_h = model.flat_params().double().cpu().numpy() - w0
    err = np.linalg.norm(dw_h - dw_o) / np.linalg.norm(dw_o)
    print(name, "review pass:", len(rec), "batches, update err", err)
    assert err < 1e-2
    sd = model.state_dict()
    for k, v in oa.state_dict().items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            assert np.abs(sd[k].cpu().numpy() - v.numpy()).max() < 1e-4 * max(1.0, float(v.abs().max())), k
        if k.endswith("num_batches_tracked"):
            assert int(sd[k]) == int(v), k
    assert agent.old_labels == oa.old_labels and agent.task_seen == 1


# ---------------------------------------------------------------------------------------------------------------------
# ASER at BASELINE size, against the oracle
# ---------------------------------------------------------------------------------------------------------------------

def test_aser_cosim_at_baseline_size(cuda):
    """BASELINE config 3 (ER, ASER retrieve + update, mem_size 5000 full with all 100 classes, k = 3, n_smp_cls 1.5, asvm): five
    teacher-forced steps against the oracle.  Candidate / evaluation index sets identical (100 class-balanced candidates, 100
    cooperative samples, 150 + 10 update candidates: torch-CPU randperm per class, CPython set order, numpy choice), the
    kernel's kNN order a valid ascending order of the oracle's distances, Shapley scores within 1e-5 given that order,
    selections valid top-N under the oracle's scores, combined-batch loss within 1e-4 whenever both sides retrieved the same
    rows.  Exact SV ties are ubiquitous and torch's argsort is unstable, so once a tie is ordered differently the buffers
    legitimately differ: both sides are re-synchronised from the oracle after every step."""
    from ocl_amd.plugins.buffer_utils import ClassBalancedRandomSampling as CB
    cfg = dict(agent="ER", retrieve="ASER", update="ASER", data="cifar100", mem_size=5000, eps_mem_batch=10, seed=33,
               tasks=[list(range(100))], n_train=1, n_test=1, k=3, n_smp_cls=1.5, aser_type="asvm")
    rng = np.random.default_rng(12)
    protos = np.random.default_rng(13).random((100, 3, 32, 32)).astype(np.float32)

    def prefill(agent, oa):
        # what ASER_update.update does while the memory fills (aser_update.py:27-36: class cache, then reservoir append) on both
        # sides, slot by slot -- without its full-memory branch, which would run with an empty batch on the chunk that completes
        # the fill
        for s in range(0, 5000, 500):
            ys = np.concatenate([np.arange(100), rng.integers(0, 100, 400)]).astype(np.int64)   # every class present
            xs = (0.5 * protos[ys] + 0.5 * rng.random((500, 3, 32, 32), dtype=np.float32)).astype(np.float32)
            b = agent.buffer
            CB.update_cache(b.label_host, 100, new_y=ys, ind=list(range(s, s + 500)))
            b.update_method.reservoir_update.update(b, torch.from_numpy(xs).to(cuda), torch.from_numpy(ys).to(cuda), y_host=ys)
            oa.cache.update(oa.buf.label, 100, new_y=torch.from_numpy(ys), ind=range(s, s + 500))
            O.reservoir_update(oa.buf, torch.from_numpy(xs), torch.from_numpy(ys))
        assert agent.buffer.current_index == oa.buf.current_index == 5000 == agent.buffer.n_seen_so_far
        # n_seen_so_far == mem_size after the fill: the first step retrieves at random, its update pushes n_seen past mem_size
    xs = np.concatenate([class_images(c, 1, (32, 32), np.random.default_rng(500 + c)) for c in range(60)], 0)
    ys = np.arange(60, dtype=np.int64)
    eps = 1e-5
    n_ret = n_upd = 0
    for it, ev, ol, chk in cosim(cfg, 6, cuda, prefill=prefill, x_stream=(xs, ys)):
        assert chk["rng_equal"], "host RNG streams diverged at iteration %d" % it
        agent, oa = chk["agent"], chk["oa"]
        ret_ev = [e for t, e in ev if t == "aser_retrieve"]
        upd_ev = [e for t, e in ev if t == "aser_update"]
        same_rows = False
        if ol.get("cand") is not None:
            r = ret_ev[0]
            assert len(r["cand_ind"]) == 100 and np.array_equal(r["cand_ind"], ol["cand"]), "candidate set differs (iteration %d)" % it
            sv_adv = _sv_given_order(ol["ret_aux"]["adv"], r["order_adv"], cfg["k"])
            sv_coop = _sv_given_order(ol["ret_aux"]["coop"], r["order_coop"], cfg["k"])
            assert sv_coop.shape == (100, 100) and sv_adv.shape == (10, 100)
            sv_exp = O.aser_score(sv_adv, sv_coop, "asvm")
            assert np.abs(r["sv"] - sv_exp).max() < eps, np.abs(r["sv"] - sv_exp).max()
            thr = np.sort(sv_exp)[::-1][9]
            pos = {c: j for j, c in enumerate(ol["cand"].tolist())}
            assert len(r["ret"]) == 10 and min(sv_exp[pos[c]] for c in r["ret"].tolist()) >= thr - eps, "not a valid top-10"
            same_rows = np.array_equal(np.sort(r["ret"]), np.sort(ol["ret_idx"]))
            n_ret += 1
        else:
            assert not ret_ev and np.array_equal([e["indices"] for t, e in ev if t == "random_retrieve"][0], ol["ret_idx"])
            same_rows = True
        if same_rows:
            assert abs([e["loss"] for t, e in ev if t == "er_loss_combined"][0] - ol["loss"]) < 1e-4 * (1 + abs(ol["loss"]))
            assert chk["upd_err"] < 1e-2, chk["upd_err"]
        u, ou = upd_ev[0], ol["upd"]
        assert len(u["eval_indices"]) == 100 and len(u["cand_ind"]) == 150
        assert np.array_equal(u["eval_indices"], ou["eval_indices"]) and np.array_equal(u["cand_ind"], ou["cand_ind"])
        assert u["n_minority"] == ou["n_minority"]
        if same_rows:   # the update's features come from the weights after this step's SGD update: comparable only then
            sv_exp = _sv_given_order(ou["aux"], u["knn_order"], cfg["k"]).sum(0)
            assert np.abs(u["sv"] - sv_exp).max() < eps * 10, np.abs(u["sv"] - sv_exp).max()   # sum over 100+ rows
            n_buf = 150
            thr = np.sort(sv_exp)[::-1][n_buf - 1]
            large, small = u["order"][:n_buf], u["order"][n_buf:]
            assert sv_exp[large].min() >= thr - 1e-4 and (len(small) == 0 or sv_exp[small].max() <= thr + 1e-4), "invalid SV partition"
            n_upd += 1
        # a legitimate tie divergence must not leak into the next step: copy the oracle's memory and rebuild BOTH class caches from the
        # labels (the reference's own reset path, buffer_utils.py:156-160), which gives both sides sets with identical histories
        if not _buffers_equal(agent, oa):
            print("tie divergence at iteration", it, "- re-synchronising")
            b = agent.buffer
            b.buffer_img.copy_(oa.buf.img.to(cuda))
            b.buffer_label.copy_(oa.buf.label.to(cuda))
            b.label_host[:] = oa.buf.label.numpy()
            CB.update_cache(b.label_host, 100)
            oa.cache.update(oa.buf.label, 100)
            CB.class_num_cache = oa.cache.count.clone()
    print("ASER at BASELINE size: %d retrievals, %d updates compared" % (n_ret, n_upd))
    assert n_ret >= 4 and n_upd >= 3


# ---------------------------------------------------------------------------------------------------------------------
# MIR with the full memory and the engine's own gradient
# ---------------------------------------------------------------------------------------------------------------------

def test_mir_full_memory_free_gradient(cuda):
    """BASELINE config 4: ER + MIR, Mini-ImageNet 84x84, mem_size 10000 FULL (847 MB on either side), subsample 50 -> 10.
    Nothing is injected: the plugin reads the engine's own gradient.  The comparison with the oracle is decomposed:
      (1) candidate subsample identical (numpy RNG over 10000 slots);
      (2) the engine's gradient vector is the oracle's to 5e-3 in norm (observed 1.3e-3 - 2.1e-3 depending on the summation order of
          the conv kernels: ReLU sign flips at ~0; with the activation pattern teacher-forced the figure is 2e-4, test_gpu_net);
      (3) for the ENGINE's virtual step -- the oracle's two scoring forwards evaluated at theta and theta - lr * g_engine -- the
          interference scores agree to 1e-4 and the retrieved set is a valid top-10 of them: the scoring path itself is exact;
      (4) hence |s_engine - s_oracle| <= 1e-4 + |s_oracle(g_engine) - s_oracle(g_oracle)|: whatever free-running difference
          remains is the ORACLE's own response to the gradient difference of (2), computed here on the CPU and asserted."""
    import ocl_amd.plugins.mir_retrieve as mr
    cfg = dict(agent="ER", retrieve="MIR", update="random", data="mini_imagenet", mem_size=10000, eps_mem_batch=10, seed=43,
               tasks=[[3, 4]], n_train=5, n_test=1, subsample=50)
    grabbed = {}
    orig = mr.get_grad_vector

    def spy(model):
        g = orig(model)
        grabbed["g"] = g.detach().cpu().clone()
        return g
    mr.get_grad_vector = spy
    try:
        for it, ev, ol, chk in cosim(cfg, 1, cuda, prefill=_prefill_fn(10000, 20000, list(range(20, 40)), 84, 79)):
            assert chk["rng_equal"]
            assert np.array_equal([e["indices"] for t, e in ev if t == "random_retrieve"][0], ol["sub"]) and len(ol["sub"]) == 50
            mir_ev = [e for t, e in ev if t == "mir"][0]
            oa = chk["oa"]
            g_h, g_o = grabbed["g"], ol["grad"]
            g_err = float((g_h - g_o).norm() / g_o.norm())
            sub = torch.from_numpy(ol["sub"])
            sc_at_h = O.mir_scores_for_gradient(chk["state_before"], oa.names, g_h, chk["buf_before"][0][sub], chk["buf_before"][1][sub], lr=0.1)
            sc_h, sc_o = mir_ev["scores"], ol["scores"]
            scale = 1 + np.abs(sc_o).max()
            induced = np.abs(sc_at_h - sc_o).max()
            print("MIR full memory: |dg|/|g| = %.2e; score err vs oracle(g_engine) = %.2e; oracle(g_engine) vs oracle(g_oracle) = %.2e; "
                  "free-running = %.2e (scale %.2f)" % (g_err, np.abs(sc_h - sc_at_h).max(), induced, np.abs(sc_h - sc_o).max(), scale))
            assert g_err < 5e-3
            assert np.abs(sc_h - sc_at_h).max() < 1e-4 * scale
            thr = np.sort(sc_at_h)[::-1][9]
            assert sc_at_h[mir_ev["big_ind"]].min() >= thr - 1e-4 * scale
            assert np.abs(sc_h - sc_o).max() <= 1e-4 * scale + induced
            assert induced < 1e-2 * scale
            a = chk["agent"]
            assert [a.buffer.current_index, a.buffer.n_seen_so_far] == [oa.buf.current_index, oa.buf.n_seen_so_far] == [10000, 20010]
            assert np.array_equal(a.buffer.buffer_label.cpu().numpy(), oa.buf.label.numpy())
    finally:
        mr.get_grad_vector = orig


# ---------------------------------------------------------------------------------------------------------------------
# the per-step update error is ReLU sign flips
# ---------------------------------------------------------------------------------------------------------------------

def _double_state(sd):
    st = O.OrderedDict()
    for k, v in sd.items():
        t = v.detach().clone()
        if t.is_floating_point():
            t = t.double()
            if not (k.endswith("running_mean") or k.endswith("running_var")):
                t.requires_grad_(True)
        st[k] = t
    return st


def test_er_step_gradient_with_forced_relu_pattern_at_early_iterations(cuda):
    """The free-running ER co-simulation shows its largest update errors (1e-3) at iterations 1-2, when the memory batch is the
    previous stream batch and BatchNorm statistics come from ten samples (near-constant channels: 1/std amplifies round-off).
    At exactly those states (the oracle's weights after 1 and 2 iterations; the stream batch and the memory batch of that iteration
    as the two groups of one pass) three gradients are computed with the SAME ReLU activation pattern (the engine's): the
    engine's, ATen's in fp32, and ATen's in fp64 as ground truth.  The engine must be as close to the fp64 gradient as ATen's own
    fp32 arithmetic is (within 3x, per tensor, floor 2e-4 of the tensor's max), and every element whose pattern differs from
    ATen's has |pre-activation| < 1e-5 of the layer's max: what the free-running comparison sees is fp32 conditioning plus sign
    flips at ~0, not an arithmetic difference."""
    from test_gpu_net import engine_masks
    from ocl_amd.loss import cross_entropy_mean
    cfg = dict(STEP_CASES["er_c10"], mem_size=30)
    params, model, agent = build_agent(cfg)
    seed_all(cfg["seed"])
    oa = O.OracleAgent(cfg)
    tasks, _ = make_stream(cfg)
    xs, ys = tasks[0]
    seed_all(1000 + cfg["seed"])
    for it in range(3):
        bx, by = O.to_tensor(xs[it * 10:(it + 1) * 10]), torch.from_numpy(ys[it * 10:(it + 1) * 10])
        if it >= 1:
            sd = {k: v.clone() for k, v in oa.state_dict().items()}
            st_rng = _rng_get()
            idx = O.random_retrieve_indices(oa.buf, 10)
            _rng_set(st_rng)
            mx, my = oa.buf.img[idx], oa.buf.label[idx]
            assert len(my) == len(by)
            groups = [(bx, by), (mx, my)]
            model.load_state_dict(sd)
            model.train()
            out = model.forward_views([bx.to(cuda), mx.to(cuda)])
            loss = cross_entropy_mean(out[:10], by.to(cuda)) + cross_entropy_mean(out[10:], my.to(cuda))
            agent.opt.zero_grad()
            probe = O.OracleNet(O.clone_state(sd, requires_grad=False), head=None, training=True)
            probe.pre_act = {}
            with torch.no_grad():
                probe.forward(bx)
            shapes = {k: (20,) + tuple(v.shape[1:]) for k, v in probe.pre_act.items()}
            masks = engine_masks(model, shapes, "")
            loss.backward()
            grads = {}
            for tag, st, cast in (("f32", O.clone_state(sd), lambda t: t), ("f64", _double_state(sd), lambda t: t.double())):
                net = O.OracleNet(st, head=None, training=True)
                lref, off, n_mis, worst_amb = 0.0, 0, 0, 0.0
                for gx, gy in groups:
                    net.mask_override = {k: cast(v[off:off + len(gy)]) for k, v in masks.items()}
                    net.pre_act = {}
                    lref = lref + torch.nn.functional.cross_entropy(net.forward(cast(gx)), gy)
                    for k, pa in net.pre_act.items():
                        mis = (pa > 0) != (net.mask_override[k] > 0)
                        n_mis += int(mis.sum())
                        if mis.any():
                            worst_amb = max(worst_amb, float(pa[mis].abs().max() / pa.abs().max()))
                    off += len(gy)
                lref.backward()
                grads[tag] = ({k: st[k].grad.double() for k in st if st[k].requires_grad}, float(lref.detach()), n_mis, worst_amb)
            g64, l64, _, _ = grads["f64"]
            g32, l32, n_mis, worst_amb = grads["f32"]
            worst_ratio, worst_h, worst_o = 0.0, 0.0, 0.0
            for k, p in model.named_parameters():
                den = 1e-30 + float(g64[k].abs().max())
                e_h = float((p.grad.cpu().double() - g64[k]).abs().max()) / den
                e_o = float((g32[k] - g64[k]).abs().max()) / den
                worst_h, worst_o = max(worst_h, e_h), max(worst_o, e_o)
                assert e_h <= max(2e-4, 3 * e_o), (k, e_h, e_o)
            print("ER iteration %d: error vs the fp64 gradient: engine %.2e, ATen fp32 %.2e; %d pattern mismatches (worst ambiguity %.1e)"
                  % (it, worst_h, worst_o, n_mis, worst_amb))
            assert abs(float(loss.detach()) - l64) < 1e-4 and worst_amb < 1e-5
        oa.train_learner(xs[it * 10:(it + 1) * 10], ys[it * 10:(it + 1) * 10])


# ---------------------------------------------------------------------------------------------------------------------
# free-running trajectories
# ---------------------------------------------------------------------------------------------------------------------

def _oracle_run(cfg, threads=None, ulp=0):
    if threads:
        torch.set_num_threads(threads)
    seed_all(cfg["seed"])
    oa = O.OracleAgent(cfg)
    if ulp:      # every initial weight moved by one unit in the last place (up / down): the smallest perturbation fp32 can express
        with torch.no_grad():
            for k in oa.names:
                oa.state[k].copy_(torch.nextafter(oa.state[k].detach(), torch.full_like(oa.state[k], float("inf") * ulp)))
    tasks, tests = make_stream(cfg)
    accs = []
    for x, y in tasks:
        oa.train_learner(x, y)
        accs.append(oa.evaluate(tests))
    return np.array(accs), digest_state(oa.state_dict()), oa


@pytest.mark.parametrize("name", ["er_traj", "scr_traj"])
def test_free_running_trajectory_inside_the_oracles_own_spread(cuda, name):
    """Whole free-running runs (3 tasks, ~30 SGD steps at lr 0.1, evaluate after every task), 3 seeds.  Everything the host RNGs
    drive (buffer labels, counters) must equal the oracle's exactly.  The weights follow a trajectory that amplifies fp32
    round-off: the oracle itself, run with 1 / 8 / 16 intra-op threads or from initial weights perturbed by ONE ulp, ends in
    states that differ from each other; the HIP run's distance from the reference oracle run must not exceed 5x the largest of
    those self-distances (state digest), and its end accuracy must lie within [the largest accuracy spread the oracle shows for one
    seed] + 3 binomial standard errors of the oracle's median.  This is the statistical statement that replaces a bit-wise
    trajectory comparison: ER at lr 0.1 on ten-sample batches is chaotic (a 1-ulp change of the initial weights moves the end
    accuracy of one seed by tens of points), SCR with the NCM classifier is not."""
    from ocl_amd.data import setup_test_loader
    base = dict(er_traj=dict(agent="ER", retrieve="random", update="random", data="cifar10", mem_size=60, eps_mem_batch=10,
                             tasks=[[0, 1], [2, 3], [4, 5]], n_train=50, n_test=50),
                scr_traj=dict(agent="SCR", retrieve="random", update="random", data="cifar100", mem_size=60, eps_mem_batch=20,
                              tasks=[[3, 17], [40, 41], [7, 9]], n_train=40, n_test=50, temp=0.07, head="mlp"))[name]
    default_threads = torch.get_num_threads()
    rows = []
    n_test_total = base["n_test"] * sum(len(t) for t in base["tasks"])
    try:
        for seed in (101, 102, 103):
            cfg = dict(base, seed=seed)
            acc_ref, dig_ref, oa_ref = _oracle_run(cfg, threads=1)
            variants = [_oracle_run(cfg, threads=8), _oracle_run(cfg, threads=16), _oracle_run(cfg, threads=1, ulp=1),
                        _oracle_run(cfg, threads=1, ulp=-1)]
            torch.set_num_threads(default_threads)
            scale = np.abs(dig_ref).max()
            self_dist = max(np.abs(v[1] - dig_ref).max() / scale for v in variants)
            end_accs = [acc_ref[-1].mean()] + [v[0][-1].mean() for v in variants]
            params, model, agent = build_agent(cfg)
            tasks, tests = make_stream(cfg)
            loaders = setup_test_loader(tests, params)
            accs = []
            for x, y in tasks:
                agent.train_learner(x, y)
                accs.append(agent.evaluate(loaders))
            acc_h = np.array(accs)
            assert np.array_equal(agent.buffer.buffer_label.cpu().numpy(), oa_ref.buf.label.numpy())
            assert [agent.buffer.current_index, agent.buffer.n_seen_so_far] == [oa_ref.buf.current_index, oa_ref.buf.n_seen_so_far]
            hip_dist = np.abs(digest_state(model.state_dict()) - dig_ref).max() / scale
            print("%s seed %d: oracle self-distance %.3e (threads 8 / 16, +-1-ulp init), HIP distance %.3e; end acc HIP %.4f, oracle %s"
                  % (name, seed, self_dist, hip_dist, acc_h[-1].mean(), np.round(end_accs, 4).tolist()))
            assert self_dist > 0, "the oracle did not move under a 1-ulp perturbation: the run is too short to say anything"
            assert hip_dist <= 5 * self_dist + 1e-6, (hip_dist, self_dist)
            rows.append((acc_h[-1].mean(), end_accs, hip_dist / self_dist))
    finally:
        torch.set_num_threads(default_threads)
    # accuracy: the regime's own noise = the largest spread the oracle shows for one seed under rounding-level perturbations, plus
    # three binomial standard errors of the test set
    pooled = max(max(e) - min(e) for _, e, _ in rows)
    for acc_hip, end_accs, _ in rows:
        p = float(np.clip(np.median(end_accs), 0.02, 0.98))
        tol = pooled + 3 * np.sqrt(p * (1 - p) / n_test_total)
        assert abs(acc_hip - np.median(end_accs)) <= tol, (acc_hip, end_accs, tol)
    print(name, "HIP distance / oracle self-distance per seed:", np.round([r[2] for r in rows], 2), "| pooled oracle accuracy spread %.4f" % pooled)


# ---------------------------------------------------------------------------------------------------------------------
# GSS-Greedy
# ---------------------------------------------------------------------------------------------------------------------

def test_cosim_gss(cuda):
    """ER + GSS-Greedy update (utils/buffer/gss_greedy_update.py) from identical state each step (weights, memory, slot scores):
    eval-mode gradient similarities within 1e-3 (fill phase: every sample's score; full memory: the batch score whose sign is
    the replacement decision), identical multinomial draws (slots, swap outcomes) and buffers, RNG in step; the run takes the
    replacement branch at least once."""
    cfg = STEP_CASES["er_gss"]
    n_fill = n_repl = n_full = 0
    tasks, _ = make_stream(cfg)
    stream = (np.concatenate([tasks[0][0], tasks[1][0]]), np.concatenate([tasks[0][1], tasks[1][1]]))   # 30 of class 0, then class 1
    gen = cosim(cfg, 6, cuda, x_stream=stream, sync_extra=lambda agent, oa: agent.buffer.update_method.buffer_score.copy_(oa.gss.score))
    for it, ev, ol, chk in gen:
        assert chk["rng_equal"], "host RNG streams diverged at iteration %d" % it
        g, og = [e for t, e in ev if t == "gss"][0], ol["gss"]
        if "fill" in og:
            assert g["fill"] == og["fill"] and np.abs(g["item_sim"] - og["item_sim"]).max() < 1e-3
            n_fill += 1
        else:
            n_full += 1
            assert abs(g["batch_sim"] - og["batch_sim"]) < 1e-3, (g["batch_sim"], og["batch_sim"])
            assert ("index" in g) == ("index" in og)
            if "index" in og:
                assert np.array_equal(g["index"], og["index"]) and np.array_equal(g["sub"], og["sub"])
                assert np.abs(g["item_sim"] - og["item_sim"]).max() < 1e-3
                n_repl += 1
        a, oa = chk["agent"], chk["oa"]
        assert _buffers_equal_gss(a, oa)
        assert np.abs(a.buffer.update_method.buffer_score.numpy() - oa.gss.score.numpy()).max() < 1e-3
        assert chk["model"].training
    print("GSS: %d fill steps, %d full-memory steps, %d replacements" % (n_fill, n_full, n_repl))
    assert n_fill >= 2 and n_repl >= 1


def _buffers_equal_gss(agent, oa):
    return (np.array_equal(agent.buffer.buffer_label.cpu().numpy(), oa.buf.label.numpy()) and torch.equal(agent.buffer.buffer_img.cpu(), oa.buf.img)
            and agent.buffer.current_index == oa.buf.current_index)


def test_eval_mode_gradient_vs_oracle(cuda):
    """The engine capability GSS rests on: gradients of an eval-mode forward (BatchNorm = affine map of its running statistics,
    OCL_FWD_FROZEN_BN) w.r.t. every parameter, batch sizes 1 and 10, against torch autograd on the oracle with the ReLU
    pattern teacher-forced; running statistics untouched; and ocl_cosine_max against torch's cosine similarity."""
    from test_gpu_net import build, engine_masks
    from ocl_amd.loss import cross_entropy_mean
    from ocl_amd import ops
    m, sd = build("ER", "cifar10", cuda=cuda)
    rng = np.random.default_rng(5)
    for k in sd:
        if k.endswith("running_mean"):
            sd[k] = torch.from_numpy(rng.standard_normal(sd[k].shape).astype(np.float32) * 0.1)
        if k.endswith("running_var"):
            sd[k] = torch.from_numpy((0.5 + rng.random(sd[k].shape)).astype(np.float32))
    m.load_state_dict(sd)
    grads = []
    for n in (1, 10):
        x = rng.random((n, 3, 32, 32)).astype(np.float32)
        y = rng.integers(0, 10, n).astype(np.int64)
        m.eval()
        m.mark_grads_zero()
        out = m.forward(torch.from_numpy(x).to(cuda))
        assert out.requires_grad
        loss = cross_entropy_mean(out, torch.from_numpy(y).to(cuda))
        probe = O.OracleNet(O.clone_state(sd, requires_grad=False), head=None, training=False)
        probe.pre_act = {}
        with torch.no_grad():
            probe.forward(torch.from_numpy(x))
        masks = engine_masks(m, {k: tuple(v.shape) for k, v in probe.pre_act.items()}, "")
        loss.backward()
        st = O.clone_state(sd)
        net = O.OracleNet(st, head=None, training=False)
        net.mask_override = masks
        lref = torch.nn.functional.cross_entropy(net.forward(torch.from_numpy(x)), torch.from_numpy(y))
        lref.backward()
        assert abs(float(loss) - float(lref.detach())) < 1e-4
        worst = 0.0
        for k, p in m.named_parameters():
            gref = st[k].grad
            worst = max(worst, float(np.abs(p.grad.cpu().numpy() - gref.numpy()).max() / (1e-12 + float(gref.abs().max()))))
        print("eval-mode gradient, batch %d: worst relative error %.2e" % (n, worst))
        assert worst < 2e-4
        grads.append(m.flat_grads().clone())
        sd2 = m.state_dict()
        for k in sd:
            if "running" in k or "num_batches" in k:
                assert torch.equal(sd2[k].cpu(), sd[k]), "eval-mode tape touched %s" % k
    stack = torch.stack([grads[0], grads[1], -grads[0] + 0.5 * grads[1]])
    got = float(ops.cosine_max(stack, grads[1]).cpu())
    ref32 = float(max(O.cosine_similarity(stack.cpu(), grads[1].cpu().unsqueeze(0))))
    ref64 = float(max(O.cosine_similarity(stack.cpu().double(), grads[1].cpu().double().unsqueeze(0))))
    # the kernel accumulates the 10^6-term dot products in fp64; torch.mm / norm in fp32 lose ~2e-5 on a vector against itself
    assert abs(got - ref64) < 1e-6 and abs(got - ref32) < 1e-4 and abs(got - 1.0) < 1e-6


# ---------------------------------------------------------------------------------------------------------------------
# the sharded driver
# ---------------------------------------------------------------------------------------------------------------------

def test_single_run_driver_on_two_task_stream(cuda):
    """ocl_amd.run.single_run (experiment/run.py:36-56 for one run) on the scr_c100 stream: same seed as the golden run, so buffer
    labels / counters must match the REAL reference's recorded run, accuracies are a [T, T] array in [0, 1]."""
    from ocl_amd.run import single_run
    g = gold("steps")
    cfg = STEP_CASES["scr_c100"]
    params = make_params(cfg)
    tasks, tests = make_stream(cfg)
    import ocl_amd.agents.scr as scr_mod
    orig = scr_mod.ScrAugment.__call__
    scr_mod.ScrAugment.__call__ = lambda self, x: x      # identity augmentation, as in the golden run (kornia is unpinned)
    try:
        acc, t_train, n_img, agent = single_run(params, tasks, tests, cfg["seed"])
    finally:
        scr_mod.ScrAugment.__call__ = orig
    assert acc.shape == (2, 2) and (acc >= 0).all() and (acc <= 1).all() and n_img == 100 and t_train > 0
    assert np.array_equal(agent.buffer.buffer_label.cpu().numpy(), g["scr_c100_t1_buf_label"])
    assert [agent.buffer.current_index, agent.buffer.n_seen_so_far] == g["scr_c100_t1_counters"].tolist()


def test_sharded_runs_two_processes(cuda):
    """run.sharded_runs under torch.distributed.run with 2 ranks (RCCL when two GPUs are visible, otherwise both ranks share
    GPU 0 over gloo: the all_gather of the [T, T] accuracy arrays and the summary metrics are what is exercised)."""
    n_gpu = torch.cuda.device_count()
    env = dict(os.environ, OCL_SHARD_BACKEND="nccl" if n_gpu >= 2 else "gloo", PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(ROOT, "tests", "shard_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "SHARDED_OK world=2" in r.stdout, r.stdout[-2000:]


def test_rccl_group_of_one_runs_the_exchange(cuda):
    """The RCCL branch on a one-GPU box: a world-size-1 "nccl" group, gather_runs / max_over_ranks / sum_over_ranks / gather_scalars with
    device payloads, then run.sharded_runs over it (tests/nccl_worker.py)."""
    env = dict(os.environ, PYTHONPATH=ROOT, OCL_TEST_PORT="29547")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "nccl_worker.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "RCCL_OK world=1 backend=nccl" in r.stdout, r.stdout[-2000:]


def test_bench_gpus_2_launches_its_own_ranks(cuda):
    """`python bench.py --gpus 2` with no launcher around it: bench.py re-executes itself under torch.distributed.run with two ranks
    (one per GPU over RCCL; on a one-GPU box both ranks share GPU 0 and the scalars travel over gloo), each rank runs its own
    independent stream (experiment/run.py:34 sharded), and rank 0 prints ONE line with n_gpus = 2 and the per-rank rates."""
    import json
    env = dict(os.environ, PYTHONPATH=ROOT, PYTHONDONTWRITEBYTECODE="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--workload", "er", "--no-roofline"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and len(d["per_rank_images_per_s"]) == 2
    assert d["value"] > 0 and abs(d["value"] - 2 * 20 * 10 / (d["ms_per_step"] * 20 * 1e-3)) < 1e-6 * d["value"]   # whole-job rate = all ranks' images / max-over-ranks time
    assert "cpu_baseline" not in d and "accuracy" not in d                                                       # single-GPU record only


# ---------------------------------------------------------------------------------------------------------------------
# run-to-run reproducibility
# ---------------------------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("workload", ["scr", "er"])
def test_training_steps_are_bit_reproducible(cuda, workload):
    """Deterministic mode (ocl_set_deterministic): the BatchNorm batch sums (forward statistics, backward reductions) are accumulated as
    fixed-point integers (csrc/conv.h StatCell), the weight-gradient slabs are reduced in a fixed order: nothing in a step depends on
    the order in which workgroups finish, whatever
    the two engine streams and the data stream do.  Two fresh agents, same seeds, same stream: after 12 steps at BASELINE size (SCR:
    110 + 110 views through the two-stream backward; ER: the merged 20-image pass) every weight, BatchNorm buffer and memory row is
    BIT-IDENTICAL.  (The reference's CPU path has this property at a fixed thread count.)"""
    import random
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)

    def run():
        params, model, agent, hw, ncls = bench.build_agent(workload, 3, cuda)
        x, y = bench.synth_u8(12 * params.batch, hw, ncls, 17)
        np.random.seed(5); random.seed(5); torch.manual_seed(5)
        agent.train_learner(torch.from_numpy(x).to(cuda), y)
        torch.cuda.synchronize()
        sd = {k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}
        return sd, agent.buffer.buffer_img.detach().cpu().numpy().copy(), agent.buffer.buffer_label.detach().cpu().numpy().copy()

    from ocl_amd import ops
    ops.set_deterministic(True)
    try:
        a, b = run(), run()
    finally:
        ops.set_deterministic(False)
    assert a[0].keys() == b[0].keys()
    diff = [k for k in a[0] if not np.array_equal(a[0][k], b[0][k])]
    assert not diff, "tensors differ between two identical runs: %s" % diff[:5]
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
