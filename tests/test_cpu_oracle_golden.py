"""CPU: the oracle restatement (oracle/ocl_oracle.py) against the golden vectors generated from the REAL reference
(oracle/make_golden.py).  Integer / index results exact; fp32 within the stated tolerances."""
import numpy as np
import pytest
import torch

from conftest import gold
from oracle import ocl_oracle as O
from oracle.synth import STEP_CASES, make_stream, seed_all, digest_state


def test_knn_sv_matches_reference_bit_exact():
    g = gold("knn_sv")
    for ci in range(int(g["n_cases"])):
        sv, order = O.knn_sv(g["c%d_ef" % ci], g["c%d_ey" % ci], g["c%d_cf" % ci], g["c%d_cy" % ci], int(g["c%d_k" % ci]))
        assert np.array_equal(sv, g["c%d_sv" % ci]), "case %d" % ci
        assert np.array_equal(order, g["c%d_order" % ci])
    sv, _ = O.knn_sv(g["tie_ef"], g["tie_ey"], g["tie_cf"], g["tie_cy"], int(g["tie_k"]))
    assert np.array_equal(sv, g["tie_sv"])   # exact distance ties between same-label candidates: order-independent


@pytest.mark.parametrize("n,k", [(6, 2), (7, 3), (5, 5), (6, 6)])
def test_knn_sv_equals_bruteforce_shapley(n, k):
    """Independent known-answer for N >= K (incl. N == K): tolerance 1e-6 (fp32 recursion).  For N < K the reference's
    closed form (last factor 1/N, aser_utils.py:46-49) is NOT the Shapley value of the 1/K-normalised utility (the
    farthest point would get 1/K); the oracle mirrors the reference there and is pinned by golden case 3 (N=4, K=7)."""
    g = gold("knn_sv")
    dist, match, phi = g["bf_%d_%d_dist" % (n, k)], g["bf_%d_%d_match" % (n, k)], g["bf_%d_%d_phi" % (n, k)]
    assert np.allclose(O.knn_shapley_bruteforce(dist, match, k), phi)
    order = np.argsort(dist, kind="stable")[None]
    sv, _ = O.knn_sv(np.zeros((1, 1), np.float32), np.array([1]), np.zeros((n, 1), np.float32), match.astype(np.int64), k, order=order)
    assert np.abs(sv[0] - phi).max() < 1e-6


def test_supcon_matches_reference():
    g = gold("supcon")
    for ci in range(int(g["n_cases"])):
        f = torch.from_numpy(g["c%d_f" % ci]).requires_grad_(True)
        loss = O.supcon_loss(f, torch.from_numpy(g["c%d_y" % ci]), float(g["c%d_t" % ci]))
        loss.backward()
        assert abs(float(loss.detach()) - float(g["c%d_loss" % ci])) < 1e-6
        assert np.abs(f.grad.numpy() - g["c%d_grad" % ci]).max() < 1e-6


def test_ce_tricks_match_reference():
    """Labels trick / separated softmax (agents/base.py:96-108) vs the reference's ContinualLearner.criterion + autograd."""
    g = gold("ce_tricks")
    for ci in range(int(g["n_cases"])):
        lt = torch.from_numpy(g["c%d_logits" % ci]).requires_grad_(True)
        y = torch.from_numpy(g["c%d_y" % ci])
        old, new = g["c%d_old" % ci].tolist(), g["c%d_new" % ci].tolist()
        if str(g["c%d_kind" % ci]) == "labels":
            loss = O.ce_labels_trick(lt, y)
        else:
            loss = O.ce_separated_softmax(lt, y, old, new, {l: i for i, l in enumerate(old + new)})
        loss.backward()
        assert abs(float(loss.detach()) - float(g["c%d_loss" % ci])) < 1e-6
        assert np.abs(lt.grad.numpy() - g["c%d_grad" % ci]).max() < 1e-7


def test_kd_loss_matches_reference():
    g = gold("kd")
    for ci in range(int(g["n_cases"])):
        st = torch.from_numpy(g["c%d_s" % ci]).requires_grad_(True)
        loss = O.loss_fn_kd(st, torch.from_numpy(g["c%d_t" % ci]), float(g["c%d_T" % ci]))
        loss.backward()
        assert abs(float(loss.detach()) - float(g["c%d_loss" % ci])) < 1e-6
        assert np.abs(st.grad.numpy() - g["c%d_grad" % ci]).max() < 1e-7


def test_reservoir_and_random_retrieve_sequences_exact():
    g = gold("buffer_ops")
    for ci in range(int(g["n_cases"])):
        mem, bs, steps, seed = [int(v) for v in g["c%d_cfg" % ci]]
        torch.manual_seed(seed)
        np.random.seed(seed)
        rng = np.random.default_rng(seed)
        xs = [torch.from_numpy(rng.standard_normal((bs, 3, 2, 2)).astype(np.float32)) for _ in range(steps)]
        ys = [torch.from_numpy(rng.integers(0, 10, bs).astype(np.int64)) for _ in range(steps)]
        buf = O.OracleBuffer(mem, (3, 2, 2))
        slots, retr = [], []
        for s in range(steps):
            retr.append(O.random_retrieve_indices(buf, 7))
            slots.append(np.array(O.reservoir_update(buf, xs[s], ys[s]), dtype=np.int64))
        assert np.array_equal(np.concatenate(slots), g["c%d_slots" % ci])
        assert np.array_equal([len(s) for s in slots], g["c%d_slot_counts" % ci])
        assert np.array_equal(np.concatenate(retr), g["c%d_retr" % ci])
        assert np.array_equal(buf.label.numpy(), g["c%d_final_label" % ci])
        assert [buf.current_index, buf.n_seen_so_far] == g["c%d_final_n" % ci].tolist()


def test_mem_match_retrieve_matches_reference_plugin():
    """utils/buffer/mem_match.py through the oracle's restatement against the reference plugin's recorded picks (ids of the samples
    it returned), the memory it leaves behind and the position of numpy's / Python's global generators."""
    import random
    g = gold("mem_match")
    for ci in range(int(g["n_cases"])):
        mem, bs, steps, nret, warmup, ncls, seed = [int(v) for v in g["c%d_cfg" % ci]]
        torch.manual_seed(seed)
        np.random.seed(seed)
        random.seed(seed)
        buf, tr = O.OracleBuffer(mem, (3, 32, 32)), O.ClassTracker(10)
        pos = {"cand_id": 0, "match_id": 0}
        for s in range(steps):
            cand, part = O.mem_match_indices(buf, tr, nret, warmup)
            for k, idx in (("cand_id", cand), ("match_id", part)):
                n = int(g["c%d_%s_counts" % (ci, k)][s])
                exp = g["c%d_%s" % (ci, k)][pos[k]:pos[k] + n]
                pos[k] += n
                assert np.array_equal(buf.img[idx][:, 0, 0, 0].numpy() if len(idx) else np.zeros(0, np.float32), exp), (ci, s, k)
            x = torch.zeros(bs, 3, 32, 32)
            x[:, 0, 0, 0] = torch.from_numpy(g["c%d_ids" % ci][s])
            O.reservoir_update(buf, x, torch.from_numpy(g["c%d_ys" % ci][s]), tracker=tr)
        assert np.array_equal(buf.img[:, 0, 0, 0].numpy(), g["c%d_final_ids" % ci]) and np.array_equal(buf.label.numpy(), g["c%d_final_label" % ci])
        assert np.array_equal(np.random.get_state()[1][:8].astype(np.int64), g["c%d_np_state" % ci])
        assert random.random() == float(g["c%d_py_draw" % ci])


@pytest.mark.parametrize("name,agent,data,hw,n,head", [("rr18_c100", "ER", "cifar100", 32, 6, None), ("scr_mlp", "SCR", "cifar100", 32, 6, "mlp"),
                                                       ("rr18_mini", "ER", "mini_imagenet", 84, 3, None)])
def test_resnet_forward_backward_matches_reference(name, agent, data, hw, n, head):
    g = gold("resnet")
    torch.manual_seed(11)
    s = O.init_state(agent, data, head or "mlp")
    init = np.array([[float(v.double().sum()), float(v.double().norm())] for v in s.values() if v.is_floating_point()])
    assert np.allclose(init, g[name + "_init_digest"], rtol=0, atol=0), "seeded initialisation differs from the reference's"
    rng = np.random.default_rng(5)
    x = torch.from_numpy(rng.random((n, 3, hw, hw)).astype(np.float32))
    y = torch.from_numpy(rng.integers(0, 100, n).astype(np.int64))
    net = O.OracleNet(s, head=head, training=True)
    o = net.forward(x)
    loss = torch.nn.functional.cross_entropy(o, y) if head is None else (o * torch.linspace(-1, 1, o.numel()).view_as(o)).sum()
    loss.backward()
    assert np.abs(o.detach().numpy() - g[name + "_out"]).max() < 1e-5
    names = [k for k in s if s[k].requires_grad]
    dig = np.array([[float(s[k].grad.double().sum()), float(s[k].grad.double().norm())] if s[k].grad is not None else [0.0, 0.0]
                    for k in names])
    gd = g[name + "_grad_digest"]
    assert np.abs(dig - gd).max() <= 1e-4 * (1 + np.abs(gd).max())
    for k in g[name + "_picked"]:
        gg = g[name + "_g_" + str(k)]
        assert np.abs(s[str(k)].grad.numpy() - gg).max() <= 1e-5 * (1 + np.abs(gg).max()), k
    net.training = False
    with torch.no_grad():
        fe = net.features(x)
    assert np.abs(fe.numpy() - g[name + "_feat_eval"]).max() < 1e-5


@pytest.mark.parametrize("name", [n for n, c in STEP_CASES.items() if c.get("golden", True)])
def test_step_cases_match_reference_runs(name):
    """Whole tasks (train_learner + evaluate) of the oracle agent vs the reference agent's recorded run: buffer labels,
    counters and accuracies exact; weights / BN buffers / buffer images within 1e-6 relative (same ATen kernels)."""
    g = gold("steps")
    cfg = STEP_CASES[name]
    torch.set_num_threads(1)
    seed_all(cfg["seed"])
    ag = O.OracleAgent(cfg)
    tasks, tests = make_stream(cfg)
    for t, (x, y) in enumerate(tasks):
        ag.train_learner(x, y)
        acc = ag.evaluate(tests)
        pre = "%s_t%d_" % (name, t)
        assert np.array_equal(ag.buf.label.numpy(), g[pre + "buf_label"])
        assert [ag.buf.current_index, ag.buf.n_seen_so_far] == g[pre + "counters"].tolist()
        assert np.array_equal(acc, g[pre + "acc"])
        assert np.abs(ag.buf.img.double().sum(dim=(1, 2, 3)).numpy() - g[pre + "buf_rowsum"]).max() < 1e-6
        ds, gs = digest_state(ag.state_dict()), g[pre + "state"]
        assert np.abs(ds - gs).max() <= 1e-6 * (1 + np.abs(gs).max())
