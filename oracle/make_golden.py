"""TEST INFRASTRUCTURE ONLY — generates tests/golden/*.npz by running the REAL reference (/root/reference, imported
read-only under oracle/stubs) on seeded inputs, and checks oracle/ocl_oracle.py against it while doing so.

Run in the build container only (the reference does not exist on the GPU box):
    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py
The fixtures are committed; tests read them and never import the reference.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import ref_import as R  # noqa: E402
from oracle import ocl_oracle as O  # noqa: E402
from oracle.synth import make_stream, seed_all, digest_state, STEP_CASES, case_params  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def _ref_modules():
    R.activate()
    import utils.buffer.aser_utils as aser_utils
    import utils.loss as ref_loss
    import utils.buffer.reservoir_update as ref_res
    import utils.buffer.buffer_utils as ref_bu
    return aser_utils, ref_loss, ref_res, ref_bu


class _IdentityFeatures(torch.nn.Module):
    """compute_knn_sv(model, ...) calls model.features on the raw inputs: feed features directly."""

    def features(self, x):
        return x.reshape(x.shape[0], -1)


def gen_knn_sv():
    aser_utils, _, _, _ = _ref_modules()
    out = {}
    cases = [(10, 100, 160, 3, 100), (110, 160, 160, 3, 100), (5, 7, 8, 3, 3), (3, 4, 6, 7, 2), (2, 1, 4, 3, 2), (6, 33, 16, 1, 4),
             (4, 64, 640, 5, 10)]
    for ci, (ne, nc, d, k, ncls) in enumerate(cases):
        rng = np.random.default_rng(100 + ci)
        ef = rng.standard_normal((ne, d)).astype(np.float32)
        cf = rng.standard_normal((nc, d)).astype(np.float32)
        ey = rng.integers(0, ncls, ne).astype(np.int64)
        cy = rng.integers(0, ncls, nc).astype(np.int64)
        sv = aser_utils.compute_knn_sv(_IdentityFeatures(), torch.from_numpy(ef), torch.from_numpy(ey), torch.from_numpy(cf),
                                       torch.from_numpy(cy), k, device="cpu").numpy()
        mine, order = O.knn_sv(ef, ey, cf, cy, k)
        assert np.array_equal(sv, mine), "oracle knn_sv differs from the reference (case %d): max %g" % (ci, np.abs(sv - mine).max())
        out["c%d_ef" % ci], out["c%d_cf" % ci], out["c%d_ey" % ci], out["c%d_cy" % ci] = ef, cf, ey, cy
        out["c%d_k" % ci] = np.int64(k)
        out["c%d_sv" % ci] = sv
        out["c%d_order" % ci] = order
    # tie case: duplicated candidates -> exact distance ties; store the reference result for the tie-aware check
    rng = np.random.default_rng(7)
    ef = rng.standard_normal((3, 8)).astype(np.float32)
    base = rng.standard_normal((5, 8)).astype(np.float32)
    cf = np.concatenate([base, base[:3]], 0)
    ey = np.array([0, 1, 0], dtype=np.int64)
    cy = np.array([0, 1, 0, 1, 0, 0, 1, 0], dtype=np.int64)   # tied pairs share the label -> SV is order-independent
    sv = aser_utils.compute_knn_sv(_IdentityFeatures(), torch.from_numpy(ef), torch.from_numpy(ey), torch.from_numpy(cf),
                                   torch.from_numpy(cy), 3, device="cpu").numpy()
    mine, _ = O.knn_sv(ef, ey, cf, cy, 3)
    assert np.array_equal(sv, mine)
    out.update(tie_ef=ef, tie_cf=cf, tie_ey=ey, tie_cy=cy, tie_k=np.int64(3), tie_sv=sv)
    out["n_cases"] = np.int64(len(cases))
    # brute-force Shapley known-answers (independent of the reference)
    for (n, k) in [(6, 2), (7, 3), (5, 5), (6, 6)]:
        rng = np.random.default_rng(n * 10 + k)
        dist = rng.random(n)
        match = rng.integers(0, 2, n).astype(np.float64)
        phi = O.knn_shapley_bruteforce(dist, match, k)
        out["bf_%d_%d_dist" % (n, k)], out["bf_%d_%d_match" % (n, k)], out["bf_%d_%d_phi" % (n, k)] = dist, match, phi
    np.savez_compressed(os.path.join(GOLD, "knn_sv.npz"), **out)
    print("knn_sv: %d cases ok" % len(cases))


def gen_supcon():
    _, ref_loss, _, _ = _ref_modules()
    out = {}
    cases = [(110, 128, 10, 0.07), (20, 128, 3, 0.1), (4, 8, 2, 0.5), (2, 4, 1, 0.07)]
    for ci, (b, d, ncls, t) in enumerate(cases):
        rng = np.random.default_rng(200 + ci)
        f = rng.standard_normal((b, 2, d)).astype(np.float32)
        f /= np.linalg.norm(f, axis=2, keepdims=True)
        y = rng.integers(0, ncls, b).astype(np.int64)
        ft = torch.from_numpy(f).requires_grad_(True)
        loss = ref_loss.SupConLoss(temperature=t)(ft, torch.from_numpy(y))
        loss.backward()
        ft2 = torch.from_numpy(f).requires_grad_(True)
        l2 = O.supcon_loss(ft2, torch.from_numpy(y), t)
        l2.backward()
        assert abs(float(loss) - float(l2)) < 1e-6 and (ft.grad - ft2.grad).abs().max() < 1e-6
        out["c%d_f" % ci], out["c%d_y" % ci], out["c%d_t" % ci] = f, y, np.float64(t)
        out["c%d_loss" % ci], out["c%d_grad" % ci] = np.float64(float(loss)), ft.grad.numpy()
    out["n_cases"] = np.int64(len(cases))
    np.savez_compressed(os.path.join(GOLD, "supcon.npz"), **out)
    print("supcon ok")


def gen_ce_tricks():
    """agents/base.py:93-108 through the reference's own ContinualLearner.criterion (unbound, on a stand-in `self`)."""
    from types import SimpleNamespace
    R.activate()
    import agents.base as ref_base
    out = {}
    cases = [("labels", 10, 100, list(range(20, 30)), []), ("labels", 20, 10, [3, 7], []), ("labels", 1, 100, [5], []),
             ("sep", 10, 100, list(range(10)), list(range(10, 20))), ("sep", 20, 10, [0, 1, 2, 3], [8, 9]),
             ("sep", 10, 100, [], [4, 5, 6])]
    for ci, (kind, n, c, cls_a, cls_b) in enumerate(cases):
        rng = np.random.default_rng(700 + ci)
        logits = (3.0 * rng.standard_normal((n, c))).astype(np.float32)
        pool = np.array(cls_a + cls_b if kind == "sep" else cls_a)
        y = pool[rng.integers(0, len(pool), n)].astype(np.int64)
        trick = {k: False for k in ('labels_trick', 'kd_trick', 'separated_softmax', 'review_trick', 'ncm_trick', 'kd_trick_star')}
        trick['labels_trick' if kind == "labels" else 'separated_softmax'] = True
        if kind == "sep":
            old, new = list(cls_a), list(cls_b)
            inv = {l: i for i, l in enumerate(old + new)}
        else:
            old, new, inv = [], [], {}
        fake = SimpleNamespace(params=SimpleNamespace(trick=trick, agent="ER", temp=0.07), old_labels=old, new_labels=new, lbl_inv_map=inv)
        lt = torch.from_numpy(logits).requires_grad_(True)
        loss = ref_base.ContinualLearner.criterion(fake, lt, torch.from_numpy(y))
        loss.backward()
        lt2 = torch.from_numpy(logits).requires_grad_(True)
        l2 = O.ce_labels_trick(lt2, torch.from_numpy(y)) if kind == "labels" else O.ce_separated_softmax(lt2, torch.from_numpy(y), old, new, inv)
        l2.backward()
        assert abs(float(loss) - float(l2)) < 1e-6 and (lt.grad - lt2.grad).abs().max() < 1e-7
        out["c%d_kind" % ci] = np.array(kind)
        out["c%d_logits" % ci], out["c%d_y" % ci] = logits, y
        out["c%d_old" % ci], out["c%d_new" % ci] = np.array(old, dtype=np.int64), np.array(new, dtype=np.int64)
        out["c%d_loss" % ci], out["c%d_grad" % ci] = np.float64(float(loss)), lt.grad.numpy()
    out["n_cases"] = np.int64(len(cases))
    np.savez_compressed(os.path.join(GOLD, "ce_tricks.npz"), **out)
    print("ce tricks ok")


def gen_kd():
    """utils/kd_manager.py:6-11 through the reference's own loss_fn_kd + autograd."""
    R.activate()
    import utils.kd_manager as ref_kd
    out = {}
    cases = [(10, 100, 2.0), (20, 10, 2.0), (1, 100, 2.0), (10, 10, 4.0)]
    for ci, (n, c, T) in enumerate(cases):
        rng = np.random.default_rng(800 + ci)
        s_ = (3.0 * rng.standard_normal((n, c))).astype(np.float32)
        t_ = (3.0 * rng.standard_normal((n, c))).astype(np.float32)
        st = torch.from_numpy(s_).requires_grad_(True)
        loss = ref_kd.loss_fn_kd(st, torch.from_numpy(t_), T)
        loss.backward()
        st2 = torch.from_numpy(s_).requires_grad_(True)
        l2 = O.loss_fn_kd(st2, torch.from_numpy(t_), T)
        l2.backward()
        assert abs(float(loss) - float(l2)) < 1e-6 and (st.grad - st2.grad).abs().max() < 1e-7
        out["c%d_s" % ci], out["c%d_t" % ci], out["c%d_T" % ci] = s_, t_, np.float64(T)
        out["c%d_loss" % ci], out["c%d_grad" % ci] = np.float64(float(loss)), st.grad.numpy()
    out["n_cases"] = np.int64(len(cases))
    np.savez_compressed(os.path.join(GOLD, "kd.npz"), **out)
    print("kd ok")


def gen_buffer_ops():
    """reservoir slot sequences + random_retrieve index sequences for fixed seeds."""
    _, _, ref_res, ref_bu = _ref_modules()
    from types import SimpleNamespace
    out = {}
    p = SimpleNamespace(buffer_tracker=False)

    class _B(object):
        pass

    def mk(mem):
        b = _B()
        b.buffer_img = torch.zeros(mem, 3, 2, 2)
        b.buffer_label = torch.zeros(mem, dtype=torch.long)
        b.current_index = 0
        b.n_seen_so_far = 0
        b.params = p
        return b

    for ci, (mem, bs, steps, seed) in enumerate([(50, 10, 40, 0), (23, 10, 30, 1), (8, 3, 25, 2)]):
        torch.manual_seed(seed)
        np.random.seed(seed)
        b = mk(mem)
        ob = O.OracleBuffer(mem, (3, 2, 2))
        upd = ref_res.Reservoir_update(p)
        slots_all, retr_all = [], []
        rng = np.random.default_rng(seed)
        xs = [torch.from_numpy(rng.standard_normal((bs, 3, 2, 2)).astype(np.float32)) for _ in range(steps)]
        ys = [torch.from_numpy(rng.integers(0, 10, bs).astype(np.int64)) for _ in range(steps)]
        for s in range(steps):
            _, _, idx = ref_bu.random_retrieve(b, 7, return_indices=True)
            retr_all.append(idx.numpy())
            slots_all.append(np.array(upd.update(b, xs[s], ys[s]), dtype=np.int64))
        # oracle replay with identical seeds
        torch.manual_seed(seed)
        np.random.seed(seed)
        for s in range(steps):
            idx = O.random_retrieve_indices(ob, 7)
            assert np.array_equal(idx, retr_all[s])
            sl = np.array(O.reservoir_update(ob, xs[s], ys[s]), dtype=np.int64)
            assert np.array_equal(sl, slots_all[s]), (s, sl, slots_all[s])
        assert torch.equal(ob.label, b.buffer_label) and torch.equal(ob.img, b.buffer_img)
        out["c%d_cfg" % ci] = np.array([mem, bs, steps, seed], dtype=np.int64)
        out["c%d_slots" % ci] = np.concatenate([np.array([-1 - len(s)]) if False else s for s in slots_all]) if slots_all else np.zeros(0)
        out["c%d_slot_counts" % ci] = np.array([len(s) for s in slots_all], dtype=np.int64)
        out["c%d_retr" % ci] = np.concatenate(retr_all) if retr_all else np.zeros(0)
        out["c%d_retr_counts" % ci] = np.array([len(r) for r in retr_all], dtype=np.int64)
        out["c%d_final_label" % ci] = b.buffer_label.numpy()
        out["c%d_final_n" % ci] = np.array([b.current_index, b.n_seen_so_far], dtype=np.int64)
    out["n_cases"] = np.int64(3)
    np.savez_compressed(os.path.join(GOLD, "buffer_ops.npz"), **out)
    print("buffer ops ok")


def gen_mem_match():
    """utils/buffer/mem_match.py:5-21 (`retrieve_methods['mem_match']`) through the reference's own plugin: random candidates, then
    their label-matched partners from the BufferClassTracker, redrawn until every candidate has a partner outside the draw.  The
    memory is filled by the reference's reservoir update (tracker kept by it); every image carries a unique id in its first
    element, so the recorded ids name the slots the plugin picked.  numpy's and Python's global generators drive the draws."""
    R.activate()
    import random as pyrandom
    from types import SimpleNamespace
    import utils.buffer.mem_match as ref_mm
    import utils.buffer.reservoir_update as ref_res
    import utils.buffer.buffer_utils as ref_bu
    out = {}
    # (the reference's loop never ends when the memory cannot hold a partner for every candidate: warm-ups chosen so that it can)
    cases = [(40, 10, 14, 6, 4, 4, 0), (24, 5, 20, 4, 3, 3, 1), (30, 10, 9, 8, 3, 2, 2)]   # mem, batch, steps, eps_mem_batch, warmup, classes, seed
    for ci, (mem, bs, steps, nret, warmup, ncls, seed) in enumerate(cases):
        p = SimpleNamespace(buffer_tracker=True, eps_mem_batch=nret, warmup=warmup)

        class _B(object):
            pass
        b = _B()
        b.buffer_img = torch.zeros(mem, 3, 32, 32)
        b.buffer_label = torch.zeros(mem, dtype=torch.long)
        b.current_index = b.n_seen_so_far = 0
        b.params = p
        b.buffer_tracker = ref_bu.BufferClassTracker(10, "cpu")
        torch.manual_seed(seed)
        np.random.seed(seed)
        pyrandom.seed(seed)
        upd, mm = ref_res.Reservoir_update(p), ref_mm.MemMatch_retrieve(p)
        rng = np.random.default_rng(100 + seed)
        next_id = 1
        ys_all, ids_all, rec = [], [], []
        for s_ in range(steps):
            ys = rng.integers(0, ncls, bs).astype(np.int64)
            ids = np.arange(next_id, next_id + bs).astype(np.float32)
            next_id += bs
            x = torch.zeros(bs, 3, 32, 32)
            x[:, 0, 0, 0] = torch.from_numpy(ids)
            with R.quiet():
                cx, cy, mx, my = mm.retrieve(b)
            rec.append((cx[:, 0, 0, 0].numpy().copy() if cx.numel() else np.zeros(0, np.float32), cy.numpy().astype(np.int64) if cy.numel() else np.zeros(0, np.int64),
                        mx[:, 0, 0, 0].numpy().copy() if mx.numel() else np.zeros(0, np.float32), my.numpy().astype(np.int64) if my.numel() else np.zeros(0, np.int64)))
            upd.update(b, x, torch.from_numpy(ys))
            ys_all.append(ys); ids_all.append(ids)
        # the oracle's restatement replayed with the same seeds: same candidates, same partners, same slots, same generator positions
        st_np, st_py_draw = np.random.get_state()[1][:8].copy(), None
        torch.manual_seed(seed)
        np.random.seed(seed)
        pyrandom.seed(seed)
        ob, tr = O.OracleBuffer(mem, (3, 32, 32)), O.ClassTracker(10)
        for s_ in range(steps):
            cand, part = O.mem_match_indices(ob, tr, nret, warmup)
            assert np.array_equal(ob.img[cand][:, 0, 0, 0].numpy() if len(cand) else np.zeros(0, np.float32), rec[s_][0]), (ci, s_)
            assert np.array_equal(ob.img[part][:, 0, 0, 0].numpy() if len(part) else np.zeros(0, np.float32), rec[s_][2]), (ci, s_)
            x = torch.zeros(bs, 3, 32, 32)
            x[:, 0, 0, 0] = torch.from_numpy(ids_all[s_])
            O.reservoir_update(ob, x, torch.from_numpy(ys_all[s_]), tracker=tr)
        assert torch.equal(ob.label, b.buffer_label) and torch.equal(ob.img, b.buffer_img)
        assert np.array_equal(np.random.get_state()[1][:8], st_np)
        out["c%d_cfg" % ci] = np.array([mem, bs, steps, nret, warmup, ncls, seed], dtype=np.int64)
        out["c%d_ys" % ci] = np.stack(ys_all)
        out["c%d_ids" % ci] = np.stack(ids_all)
        for k, name in enumerate(("cand_id", "cand_y", "match_id", "match_y")):
            out["c%d_%s" % (ci, name)] = np.concatenate([r[k] for r in rec])
            out["c%d_%s_counts" % (ci, name)] = np.array([len(r[k]) for r in rec], dtype=np.int64)
        out["c%d_final_ids" % ci] = b.buffer_img[:, 0, 0, 0].numpy().copy()
        out["c%d_final_label" % ci] = b.buffer_label.numpy().copy()
        out["c%d_np_state" % ci] = np.random.get_state()[1][:8].astype(np.int64)
        out["c%d_py_draw" % ci] = np.float64(pyrandom.random())
        n_ret = int((out["c%d_match_id_counts" % ci] > 0).sum())
        assert n_ret >= 3, "case %d retrieves too rarely (%d)" % (ci, n_ret)
        print("mem_match case", ci, "retrievals with a match:", n_ret, "of", steps)
    out["n_cases"] = np.int64(len(cases))
    np.savez_compressed(os.path.join(GOLD, "mem_match.npz"), **out)
    print("mem_match ok")


def gen_resnet():
    """Seeded init + seeded input: forward outputs, loss, gradient / running-stat digests of the reference modules."""
    R.activate()
    from models.resnet import Reduced_ResNet18, SupConResNet
    import torch.nn as nn
    out = {}
    cases = [("rr18_c100", 32, 6, None), ("scr_mlp", 32, 6, "mlp"), ("rr18_mini", 84, 3, None)]
    for name, hw, n, head in cases:
        torch.manual_seed(11)
        if head is None:
            m = Reduced_ResNet18(100)
            if hw == 84:
                m.linear = nn.Linear(640, 100, bias=True)
        else:
            m = SupConResNet(head=head)
        rng = np.random.default_rng(5)
        x = torch.from_numpy(rng.random((n, 3, hw, hw)).astype(np.float32))
        y = torch.from_numpy(rng.integers(0, 100, n).astype(np.int64))
        st0 = {k: v.clone() for k, v in m.state_dict().items()}
        m.train()
        o = m(x)
        if head is None:
            loss = torch.nn.functional.cross_entropy(o, y)
        else:
            loss = (o * torch.linspace(-1, 1, o.numel()).view_as(o)).sum()
        loss.backward()
        # oracle check from the same initial state
        s = O.clone_state(st0)
        net = O.OracleNet(s, head=head, training=True)
        o2 = net.forward(x)
        l2 = torch.nn.functional.cross_entropy(o2, y) if head is None else (o2 * torch.linspace(-1, 1, o2.numel()).view_as(o2)).sum()
        l2.backward()
        assert (o - o2).abs().max() < 1e-5, (o - o2).abs().max()
        names = [k for k, _ in m.named_parameters()]
        for k, p in m.named_parameters():
            if p.grad is None:
                assert s[k].grad is None, k
            else:
                assert (p.grad - s[k].grad).abs().max() <= 1e-4 * (1 + p.grad.abs().max()), k
        m.eval()
        with torch.no_grad():
            fe = m.features(x)
        out[name + "_out"] = o.detach().numpy()
        out[name + "_loss"] = np.float64(float(loss))
        out[name + "_feat_eval"] = fe.numpy()
        out[name + "_grad_digest"] = np.array([[float(p.grad.double().sum()), float(p.grad.double().norm())] if p.grad is not None
                                               else [0.0, 0.0] for _, p in m.named_parameters()])
        # a few full gradients (first conv, a strided conv, a shortcut, a BN, the classifier/head)
        pick = [names[0], names[1]] + [k for k in names if k.endswith("layer2.0.conv1.weight") or k.endswith("layer2.0.shortcut.0.weight")
                                       or k.endswith("layer4.1.conv2.weight") or k.endswith("layer3.0.shortcut.1.weight")]
        pick += [names[-2]]
        for k in pick:
            g = dict(m.named_parameters())[k].grad
            out[name + "_g_" + k] = g.numpy() if g is not None else np.zeros(1, dtype=np.float32)
        out[name + "_picked"] = np.array(pick)
        sd = m.state_dict()
        out[name + "_running_digest"] = np.array([[float(v.double().sum()), float(v.double().norm())] for k, v in sd.items()
                                                  if k.endswith("running_mean") or k.endswith("running_var")])
        out[name + "_init_digest"] = np.array([[float(v.double().sum()), float(v.double().norm())] for k, v in st0.items()
                                               if v.is_floating_point()])
    np.savez_compressed(os.path.join(GOLD, "resnet.npz"), **out)
    print("resnet ok")


def run_reference_case(name):
    """Drives the reference agent through the synthetic tasks of a STEP_CASES entry; returns per-task records."""
    R.activate()
    from continuum.data_utils import setup_test_loader
    cfg = STEP_CASES[name]
    params = R.default_params(**case_params(cfg))
    seed_all(cfg["seed"])
    model, opt, agent = R.build_agent(params)
    tasks, tests = make_stream(cfg)
    with R.quiet():
        test_loaders = setup_test_loader(tests, params)
    recs = []
    for (x, y) in tasks:
        with R.quiet():
            agent.train_learner(x, y)
            acc = agent.evaluate(test_loaders)
        recs.append(dict(acc=np.asarray(acc, dtype=np.float64),
                         buf_label=agent.buffer.buffer_label.numpy().copy(),
                         buf_rowsum=agent.buffer.buffer_img.double().sum(dim=(1, 2, 3)).numpy(),
                         counters=np.array([agent.buffer.current_index, agent.buffer.n_seen_so_far], dtype=np.int64),
                         state=digest_state(model.state_dict())))
    return recs


def run_oracle_case(name, sort_fn):
    cfg = STEP_CASES[name]
    O.ARGSORT_DESC = sort_fn
    try:
        seed_all(cfg["seed"])
        ag = O.OracleAgent(cfg)
        tasks, tests = make_stream(cfg)
        recs = []
        for (x, y) in tasks:
            ag.train_learner(x, y)
            acc = ag.evaluate(tests)
            recs.append(dict(acc=np.asarray(acc, dtype=np.float64), buf_label=ag.buf.label.numpy().copy(),
                             buf_rowsum=ag.buf.img.double().sum(dim=(1, 2, 3)).numpy(),
                             counters=np.array([ag.buf.current_index, ag.buf.n_seen_so_far], dtype=np.int64),
                             state=digest_state(ag.state_dict())))
    finally:
        O.ARGSORT_DESC = O.argsort_desc_stable
    return recs


def gen_steps(only=None):
    """only: a list of case names to (re)generate into the existing fixture (the others keep their recorded arrays; a full run
    reproduces them bit for bit -- every case is seeded on its own)."""
    out = {}
    path = os.path.join(GOLD, "steps.npz")
    if only and os.path.exists(path):
        with np.load(path) as f:
            out = {k: f[k] for k in f.files if not any(k.startswith(n + "_") for n in only)}
    for name, cfg in STEP_CASES.items():
        if only and name not in only:
            continue
        recs = run_reference_case(name)
        recs2 = run_reference_case(name)   # determinism of the reference at a fixed thread count
        golden = cfg.get("golden", True)
        # the oracle executing torch's own (unstable) argsort must reproduce the reference bit-for-bit: pins its logic
        rt = run_oracle_case(name, O.argsort_desc_torch)
        # with the deterministic tie-break it must do so too for the golden cases (i.e. they are tie-free)
        rs = run_oracle_case(name, O.argsort_desc_stable) if golden else None
        for t, (r, r2) in enumerate(zip(recs, recs2)):
            for k in r:
                assert np.array_equal(r[k], r2[k]), (name, t, k)
                assert np.array_equal(r[k], rt[t][k]), "oracle(torch sort) != reference: %s task %d %s" % (name, t, k)
                if golden:
                    assert np.array_equal(r[k], rs[t][k]), "case %s is not tie-free (task %d, %s): pick another seed" % (name, t, k)
                    out["%s_t%d_%s" % (name, t, k)] = r[k]
        out[name + "_ntasks"] = np.int64(len(recs))
        out[name + "_golden"] = np.int64(1 if golden else 0)
        print("steps:", name, "golden" if golden else "oracle-pinned only", "acc", [np.round(r["acc"], 3).tolist() for r in recs])
    np.savez_compressed(os.path.join(GOLD, "steps.npz"), **out)


if __name__ == "__main__":
    assert R.available(), "reference tree not found"
    torch.set_num_threads(1)
    os.makedirs(GOLD, exist_ok=True)
    which = sys.argv[1:] or ["knn", "supcon", "ce_tricks", "kd", "buffer", "mem_match", "resnet", "steps"]
    if "knn" in which:
        gen_knn_sv()
    if "supcon" in which:
        gen_supcon()
    if "ce_tricks" in which:
        gen_ce_tricks()
    if "kd" in which:
        gen_kd()
    if "buffer" in which:
        gen_buffer_ops()
    if "mem_match" in which:
        gen_mem_match()
    if "resnet" in which:
        gen_resnet()
    if "steps" in which:
        gen_steps()
    for w in which:   # steps:case_a,case_b -> only those cases
        if w.startswith("steps:"):
            gen_steps(w[len("steps:"):].split(","))
