"""ER + ASER on bench.py's accuracy.aser stream (500 slots, tasks of 10 new classes, 20 images per class), HIP against the CPU oracle ONE STEP AT A
TIME from identical state (weights, BatchNorm buffers, memory, class caches, host RNG) -- through the fill phase, the full-memory phase and the task
boundaries.  Per step: host RNG state, candidate / evaluation index sets, minority count, Shapley scores under the kernel's own neighbour order,
validity of both selections under the ORACLE's scores, combined loss, the SGD update, and the BatchNorm RUNNING statistics after the step (the
co-simulations of tests/ teacher-force them before every step and never compared them after it).  Legitimate tie divergences are re-synchronised
from the oracle.  Question behind it: accuracy.aser's HIP runs end lower than the oracle's (0.16 vs 0.22 - 0.25 over 5 - 10 runs).

    python scripts/aser_cosim_probe.py [n_steps] [seed]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import importlib.util  # noqa: E402

spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
import test_gpu_steps as TS  # noqa: E402
from oracle import ocl_oracle as O  # noqa: E402


def main():
    n_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    cuda = torch.device("cuda:0")
    from ocl_amd.plugins.buffer_utils import ClassBalancedRandomSampling as CB
    c = bench.ACC_CFG
    tasks, _ = bench.accuracy_stream(seed, c["n_tasks"], c["classes_per_task"], bench.ACC_ASER["n_train"], c["n_test"], c["blend"], kind="texture_prototype")
    xs = np.concatenate([np.asarray(t[0]) for t in tasks], 0)
    ys = np.concatenate([np.asarray(t[1]) for t in tasks], 0).astype(np.int64)
    print("stream:", xs.shape, xs.dtype, "labels of the first three steps", ys[:30].tolist())
    cfg = dict(bench.WORKLOADS["aser"], seed=seed, tasks=[list(range(100))], n_train=1, n_test=1, mem_size=bench.ACC_ASER["mem_size"])
    eps = 1e-5
    bad = []
    stats = dict(ret=0, upd=0, resync=0, bn_max=0.0, upd_err_max=0.0, loss_max=0.0)
    for it, ev, ol, chk in TS.cosim(cfg, n_steps, cuda, x_stream=(xs, ys)):
        agent, oa, model = chk["agent"], chk["oa"], chk["model"]
        def flag(msg):
            bad.append((it, msg))
            print("!! step %d: %s" % (it, msg))
        if not chk["rng_equal"]:
            flag("host RNG streams diverged")
        ret_ev = [e for t, e in ev if t == "aser_retrieve"]
        upd_ev = [e for t, e in ev if t == "aser_update"]
        same_rows = False
        if ol.get("cand") is not None:
            if not ret_ev:
                flag("oracle retrieved by ASER, HIP did not")
            else:
                r = ret_ev[0]
                if not np.array_equal(r["cand_ind"], ol["cand"]):
                    flag("retrieve: candidate set differs (%d vs %d)" % (len(r["cand_ind"]), len(ol["cand"])))
                else:
                    sv_adv = TS._sv_given_order(ol["ret_aux"]["adv"], r["order_adv"], cfg["k"])
                    sv_coop = TS._sv_given_order(ol["ret_aux"]["coop"], r["order_coop"], cfg["k"])
                    sv_exp = O.aser_score(sv_adv, sv_coop, "asvm")
                    if np.abs(r["sv"] - sv_exp).max() > eps:
                        flag("retrieve: score error %.3g" % np.abs(r["sv"] - sv_exp).max())
                    k = len(r["ret"])
                    thr = np.sort(sv_exp)[::-1][k - 1]
                    pos = {cc: j for j, cc in enumerate(ol["cand"].tolist())}
                    if len(r["ret"]) != len(ol["ret_idx"]) or min(sv_exp[pos[cc]] for cc in r["ret"].tolist()) < thr - eps:
                        flag("retrieve: not a valid top-%d (oracle took %d)" % (k, len(ol["ret_idx"])))
                    same_rows = np.array_equal(np.sort(r["ret"]), np.sort(ol["ret_idx"]))
                stats["ret"] += 1
        else:
            rr = [e["indices"] for t, e in ev if t == "random_retrieve"]
            if ret_ev:
                flag("HIP retrieved by ASER, the oracle at random")
            elif len(rr) and not np.array_equal(rr[0], ol["ret_idx"]):
                flag("random retrieval differs")
            same_rows = True
        lc = [e["loss"] for t, e in ev if t == "er_loss_combined"]
        if same_rows and lc:
            stats["loss_max"] = max(stats["loss_max"], abs(lc[0] - ol["loss"]))
            stats["upd_err_max"] = max(stats["upd_err_max"], chk["upd_err"])
            if abs(lc[0] - ol["loss"]) > 1e-4 * (1 + abs(ol["loss"])):
                flag("combined loss %.6f vs %.6f" % (lc[0], ol["loss"]))
            if chk["upd_err"] > 1e-2:
                flag("SGD update off by %.3g in norm" % chk["upd_err"])
            # BatchNorm running statistics after the step (three train-mode forwards per iteration on both sides)
            so, sm = oa.state_dict(), model.state_dict()
            for kname in so:
                if "running" in kname or "num_batches" in kname:
                    a, b = so[kname].double().cpu().numpy(), sm[kname].double().cpu().numpy()
                    d = float(np.abs(a - b).max() / (1e-12 + np.abs(a).max()))
                    stats["bn_max"] = max(stats["bn_max"], d)
                    if d > 1e-4:
                        flag("BatchNorm buffer %s off by %.3g (relative to its largest entry)" % (kname, d))
                        break
        if (ol.get("upd") is not None) != bool(upd_ev):
            flag("update branch differs: oracle %s, HIP %s" % (ol.get("upd") is not None, bool(upd_ev)))
        elif upd_ev:
            u, ou = upd_ev[0], ol["upd"]
            if not (np.array_equal(u["eval_indices"], ou["eval_indices"]) and np.array_equal(u["cand_ind"], ou["cand_ind"])):
                flag("update: evaluation / candidate index sets differ")
            elif u["n_minority"] != ou["n_minority"]:
                flag("update: n_minority %d vs %d" % (u["n_minority"], ou["n_minority"]))
            elif same_rows:
                sv_exp = TS._sv_given_order(ou["aux"], u["knn_order"], cfg["k"]).sum(0)
                if np.abs(u["sv"] - sv_exp).max() > eps * 10:
                    flag("update: score error %.3g" % np.abs(u["sv"] - sv_exp).max())
                n_buf = len(u["cand_ind"])
                thr = np.sort(sv_exp)[::-1][n_buf - 1]
                large, small = u["order"][:n_buf], u["order"][n_buf:]
                if sv_exp[large].min() < thr - 1e-4 or (len(small) and sv_exp[small].max() > thr + 1e-4):
                    flag("update: invalid Shapley partition")
                stats["upd"] += 1
        if not TS._buffers_equal(agent, oa):
            stats["resync"] += 1
            b = agent.buffer
            b.buffer_img.copy_(oa.buf.img.to(cuda))
            b.buffer_label.copy_(oa.buf.label.to(cuda))
            b.label_host[:] = oa.buf.label.numpy()
            b.current_index, b.n_seen_so_far = oa.buf.current_index, oa.buf.n_seen_so_far
            CB.update_cache(b.label_host, 100)
            oa.cache.update(oa.buf.label, 100)
            CB.class_num_cache = oa.cache.count.clone()
        if it % 10 == 9:
            print("step %d: %s, buffer %d / seen %d, problems so far %d" % (it, stats, oa.buf.current_index, oa.buf.n_seen_so_far, len(bad)))
    print("DONE: %d steps, %s" % (n_steps, stats))
    print("problems: %d" % len(bad))
    for it, m in bad[:40]:
        print("  step %d: %s" % (it, m))


if __name__ == "__main__":
    main()
