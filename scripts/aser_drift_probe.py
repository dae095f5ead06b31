"""Why does the ASER step get slower with the step count (bench --repeats 15: 2.82 -> 3.00 ms over 1500 steps)?  Per repeat of 100 steps:
step time, host time inside the class-balanced draws (csrc/hostc.c), rows of the kNN calls (evaluation x candidate: the eval-mode feature
passes scale with them), size statistics of the class table's sets.

    python scripts/aser_drift_probe.py [--repeats 12] > profiles/r6_aser_drift_probe.txt"""
import argparse
import importlib.util
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--repeats", type=int, default=12)
    ap.add_argument("--steps", type=int, default=100)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    params, model, agent, hw, ncls = bench.build_agent("aser", 0, dev)
    bs = params.batch
    xp, yp = bench.synth_u8(100 * bs, hw, ncls, 5)
    xt, yt = bench.synth_u8(args.steps * bs, hw, ncls, 2)
    xp_d, xt_d = torch.from_numpy(xp).to(dev), torch.from_numpy(xt).to(dev)
    from ocl_amd import ops
    from ocl_amd.plugins import buffer_utils as BU
    acc = dict(draw_s=0.0, draws=0, knn=[])
    hostc = BU._hostc
    inner = hostc.cbrs_sample

    class Wrap(object):
        def __getattr__(self, k):
            return getattr(hostc, k)

        def cbrs_sample(self, *a):
            t = time.perf_counter()
            r = inner(*a)
            dt = time.perf_counter() - t
            acc["draw_s"] += dt
            acc["draws"] += 1
            kind = "excl" if (len(a) > 1 and a[1] is not None and len(a[1])) else "plain"
            acc.setdefault(kind, []).append(dt)
            return r
    BU._hostc = Wrap()
    knn0 = ops.knn_sv

    def knn(eval_f, eval_y, cand_f, cand_y, k, want_order=False):
        acc["knn"].append((eval_f.shape[0], cand_f.shape[0]))
        return knn0(eval_f, eval_y, cand_f, cand_y, k, want_order=want_order)
    ops.knn_sv = knn
    import ocl_amd.plugins.aser_utils as AU
    AU.ops = ops
    agent.train_learner(xp_d, yp)
    torch.cuda.synchronize()
    for r in range(args.repeats):
        acc.update(draw_s=0.0, draws=0, knn=[], excl=[], plain=[])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        agent.train_learner(xt_d, yt)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps * 1e3
        k = np.array(acc["knn"]) if acc["knn"] else np.zeros((0, 2))
        cache = BU.ClassBalancedRandomSampling.class_index_cache
        sizes = np.array([len(s) for s in cache.values()])
        print("repeat %2d: %.3f ms per step | class-balanced draws %.1f us each (%d; without exclusions %.1f us, with %.1f us) | kNN calls per step %.1f, rows eval %.1f cand %.1f | class sets: %d classes, "
              "sizes min %d median %d max %d, empty %d" % (r, dt, acc["draw_s"] / max(1, acc["draws"]) * 1e6, acc["draws"], np.mean(acc["plain"]) * 1e6 if acc["plain"] else 0, np.mean(acc["excl"]) * 1e6 if acc["excl"] else 0, len(k) / args.steps,
                                                            k[:, 0].mean() if len(k) else 0, k[:, 1].mean() if len(k) else 0, len(sizes), sizes.min(), np.median(sizes),
                                                            sizes.max(), int((sizes == 0).sum())))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
