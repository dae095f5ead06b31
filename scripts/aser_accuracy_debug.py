"""How many Shapley-valued retrievals / replacements does one run of bench.py's accuracy.aser perform?  (HIP agent; --oracle: the CPU oracle.)"""
import importlib.util
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def main():
    c = bench.ACC_CFG
    seed = 0
    tasks, tests = bench.accuracy_stream(seed, c["n_tasks"], c["classes_per_task"], bench.ACC_ASER["n_train"], c["n_test"], c["blend"], kind="texture_prototype")
    if "--oracle" in sys.argv:
        import random
        from oracle import ocl_oracle as O
        cfg = dict(bench.WORKLOADS["aser"], seed=seed, tasks=[[0]] * c["n_tasks"], n_train=0, n_test=0, mem_size=bench.ACC_ASER["mem_size"])
        np.random.seed(seed); random.seed(seed); torch.manual_seed(seed)
        calls = [0]
        k0 = O.knn_sv

        def k(*a, **kw):
            calls[0] += 1
            return k0(*a, **kw)
        O.knn_sv = k
        oa = O.OracleAgent(cfg)
        accs = []
        for ti, (x, y) in enumerate(tasks):
            oa.train_learner(x, y)
            accs.append(oa.evaluate(tests))
            print("oracle task %d: knn_sv calls so far %d, buffer index %d seen %d, end acc row mean %.3f" % (ti, calls[0], oa.buf.current_index, oa.buf.n_seen_so_far, float(np.mean(accs[-1]))))
        return
    from ocl_amd import ops
    from ocl_amd.run import single_run
    import ocl_amd.plugins.aser_utils as AU
    torch.cuda.set_device(0)
    calls = [0]
    k0 = ops.knn_sv

    def k(*a, **kw):
        calls[0] += 1
        return k0(*a, **kw)
    ops.knn_sv = k
    AU.ops = ops
    params = bench.make_params(dict(bench.WORKLOADS["aser"], num_tasks=c["n_tasks"], mem_size=bench.ACC_ASER["mem_size"]))
    print("params: mem_size", params.mem_size, "retrieve", params.retrieve, "update", params.update, "eps_mem_batch", params.eps_mem_batch)
    acc, tt, n_img, ag = single_run(params, tasks, tests, seed)
    print("hip: knn_sv calls %d, buffer index %d seen %d, end acc %.3f, per task %s" % (calls[0], ag.buffer.current_index, ag.buffer.n_seen_so_far, float(acc[-1].mean()), np.round(acc[-1], 3).tolist()))


if __name__ == "__main__":
    main()
