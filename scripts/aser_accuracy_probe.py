"""HIP side of bench.py's accuracy.aser alone (ER + ASER, 500 slots, 2000-image texture stream, five seeds): end accuracy per seed under whatever
environment switches the caller sets -- is a difference against the oracle's distribution tied to one of the schedule changes?

    OCL_ASER_SPLIT=0 python scripts/aser_accuracy_probe.py

PROBE_PERT=n: every seed n times, run j > 0 with every initial weight scaled by (1 + 1e-7 * N(0, 1)) from a private generator (about one
fp32 ulp) -- the HIP side's spread at a FIXED seed, the counterpart of scripts/aser_oracle_chaos_probe.py."""
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
    import contextlib
    from ocl_amd import run as run_mod
    from ocl_amd.run import single_run
    make_model = run_mod.setup_architecture
    pert = [0]

    def perturbed_model(p):
        model = make_model(p)
        if pert[0]:
            g = torch.Generator().manual_seed(977 * pert[0])
            with torch.no_grad():
                for w in model.parameters():
                    w.mul_(1.0 + 1e-7 * torch.randn(w.shape, generator=g))
        return model
    run_mod.setup_architecture = perturbed_model
    n_pert = int(os.environ.get("PROBE_PERT", "1"))
    torch.cuda.set_device(0)
    c = bench.ACC_CFG
    mem = int(os.environ.get("PROBE_MEM", bench.ACC_ASER["mem_size"]))
    out = []
    seeds = [int(v) for v in os.environ.get("PROBE_SEEDS", "").split(",") if v] or bench.aser_seeds([0, 100, 200])
    for seed in [s for s in seeds for _ in range(n_pert)]:
        pert[0] = len(out) % n_pert
        tasks, tests = bench.accuracy_stream(seed, c["n_tasks"], c["classes_per_task"], bench.ACC_ASER["n_train"], c["n_test"], c["blend"], kind="texture_prototype")
        params = bench.make_params(dict(bench.WORKLOADS["aser"], num_tasks=c["n_tasks"], mem_size=mem))
        with contextlib.redirect_stdout(sys.stderr):
            acc, tt, n_img, ag = single_run(params, tasks, tests, seed)
        out.append(float(acc[-1].mean()))
    print("end accuracy per seed %s mean %.4f" % ([round(v, 4) for v in out], float(np.mean(out))))


if __name__ == "__main__":
    main()
