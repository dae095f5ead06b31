"""Worker of test_gpu_parity2.test_rccl_group_of_one_runs_the_exchange: the RCCL ("nccl") branch of ocl_amd.dist on a one-GPU box.

A world-size-1 process group over RCCL on cuda:0; the run loop's only exchange (dist.gather_runs: one all_gather of the [T, T] accuracy
array + scalars, experiment/run.py:34 sharded) and the bench's timing reductions (max_over_ranks / sum_over_ranks / gather_scalars) go
through the device collective with device-resident payloads, after a real sharded run."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import ocl_amd  # noqa: F401
    from ocl_amd import dist as odist
    from ocl_amd.run import sharded_runs
    from oracle.synth import STEP_CASES, make_stream
    from test_gpu_steps import make_params
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:%s" % os.environ.get("OCL_TEST_PORT", "29547"), rank=0, world_size=1)
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    rng = np.random.default_rng(3)
    acc = rng.random((10, 10))
    accs, extras = odist.gather_runs(acc, extra=[1.5, 250.0])
    assert accs.shape == (1, 10, 10) and np.array_equal(accs[0], acc) and extras.tolist() == [[1.5, 250.0]]       # fp64 through the device, bit-exact
    assert odist.max_over_ranks(0.123456789) == 0.123456789 and odist.sum_over_ranks(2.5) == 2.5
    assert odist.gather_scalars([1.0, 2.0, 3.0]).tolist() == [[1.0, 2.0, 3.0]]
    odist.barrier()
    cfg = dict(STEP_CASES["er_c10"])
    accs, extras, perf = sharded_runs(make_params(cfg), lambda seed: make_stream(dict(cfg, seed=seed)), base_seed=5)
    assert accs.shape == (1, 2, 2) and (accs >= 0).all() and (accs <= 1).all() and extras[0, 1] == 120
    assert perf is None            # (the summary statistics need two runs; world of one = one run)
    print("RCCL_OK world=%d backend=%s end_acc=%.3f" % (dist.get_world_size(), dist.get_backend(), float(accs[0, -1].mean())))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
