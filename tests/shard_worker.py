"""Worker of test_gpu_parity2.test_sharded_runs_two_processes: one rank of `ocl_amd.run.sharded_runs` under torch.distributed.run
(one independent run per rank, one all_gather of the accuracy arrays: experiment/run.py:34 sharded)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import ocl_amd  # noqa: F401
    from ocl_amd import dist as odist
    from ocl_amd.run import sharded_runs
    from oracle.synth import STEP_CASES, make_stream
    from test_gpu_steps import make_params
    backend = os.environ.get("OCL_SHARD_BACKEND", "gloo")
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local % torch.cuda.device_count())
    rank, world, _ = odist.init_from_env(backend=backend)
    cfg = dict(STEP_CASES["er_c10"])
    params = make_params(cfg)

    def stream(seed):
        return make_stream(dict(cfg, seed=seed))
    accs, extras, perf = sharded_runs(params, stream, base_seed=5)
    assert accs.shape == (world, 2, 2) and extras.shape == (world, 2), (accs.shape, extras.shape)
    assert (accs >= 0).all() and (accs <= 1).all() and (extras[:, 1] == 120).all()
    # every rank holds every rank's array; rank r's own row is what it computed
    gathered = [None] * world
    torch.distributed.all_gather_object(gathered, accs.tolist())
    assert all(g == gathered[0] for g in gathered)
    assert perf is not None and len(perf) == 5 and all(np.isfinite(p[0]) for p in perf)
    if rank == 0:
        print("SHARDED_OK world=%d backend=%s avg_end_acc=%.3f" % (world, torch.distributed.get_backend(), perf[0][0]))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
