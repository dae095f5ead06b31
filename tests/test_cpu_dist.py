"""CPU: the N>1 path (one independent run per rank + ONE all_gather of the accuracy arrays) with world_size 2 on gloo."""
import os
import socket
import subprocess
import sys

import numpy as np

from conftest import ROOT

_WORKER = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, %(root)r)
import ocl_amd
from ocl_amd import dist as odist
from ocl_amd.metrics import compute_performance
rank, world, local = odist.init_from_env(backend="gloo")
seed = odist.run_seed(7, rank)
rng = np.random.default_rng(seed)
acc = np.tril(rng.random((3, 3)))
accs, extras = odist.gather_runs(acc, extra=[1.5 + rank, 100 * (rank + 1)])
tmax = odist.max_over_ranks(0.25 * (rank + 1))
tsum = odist.sum_over_ranks(10 * (rank + 1))
per_rank = odist.gather_scalars([2.0 + rank, -1.0 * rank])
odist.barrier()
if rank == 0:
    perf = compute_performance(accs)
    print(json.dumps(dict(shape=list(accs.shape), accs=accs.tolist(), extras=extras.tolist(), tmax=tmax, tsum=tsum, end=perf[0][0], per_rank=per_rank.tolist())))
'''


def _launch_two_ranks(script):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=180) for p in procs]
    return procs, outs


def test_two_rank_metric_allgather_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER % dict(root=ROOT))
    procs, outs = _launch_two_ranks(script)
    if not all(p.returncode == 0 for p in procs):   # the rendezvous port is taken between probing and use once in a long while: one more try, new port
        first = outs
        procs, outs = _launch_two_ranks(script)
        assert all(p.returncode == 0 for p in procs), (first, outs)
    import json
    res = json.loads(outs[0][0].strip().splitlines()[-1])
    assert res["shape"] == [2, 3, 3]
    for r in range(2):
        exp = np.tril(np.random.default_rng(7 + r).random((3, 3)))
        assert np.allclose(np.array(res["accs"][r]), exp)
        assert res["extras"][r] == [1.5 + r, 100.0 * (r + 1)]
    assert res["tmax"] == 0.5 and res["tsum"] == 30.0
    assert res["per_rank"] == [[2.0, -0.0], [3.0, -1.0]]
    exp_end = np.mean([np.mean(np.tril(np.random.default_rng(7 + r).random((3, 3)))[-1]) for r in range(2)])
    assert abs(res["end"] - exp_end) < 1e-12
