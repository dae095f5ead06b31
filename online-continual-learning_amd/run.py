"""experiment/run.py:17-87 `multiple_run` for the hot-path agents: per run build model / optimiser / agent, per task
train_learner + evaluate; runs are sharded one-per-rank (dist.py) instead of looped."""
import time

import numpy as np
import torch

from . import dist as odist
from . import name_match
from .data import setup_test_loader
from .metrics import compute_performance
from .setup_elements import setup_architecture, setup_opt
from .utils import maybe_cuda


def single_run(params, tasks, tests, seed):
    """One run of experiment/run.py:36-56 on this rank's GPU.  tasks/tests: lists of (x uint8 NHWC, y int64)."""
    import random
    np.random.seed(seed)
    random.seed(seed)
    torch.manual_seed(seed)
    model = setup_architecture(params)
    model = maybe_cuda(model, params.cuda)
    opt = setup_opt(params.optimizer, model, params.learning_rate, params.weight_decay)
    agent = name_match.agents[params.agent](model, opt, params)
    test_loaders = setup_test_loader(tests, params)
    tmp_acc = []
    eval_s = []
    t_train = 0.0
    n_img = 0
    for i, (x_train, y_train) in enumerate(tasks):
        t0 = time.perf_counter()
        agent.train_learner(x_train, y_train)
        torch.cuda.synchronize()
        t_train += time.perf_counter() - t0
        n_img += (len(y_train) // params.batch) * params.batch
        t0 = time.perf_counter()
        tmp_acc.append(agent.evaluate(test_loaders))
        torch.cuda.synchronize()
        eval_s.append(time.perf_counter() - t0)
    agent.evaluate_seconds = eval_s      # wall time of evaluate() after every task (the bench line's evaluate_ms)
    return np.array(tmp_acc), t_train, n_img, agent


def sharded_runs(params, make_stream_fn, base_seed=0):
    """Each rank performs ONE independent run (seed base+rank), then the accuracy arrays are all-gathered and every
    rank computes the reference's summary metrics."""
    rank, world, local = odist.init_from_env()
    import os
    odist.pin_to_gpu_numa(local, int(os.environ.get("LOCAL_WORLD_SIZE", str(world))))   # this rank's launch loop next to its GPU
    seed = odist.run_seed(base_seed, rank)
    tasks, tests = make_stream_fn(seed)
    acc, t_train, n_img, _ = single_run(params, tasks, tests, seed)
    accs, extras = odist.gather_runs(acc, extra=[t_train, n_img])
    perf = compute_performance(accs) if accs.shape[0] > 1 else None
    return accs, extras, perf
