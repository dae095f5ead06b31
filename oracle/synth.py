"""TEST INFRASTRUCTURE ONLY — seeded synthetic class-incremental streams (no datasets on disk, no
network) and the small teacher-forced step cases shared by oracle/make_golden.py (reference side) and the tests
(oracle side on CPU, HIP side on the MI355X)."""
import random

import numpy as np
import torch

SHAPES = {"cifar10": (32, 32), "cifar100": (32, 32), "mini_imagenet": (84, 84)}

# Small cases: a handful of iterations per task so that a free-running comparison stays inside fp32 round-off.
STEP_CASES = {
    # ER random/random (BASELINE config 1 shape)
    "er_c10": dict(agent="ER", retrieve="random", update="random", data="cifar10", mem_size=50, eps_mem_batch=10, seed=0,
                   tasks=[[0, 1], [2, 3]], n_train=30, n_test=20),
    # SCR random/random (config 2 shape), identity augmentation on both sides (kornia unpinned)
    "scr_c100": dict(agent="SCR", retrieve="random", update="random", data="cifar100", mem_size=100, eps_mem_batch=20, seed=1,
                     tasks=[[3, 17], [40, 41]], n_train=25, n_test=20, temp=0.07, head="mlp"),
    # ER --retrieve ASER --update ASER (config 3 shape).  SV totals tie EXACTLY all the time (adjacent same-indicator
    # candidates share a value, aser_utils.py:39-52) and torch's CPU argsort is unstable, so a free-running reference
    # trajectory is only defined up to tie order: golden=False -> make_golden.py proves oracle(torch argsort) ==
    # reference bit-for-bit, and the GPU tests compare single teacher-forced steps tie-aware.
    "aser_c100": dict(agent="ER", retrieve="ASER", update="ASER", data="cifar100", mem_size=80, eps_mem_batch=10, seed=2,
                      tasks=[list(range(20))], n_train=8, n_test=4, k=3, n_smp_cls=1.5, aser_type="asvm", golden=False),
    # ER --retrieve MIR (config 4 shape, 84x84)
    "mir_mini": dict(agent="ER", retrieve="MIR", update="random", data="mini_imagenet", mem_size=40, eps_mem_batch=5, seed=3,
                     tasks=[[5, 6]], n_train=20, n_test=10, subsample=20),
    # ER --retrieve MIR at 32x32, two tasks (seed chosen tie-free: saturated logits give exactly-zero scores otherwise)
    "mir_c10": dict(agent="ER", retrieve="MIR", update="random", data="cifar10", mem_size=40, eps_mem_batch=10, seed=4,
                    tasks=[[0, 1, 2, 3, 4], [5, 6, 7, 8, 9]], n_train=10, n_test=8, subsample=20),
    # review trick (agents/base.py:62-88): one pass over the buffer after every task with gradients / 10; ER with the NCM classifier
    # (ncm_trick) so that evaluate()'s exemplar-mean branch is pinned for a plain ResNet as well
    "er_review": dict(agent="ER", retrieve="random", update="random", data="cifar10", mem_size=50, eps_mem_batch=10, seed=6,
                      tasks=[[0, 1], [2, 3]], n_train=30, n_test=20, trick=dict(review_trick=True, ncm_trick=True)),
    # SCR + review trick (the paper's SCR setting, config_CVPR/agent/scr/scr_5k.yml:9-10: temp 0.1 + review_trick)
    "scr_review": dict(agent="SCR", retrieve="random", update="random", data="cifar100", mem_size=60, eps_mem_batch=20, seed=7,
                       tasks=[[3, 17], [40, 41]], n_train=25, n_test=20, temp=0.1, head="mlp", trick=dict(review_trick=True)),
    # SCR with 3 slots for 4 classes: at the second evaluate() at least one seen class has no exemplar -> the random class mean of
    # agents/base.py:135-137 (a torch.normal draw on the CPU generator, which also shifts every later RNG draw)
    "scr_tiny": dict(agent="SCR", retrieve="random", update="random", data="cifar100", mem_size=3, eps_mem_batch=2, seed=10,
                     tasks=[[3, 17], [40, 41]], n_train=15, n_test=20, temp=0.07, head="mlp"),
    # ER --retrieve match with the BufferClassTracker (utils/buffer/sc_retrieve.py, buffer_utils.py:29-49,163-203)
    "er_match": dict(agent="ER", retrieve="match", update="random", data="cifar10", mem_size=60, eps_mem_batch=10, seed=8,
                     tasks=[[0, 1], [2, 3]], n_train=30, n_test=20, buffer_tracker=True, warmup=2),
    # ER --update GSS (utils/buffer/gss_greedy_update.py): eval-mode per-sample gradients, cosine similarity, multinomial draws
    # (single-class tasks: the first batches of class 1 have gradients pointing away from every memory gradient, max cosine < 0, so
    # the replacement branch :23-43 with its two multinomial draws is taken twice in this run)
    # the KD tricks of the ER loop (agents/exp_replay.py:42-47,64-69; utils/kd_manager.py): kd_trick_star ALONE never gets a teacher
    # (agents/base.py:90 takes the copy only for kd_trick), so from the second task on it just scales the cross-entropy by
    # 1 / sqrt(t + 1); with BOTH tricks the distillation term enters twice, through the same teacher forward
    "er_kdstar": dict(agent="ER", retrieve="random", update="random", data="cifar10", mem_size=50, eps_mem_batch=10, seed=11,
                      tasks=[[0, 1], [2, 3]], n_train=30, n_test=20, trick=dict(kd_trick_star=True)),
    "er_kdboth": dict(agent="ER", retrieve="random", update="random", data="cifar10", mem_size=50, eps_mem_batch=10, seed=12,
                      tasks=[[0, 1], [2, 3], [4, 5]], n_train=30, n_test=20, trick=dict(kd_trick=True, kd_trick_star=True)),
    "er_gss": dict(agent="ER", retrieve="random", update="GSS", data="cifar10", mem_size=20, eps_mem_batch=10, seed=9,
                   tasks=[[0], [1]], n_train=30, n_test=20, gss_mem_strength=3, gss_batch_size=5, free_run_gpu=False),
}


def case_params(cfg):
    keys = ("agent", "retrieve", "update", "data", "mem_size", "eps_mem_batch", "seed", "temp", "head", "k", "n_smp_cls", "aser_type",
            "subsample", "buffer_tracker", "warmup", "gss_mem_strength", "gss_batch_size")
    p = {k: cfg[k] for k in keys if k in cfg}
    p["num_tasks"] = len(cfg["tasks"])
    if "trick" in cfg:
        trick = {k: False for k in ('labels_trick', 'kd_trick', 'separated_softmax', 'review_trick', 'ncm_trick', 'kd_trick_star')}
        trick.update(cfg["trick"])
        p["trick"] = trick
    return p


def seed_all(seed):
    """general_main.py:12-14."""
    np.random.seed(seed)
    random.seed(seed)
    torch.manual_seed(seed)


def class_images(cls, n, hw, rng, blend=0.5):
    """Class-conditional uint8 images: per-class prototype blended with uniform noise."""
    h, w = hw
    proto = np.random.default_rng(1234 + int(cls)).integers(0, 256, (h, w, 3)).astype(np.float32)
    noise = rng.integers(0, 256, (n, h, w, 3)).astype(np.float32)
    return np.clip(blend * proto[None] + (1 - blend) * noise, 0, 255).astype(np.uint8)


def make_stream(cfg):
    """Returns (tasks, tests): lists of (x uint8 [N,H,W,3], y int64 [N]) per task, classes interleaved."""
    hw = SHAPES[cfg["data"]]
    rng = np.random.default_rng(9000 + cfg["seed"])
    tasks, tests = [], []
    for classes in cfg["tasks"]:
        for store, n in ((tasks, cfg["n_train"]), (tests, cfg["n_test"])):
            xs = np.concatenate([class_images(c, n, hw, rng) for c in classes], 0)
            ys = np.concatenate([np.full(n, c, dtype=np.int64) for c in classes], 0)
            store.append((xs, ys))
    return tasks, tests


def digest_state(state_dict):
    """[n_tensors, 3] float64: (sum, L2 norm, first element) of every floating tensor in state_dict order."""
    rows = []
    for k, v in state_dict.items():
        if torch.is_tensor(v) and v.is_floating_point():
            d = v.detach().double().cpu().reshape(-1)
            rows.append([float(d.sum()), float(d.norm()), float(d[0])])
    return np.array(rows, dtype=np.float64)


def throughput_stream(data, n_classes_per_task, n_tasks, n_per_class, seed):
    """Uniform-noise uint8 stream for pure-throughput runs."""
    hw = SHAPES[data]
    rng = np.random.default_rng(seed)
    out = []
    for t in range(n_tasks):
        classes = list(range(t * n_classes_per_task, (t + 1) * n_classes_per_task))
        n = n_per_class * len(classes)
        x = rng.integers(0, 256, (n,) + hw + (3,), dtype=np.uint8)
        y = np.repeat(np.array(classes, dtype=np.int64), n_per_class)
        out.append((x, y))
    return out
