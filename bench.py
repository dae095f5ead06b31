#!/usr/bin/env python
"""bench.py — replay-step throughput of the MI355X-native hot path (BASELINE.json metric).

One "step" = one iteration of the replay inner loop over one stream minibatch of 10 synthetic images.  Default
workload (N=1): BASELINE.json configs[1] — SCR random/random, Split-CIFAR100 shape, mem_size 5000 (full),
eps_mem_batch 100, temp 0.07: retrieve 100 rows from the device-resident buffer, augment, 110+110 views through
SupConResNet (fwd+bwd, per-view BatchNorm), SupCon loss, SGD, reservoir update.  Inputs (uint8 task tensor, replay
buffer, weights) are resident in HBM before the timed region.  N>1: one independent stream per rank/GPU (weak
scaling, no data-path collective; one all_gather/all_reduce of scalars at the end).

Prints ONE JSON line on rank 0 (see the task's bench contract) including
  roofline     fp32-MFMA utilisation of the conv implicit-GEMM kernels, from HIP-event timing of every launch
  cpu_baseline the oracle restatement of the same step timed on this box's host cores (rank 0, N=1 only)
  also.aser    the second headline configuration of BASELINE.json's metric (configs[2]: ER + ASER retrieve / update, mem 5000,
               k = 3): its own timed throughput, conv roofline fraction and kNN / buffer-path GB/s from the live HIP-event legs
  accuracy     the "final avg accuracy" half of the metric: a short class-incremental Split-CIFAR100-shaped run (10 tasks x 10
               classes, class-prototype images) per rank, evaluate() after every task, accuracy arrays all-gathered over the ranks
               and summarised by experiment/metrics.py's formulas; at N=1 the CPU oracle runs the same stream for comparison.
"""
import argparse
import contextlib
import subprocess
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

# algorithmic work per image, Reduced-ResNet18 (MACs from forward hooks on the reference)
MACS = {32: dict(fwd=54636160, stem=552960, fc=16000), 84: dict(fwd=385237440, stem=3810240, fc=64000)}
KNN_BUFFER_BYTES = {"er": 0.25e6, "scr": 1.35e6, "aser": 6.5e6, "mir": 18.1e6}
PEAK_FP32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 / 32x32x2 peak (= vector fp32 peak)

WORKLOADS = {
    # name: (agent, retrieve, update, data, mem_size, eps_mem_batch, extra)
    "scr": dict(agent="SCR", retrieve="random", update="random", data="cifar100", mem_size=5000, eps_mem_batch=100, temp=0.07, head="mlp"),
    "aser": dict(agent="ER", retrieve="ASER", update="ASER", data="cifar100", mem_size=5000, eps_mem_batch=10, k=3, n_smp_cls=1.5,
                 aser_type="asvm"),
    "er": dict(agent="ER", retrieve="random", update="random", data="cifar10", mem_size=1000, eps_mem_batch=10),
    "mir": dict(agent="ER", retrieve="MIR", update="random", data="mini_imagenet", mem_size=10000, eps_mem_batch=10, subsample=50),
}


def make_params(w, cuda=True):
    from types import SimpleNamespace
    trick = {k: False for k in ('labels_trick', 'kd_trick', 'separated_softmax', 'review_trick', 'ncm_trick', 'kd_trick_star')}
    p = dict(agent="ER", retrieve="random", update="random", data="cifar100", mem_size=5000, eps_mem_batch=10, cuda=cuda, epoch=1,
             batch=10, test_batch=128, verbose=False, optimizer="SGD", learning_rate=0.1, weight_decay=0, mem_iters=1, subsample=50,
             k=3, aser_type="asvm", n_smp_cls=1.5, num_tasks=10, temp=0.07, head="mlp", buffer_tracker=False, error_analysis=False,
             trick=trick)
    p.update(w)
    return SimpleNamespace(**p)


def synth_u8(n, hw, n_classes, seed):
    rng = np.random.default_rng(seed)
    x = rng.integers(0, 256, (n, hw, hw, 3), dtype=np.uint8)
    y = rng.integers(0, n_classes, n).astype(np.int64)
    return x, y


def flops_per_step(workload, hw, n_classes_in_buffer=100):
    """Algorithmic conv flops per step by kernel class (fwd+dgrad GEMM vs wgrad); 2 flops per MAC."""
    m = MACS[hw]
    conv_fwd = m["fwd"] - m["fc"]
    # train_imgs: images of the train-mode forwards; bwd_imgs: images whose backward actually RUNS (ASER mode: the reference
    # back-propagates the batch and memory passes and then discards those gradients, agents/exp_replay.py:76-84 -- the engine skips
    # those two backward passes, so only the combined pass of 20 images counts); eval_imgs: no_grad / eval-mode forwards
    if workload == "scr":
        train_imgs, bwd_imgs, eval_imgs = 220, 220, 0
    elif workload == "aser":
        c = n_classes_in_buffer
        train_imgs, bwd_imgs, eval_imgs = 40, 20, (10 + c) + (2 * c) + (c + 160)
    elif workload == "mir":
        train_imgs, bwd_imgs, eval_imgs = 20, 20, 100
    else:
        train_imgs, bwd_imgs, eval_imgs = 20, 20, 0
    gemm = 2.0 * (train_imgs * conv_fwd + bwd_imgs * (conv_fwd - m["stem"]) + eval_imgs * conv_fwd)
    wgrad = 2.0 * bwd_imgs * conv_fwd
    return gemm, wgrad


PMC_TRAFFIC_FILES = ("r6_scr_pmc_traffic.json", "r5_scr_pmc_traffic.json", "r4_scr_pmc_traffic.json", "r3_scr_pmc_traffic.json", "r2_scr_pmc_traffic.json")   # newest first: r5 = this tree (scripts/gpu_r5g.sh)


def pmc_traffic(kernel):
    """(HBM bytes per launch of `kernel`, source file) from the committed rocprofv3 PMC passes (profiles/r*_scr_pmc_traffic.json:
    FETCH_SIZE and WRITE_SIZE collected in separate --pmc runs of this same command, gfx950 FETCH_SIZE correction applied there);
    (None, None) when no profile is shipped.  Hardware counters cannot be read from inside the process: this figure is a
    committed measurement of an earlier run of the same command, NOT something this run measured."""
    for name in PMC_TRAFFIC_FILES:
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                k = json.load(f)["kernels"]
                k = k.get("conv_fwd_dgrad") or k.get(kernel) or k["conv_gemm_kernel"]   # r3: conv_t + conv_q launches together; (the round-2 file still carries the name of the kernel conv_t_kernel replaced)
                return k["hbm_bytes_per_launch"], "profiles/" + name
        except Exception:
            continue
    return None, None


def _pci_dir(index):
    from ocl_amd import dist as odist
    return odist._pci_dir(index)


def _read(path):
    try:
        with open(path) as f:
            return f.read()
    except Exception:
        return None


def gpu_env_sample(index):
    """One reading of THIS GPU's clocks and power from sysfs (microseconds of host time, no subprocess: it is taken inside the timed
    region, after the host has enqueued the last step and before it waits for the GPU).  Missing files -> None fields."""
    d = _pci_dir(index)
    out = dict(sclk_mhz=None, mclk_mhz=None, power_w=None)
    if d is None:
        return out
    import glob
    for key, name in (("sclk_mhz", "pp_dpm_sclk"), ("mclk_mhz", "pp_dpm_mclk")):
        txt = _read(os.path.join(d, name)) or ""
        for line in txt.splitlines():
            if "*" in line:
                try:
                    out[key] = int(line.split(":")[1].strip().lower().split("mhz")[0])
                except Exception:
                    pass
    for h in glob.glob(os.path.join(d, "hwmon", "hwmon*")):
        f1 = _read(os.path.join(h, "freq1_input"))
        if f1 and out["sclk_mhz"] is None:
            out["sclk_mhz"] = int(f1) // 1000000
        pw = _read(os.path.join(h, "power1_input")) or _read(os.path.join(h, "power1_average"))
        if pw:
            out["power_w"] = int(pw) / 1e6
    return out


def gpu_env_static(index):
    d = _pci_dir(index)
    out = dict(power_cap_w=None, perf_level=None, numa_node=None, pci=os.path.basename(d) if d else None)
    if d is None:
        return out
    import glob
    lvl = _read(os.path.join(d, "power_dpm_force_performance_level"))
    out["perf_level"] = lvl.strip() if lvl else None
    nn = _read(os.path.join(d, "numa_node"))
    out["numa_node"] = int(nn) if nn and nn.strip().lstrip("-").isdigit() else None
    for h in glob.glob(os.path.join(d, "hwmon", "hwmon*")):
        cap = _read(os.path.join(h, "power1_cap"))
        if cap:
            out["power_cap_w"] = int(cap) / 1e6
    return out


def build_agent(workload, seed, device):
    import ocl_amd  # noqa: F401
    from ocl_amd import name_match
    from ocl_amd.setup_elements import setup_architecture, setup_opt, n_classes, input_size_match
    import random
    w = WORKLOADS[workload]
    params = make_params(w)
    np.random.seed(seed)
    random.seed(seed)
    torch.manual_seed(seed)
    model = setup_architecture(params).to(device)
    opt = setup_opt("SGD", model, params.learning_rate, params.weight_decay)
    agent = name_match.agents[params.agent](model, opt, params)
    hw = input_size_match[params.data][1]
    ncls = n_classes[params.data]
    # steady state: fill the replay buffer through the update plugin (builds ASER's class cache the reference's way)
    rng = np.random.default_rng(seed + 1000)
    chunk = 500
    for s in range(0, params.mem_size, chunk):
        n = min(chunk, params.mem_size - s)
        ys = rng.integers(0, ncls, n).astype(np.int64)
        xs = torch.from_numpy(rng.random((n, 3, hw, hw), dtype=np.float32)).to(device)
        agent.buffer.update(xs, torch.from_numpy(ys).to(device), y_host=ys)
    assert agent.buffer.current_index == params.mem_size
    return params, model, agent, hw, ncls


def gpu_leg(args, rank, world, local, workload=None, steps=None, warmup=None):
    from ocl_amd import dist as odist
    from ocl_amd import ops
    workload = workload or args.workload
    steps = steps or args.steps
    warmup = args.warmup if warmup is None else warmup
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    params, model, agent, hw, ncls = build_agent(workload, args.seed + rank, device)
    bs = params.batch
    xw, yw = synth_u8(max(1, warmup) * bs, hw, ncls, 1 + rank)
    xt, yt = synth_u8(steps * bs, hw, ncls, 2 + rank)
    xw_d, xt_d = torch.from_numpy(xw).to(device), torch.from_numpy(xt).to(device)     # resident in HBM before timing
    # pre-roll: REAL steps of the same workload before the warm-up (reported as preroll_steps / preroll_ms).  The driver's command
    # (--steps 20 --warmup 5) gives the GPU 12 ms of work before a 45 ms timed region: kernel plans, torch's caching pools and the
    # GPU's clocks settle here instead of inside the timed steps (profiles/r4_driver_command_repro.txt: 2.17 vs 2.13 ms without it).
    # Nothing is skipped or cached: these are ordinary training steps on their own synthetic batches, and `warmup` still runs after.
    preroll_ms = 0.0
    if args.preroll > 0:
        xp, yp = synth_u8(args.preroll * bs, hw, ncls, 5 + rank)
        xp_d = torch.from_numpy(xp).to(device)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        agent.train_learner(xp_d, yp)
        torch.cuda.synchronize()
        preroll_ms = (time.perf_counter() - t0) * 1e3
    # warm-up (the flag's W steps)
    agent.train_learner(xw_d, yw)
    torch.cuda.synchronize()
    # timed region: EXACTLY `steps` iterations between barrier + synchronize on both sides, `repeats` times back to back; the line
    # reports the MEDIAN repeat (all of them listed beside it)
    env_static = gpu_env_static(local)
    reps = []
    for r in range(max(1, args.repeats)):
        odist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        agent.train_learner(xt_d, yt)          # EXACTLY args.steps iterations (drop_last, len = steps*batch)
        torch.cuda.synchronize()
        t_own = time.perf_counter() - t0       # this rank's own stream (before it waits for the others)
        odist.barrier()
        elapsed = time.perf_counter() - t0
        env = gpu_env_sample(local)            # clocks / power right after the timed region closed (the sysfs reads are in nobody's time)
        per_rank = odist.gather_scalars([t_own], device)[:, 0]
        elapsed = odist.max_over_ranks(elapsed, device)
        reps.append(dict(elapsed=elapsed, per_rank=[float(t) for t in per_rank], env=env))
    order = sorted(range(len(reps)), key=lambda i: reps[i]["elapsed"])
    med = reps[order[len(order) // 2]]
    elapsed, per_rank = med["elapsed"], med["per_rank"]
    total_steps = odist.sum_over_ranks(steps, device)
    sclk = [r["env"]["sclk_mhz"] for r in reps if r["env"]["sclk_mhz"] is not None]
    pw = [r["env"]["power_w"] for r in reps if r["env"]["power_w"] is not None]
    out = dict(elapsed=elapsed, total_steps=total_steps, hw=hw, bs=bs, steps=steps,
               per_rank_images_per_s=[float(steps * bs / t) for t in per_rank],
               repeats_ms_per_step=[r["elapsed"] / steps * 1e3 for r in reps], preroll_ms=preroll_ms, preroll_steps=args.preroll,
               env=dict(env_static, sclk_mhz=(sorted(sclk)[len(sclk) // 2] if sclk else None), sclk_mhz_range=[min(sclk), max(sclk)] if sclk else None,
                        mclk_mhz=med["env"]["mclk_mhz"], power_w=(sorted(pw)[len(pw) // 2] if pw else None),
                        source="sysfs of this GPU's PCI function, once per timed repeat right after its last step retired (median)"))

    # ---- roofline leg: HIP events around every kernel launch, on the stream the kernels run on (rank 0) -----------
    # With profiling enabled the engine keeps the weight-gradient kernels on the same stream (no overlap), so each duration is
    # that of the kernel alone; `value` above comes from the overlapped product path.
    if rank == 0 and not args.no_roofline:
        n_prof = min(steps, 20)
        xp, yp = synth_u8(n_prof * bs, hw, ncls, 3)
        xp_d = torch.from_numpy(xp).to(device)
        ops.prof_enable(True)
        ops.prof_reset()
        agent.train_learner(xp_d, yp)
        torch.cuda.synchronize()
        cls = {}
        for i, name in enumerate(["conv_gemm", "conv_wgrad", "bn_elementwise", "head_loss", "knn_buffer"]):
            ms, n = ops.prof_query(i)
            cls[name] = dict(ms=ms, launches=n)
        ops.prof_enable(False)
        ops.prof_reset()
        # what the fp32 MFMA pipe of THIS box delivers right now (register-only instruction stream, best of 3 x ~1 ms): boxes of the
        # pool differ by up to ~19 % on the MFMA-dense kernels; `frac` stays against the nominal peak
        cal = max(ops.mfma_calibrate(20000)[0] for _ in range(3))
        gemm_fl, wgrad_fl = flops_per_step(workload, hw)
        traffic, traffic_src = pmc_traffic("conv_t_kernel") if workload == "scr" else (None, None)
        g = cls["conv_gemm"]
        wg = cls["conv_wgrad"]
        out["roofline"] = dict(
            bound="mfma", kernel="conv_t_kernel + conv_q_kernel + conv_s_kernel + conv_wx_kernel (implicit-GEMM forward + data-gradient; class PROF_CONV of ocl_prof_*)",
            achieved=(gemm_fl * n_prof / (g["ms"] * 1e-3) / 1e12) if g["ms"] > 0 else None,
            peak=PEAK_FP32_MFMA_TFLOPS, unit="TFLOP/s",
            frac=(gemm_fl * n_prof / (g["ms"] * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS) if g["ms"] > 0 else None,
            calibrated_peak=cal, frac_of_calibrated=(gemm_fl * n_prof / (g["ms"] * 1e-3) / 1e12 / cal) if g["ms"] > 0 and cal > 0 else None,
            traffic=traffic,
            traffic_source=(traffic_src + " (rocprofv3 --pmc passes of an earlier run of this command; not measured by this run)") if traffic_src else None,
            avg_launch_us=(g["ms"] * 1e3 / g["launches"]) if g["launches"] else None, launches_per_step=g["launches"] / n_prof,
            algorithmic_gflop_per_step=gemm_fl / 1e9,
            wgrad=dict(achieved=(wgrad_fl * n_prof / (wg["ms"] * 1e-3) / 1e12) if wg["ms"] > 0 else None,
                       algorithmic_gflop_per_step=wgrad_fl / 1e9, launches_per_step=wg["launches"] / n_prof,
                       avg_launch_us=(wg["ms"] * 1e3 / wg["launches"]) if wg["launches"] else None),
            # kNN / buffer path (SURVEY §8d algorithmic bytes per iteration: buffer gather + features + slot replacement [+ MIR's virtual
            # step]); latency-bound by construction, reported next to its kernel time
            knn_buffer=dict(algorithmic_bytes_per_step=KNN_BUFFER_BYTES[workload],
                            achieved_GBps=(KNN_BUFFER_BYTES[workload] / (cls["knn_buffer"]["ms"] / n_prof * 1e-3) / 1e9)
                            if cls["knn_buffer"]["ms"] > 0 else None, peak_GBps=8000.0,
                            launches_per_step=cls["knn_buffer"]["launches"] / n_prof),
            whole_step_frac=(gemm_fl + wgrad_fl) / (elapsed / steps) / 1e12 / PEAK_FP32_MFMA_TFLOPS,   # all conv flops of a step / wall time of a step
            per_step_ms={k: v["ms"] / n_prof for k, v in cls.items()},
            launches_per_step_all={k: v["launches"] / n_prof for k, v in cls.items()})
    return out


# ---- accuracy leg ("final avg accuracy" half of BASELINE.json's metric) -----------------------------------------------------
ACC_CFG = dict(n_tasks=10, classes_per_task=10, n_train=50, n_test=10, blend=0.3, seeds=3)
ACC_STREAMS = ("noise_prototype", "smooth_prototype", "texture_prototype")
# accuracy.aser: BASELINE configs[2] (ER + ASER retrieve / update) with a memory the stream FILLS early -- 500 slots on 2000 images (10 tasks x 10
# classes x 20) -- so that 150 of the 200 steps run the Shapley-valued retrieval and replacement (with configs[2]'s 5000 slots the 5000-image
# accuracy stream would never leave the uniform fill phase: aser_retrieve.py:24-26, aser_update.py:27-36)
ACC_ASER = dict(mem_size=500, n_train=20, extra_seeds=0, oracle_threads=4)   # (five seeds both sides lengthened the default run by 1.5 minutes: scripts/aser_accuracy_probe.py has them)


def aser_seeds(seeds):
    """The accuracy leg's seeds (+ ACC_ASER["extra_seeds"] more)."""
    seeds = list(seeds)
    return seeds + [seeds[-1] + 100 * (i + 1) for i in range(ACC_ASER["extra_seeds"])]


def _upsample(grid, hw):
    """Bilinear interpolation of a coarse [g, g, 3] grid to [hw, hw, 3] (a spatially smooth field)."""
    g = grid.shape[0]
    pos = (np.arange(hw) + 0.5) * g / hw - 0.5
    i0 = np.clip(np.floor(pos).astype(int), 0, g - 1)
    i1 = np.clip(i0 + 1, 0, g - 1)
    f = np.clip(pos - i0, 0.0, 1.0).astype(np.float32)
    rows = grid[i0] * (1 - f)[:, None, None] + grid[i1] * f[:, None, None]
    return rows[:, i0] * (1 - f)[None, :, None] + rows[:, i1] * f[None, :, None]


def texture_classes(n_classes):
    """Class table of the texture stream: a class is a SET of flip-symmetric plaid components (orientation theta in {0, 15, ..., 90}
    degrees rendered as the pair +theta / -theta, frequency band low = 2.6 or high = 7.5 cycles per image, waveform sine or square).
    Singles, pairs of components with different orientations and square-wave singles give 126 combinations; a fixed permutation picks
    `n_classes` of them so that every task mixes the kinds."""
    comps = [(t, b) for t in range(0, 91, 15) for b in (0, 1)]
    specs = [((t, b, 0),) for (t, b) in comps] + [((t, b, 1),) for (t, b) in comps]
    for i, (t1, b1) in enumerate(comps):
        for (t2, b2) in comps[i + 1:]:
            if t1 != t2:
                specs.append(((t1, b1, 0), (t2, b2, 0)))
    order = np.random.default_rng(4321).permutation(len(specs))
    assert len(specs) >= n_classes
    return [specs[i] for i in order[:n_classes]]


def texture_images(spec, n, hw, rng):
    """n images of one texture class: luminance only (R = G = B: hue / saturation jitter and grayscale are neutral), every component
    drawn with its own random phases, +-3 degrees of orientation and +-10 % of frequency jitter, contrast in [0.5, 1], mean 0.5 with
    +-0.04 of offset, pixel noise sigma 0.04.  What defines the class -- which orientations are present (as +-theta pairs: a horizontal
    flip maps the class to itself), in which frequency band (the bands are 2.9x apart: a RandomResizedCrop of scale >= 0.2 zooms by at most
    2.24x), with which waveform -- survives crops, flips and an affine change of the grey levels; absolute position and phase do not
    matter."""
    yy, xx = np.meshgrid(np.arange(hw, dtype=np.float32), np.arange(hw, dtype=np.float32), indexing="ij")
    img = np.zeros((n, hw, hw), dtype=np.float32)
    for (theta, band, square) in spec:
        th = np.deg2rad(theta + rng.uniform(-3, 3, n)).astype(np.float32)[:, None, None]
        f = ((2.6, 7.5)[band] * rng.uniform(0.9, 1.1, n)).astype(np.float32)[:, None, None] * (2 * np.pi / hw)
        for sign in (1.0, -1.0):
            ph = rng.uniform(0, 2 * np.pi, n).astype(np.float32)[:, None, None]
            g = np.cos(f * (xx[None] * np.cos(th) + sign * yy[None] * np.sin(th)) + ph)
            img += np.sign(g) * 0.7 if square else g
    img /= 2.0 * len(spec)
    con = rng.uniform(0.5, 1.0, n).astype(np.float32)[:, None, None]
    off = rng.uniform(-0.04, 0.04, n).astype(np.float32)[:, None, None]
    img = 0.5 + off + 0.3 * con * img + rng.normal(0, 0.04, (n, hw, hw)).astype(np.float32)
    u8 = np.clip(img * 255.0, 0, 255).astype(np.uint8)
    return np.repeat(u8[..., None], 3, axis=3)


def accuracy_stream(seed, n_tasks, classes_per_task, n_train, n_test, blend, hw=32, kind="noise_prototype"):
    """Class-incremental Split-CIFAR100-shaped stream (no datasets on disk), tasks of `classes_per_task` consecutive
    classes (general_main.py --fix_order True), a test set per task.  Two kinds of classes:
      noise_prototype   a fixed uint8 white-noise prototype per class; image = blend * prototype + (1 - blend) * white noise.  The
                        only signal is the exact pixel pattern: crops / flips / colour jitter destroy it by construction.
      smooth_prototype  a spatially smooth prototype per class (a 4x4 colour grid interpolated to hw x hw); image = 0.3 * prototype
                        + 0.7 * a per-image field of the same kind (the nuisance lives in the signal's own subspace: a nearest-class-mean
                        rule on the raw pixels reaches ~0.45 with 50 images per class) + +-16 of pixel noise.  Crops and flips of a
                        smooth field keep most of the colour layout: the stream on which an augmentation pipeline has a chance to behave.
      texture_prototype classes are luminance textures (texture_classes / texture_images): class identity = orientations, frequency
                        bands and waveform of a plaid, which random-resized crops (scale >= 0.2), horizontal flips, colour jitter and
                        grayscale leave intact BY CONSTRUCTION -- the stream on which the SCR augmentation (agents/scr.py:18-24) must
                        not hurt."""
    rng = np.random.default_rng(70000 + seed)
    tasks, tests = [], []
    tex = texture_classes(n_tasks * classes_per_task) if kind == "texture_prototype" else None
    for t in range(n_tasks):
        classes = range(t * classes_per_task, (t + 1) * classes_per_task)
        for store, n in ((tasks, n_train), (tests, n_test)):
            xs, ys = [], []
            for c in classes:
                prng = np.random.default_rng(1234 + c)
                if kind == "texture_prototype":
                    xs.append(texture_images(tex[c], n, hw, rng))
                    ys.append(np.full(n, c, dtype=np.int64))
                    continue
                if kind == "noise_prototype":
                    proto = prng.integers(0, 256, (hw, hw, 3)).astype(np.float32)
                    noise = rng.integers(0, 256, (n, hw, hw, 3)).astype(np.float32)
                    img = blend * proto[None] + (1 - blend) * noise
                else:
                    proto = _upsample(prng.integers(0, 256, (4, 4, 3)).astype(np.float32), hw)
                    field = np.stack([_upsample(rng.integers(0, 256, (4, 4, 3)).astype(np.float32), hw) for _ in range(n)])
                    img = 0.3 * proto[None] + 0.7 * field + rng.integers(-16, 17, (n, hw, hw, 3)).astype(np.float32)
                xs.append(np.clip(img, 0, 255).astype(np.uint8))
                ys.append(np.full(n, c, dtype=np.int64))
            store.append((np.concatenate(xs), np.concatenate(ys)))
    return tasks, tests


def summarise_accuracy(accs):
    """experiment/metrics.py:5-44 over the [n_run, T, T] arrays (mean and the reference's 95 % t-interval over the runs)."""
    from ocl_amd.metrics import compute_performance
    accs = np.asarray(accs)
    if accs.shape[0] > 1:
        names = ("avg_end_acc", "avg_end_fgt", "avg_acc", "avg_bwtp", "avg_fwt")
        return {k: dict(mean=float(v[0]), ci95=float(v[1])) for k, v in list(zip(names, compute_performance(accs)))[:3]}
    end = accs[0, -1, :]
    fgt = accs[0].max(axis=0) - end
    return dict(avg_end_acc=dict(mean=float(end.mean()), ci95=None), avg_end_fgt=dict(mean=float(fgt.mean()), ci95=None))


def accuracy_leg(args, rank, world, local):
    """Short SCR runs (experiment/run.py:34: independent runs, own seed each; with N ranks one run per rank and ONE all_gather of the
    [T, T] accuracy arrays -- at N = 1 the `seeds` runs follow each other), evaluate() with the NCM classifier after every task.  Per
    stream kind the product augmentation and the identity augmentation (the oracle's: kornia is absent)."""
    from ocl_amd import dist as odist
    from ocl_amd.run import single_run
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    c = ACC_CFG
    out = {}
    import ocl_amd.agents.scr as scr_mod
    seeds = [odist.run_seed(args.seed, rank) + 100 * i for i in range(c["seeds"] if world == 1 else 1)]
    for kind in ACC_STREAMS:
        res = {}
        variants = [("hip", False, {}), ("hip_identity_augmentation", True, {})]
        if kind == "texture_prototype":
            # the paper's SCR setting (config_CVPR/agent/scr/scr_5k.yml:9-10: temp 0.1 + review trick), product augmentation
            variants.append(("paper_setting", False, dict(temp=0.1, trick=dict(make_params({}).trick, review_trick=True))))
        for tag, identity, over in variants:
            runs, t_train, wall, ev_ms = [], 0.0, 0.0, []
            for seed in seeds:
                tasks, tests = accuracy_stream(seed, c["n_tasks"], c["classes_per_task"], c["n_train"], c["n_test"], c["blend"], kind=kind)
                params = make_params(dict(WORKLOADS["scr"], num_tasks=c["n_tasks"], **over))
                orig = scr_mod.ScrAugment.__call__
                if identity:
                    scr_mod.ScrAugment.__call__ = lambda self, x: x
                try:
                    t0 = time.perf_counter()
                    acc, tt, n_img, ag = single_run(params, tasks, tests, seed)
                    wall += time.perf_counter() - t0
                    t_train += tt
                    ev_ms += [1e3 * t for t in ag.evaluate_seconds]
                finally:
                    scr_mod.ScrAugment.__call__ = orig
                runs.append(acc)
            accs = np.stack(runs)
            if world > 1:
                accs, _ = odist.gather_runs(runs[0], device=device)
            res[tag] = dict(summarise_accuracy(accs), runs=int(accs.shape[0]), wall_s=wall,
                            end_acc_per_run=[float(a[-1].mean()) for a in accs])   # (per-run values feed the summary, then leave the line)
            if tag == "hip":   # evaluate() after every task: NCM over the memory + every test set seen so far
                res[tag]["evaluate_ms"] = dict(mean=float(np.mean(ev_ms)), last_task=float(np.mean(ev_ms[c["n_tasks"] - 1::c["n_tasks"]])))
        out[kind] = res
    # BASELINE configs[2] (ER + ASER retrieve / update, softmax classifier, no augmentation on either side): the texture stream, same seeds as
    # the oracle's runs
    runs, ev_ms, wall = [], [], 0.0
    for seed in aser_seeds(seeds):
        tasks, tests = accuracy_stream(seed, c["n_tasks"], c["classes_per_task"], ACC_ASER["n_train"], c["n_test"], c["blend"], kind="texture_prototype")
        params = make_params(dict(WORKLOADS["aser"], num_tasks=c["n_tasks"], mem_size=ACC_ASER["mem_size"]))
        t0 = time.perf_counter()
        acc, tt, n_img, ag = single_run(params, tasks, tests, seed)
        wall += time.perf_counter() - t0
        ev_ms += [1e3 * t for t in ag.evaluate_seconds]
        runs.append(acc)
    accs = np.stack(runs)
    if world > 1:
        accs, _ = odist.gather_runs(runs[0], device=device)
    out["aser"] = dict(hip=dict(summarise_accuracy(accs), runs=int(accs.shape[0]), wall_s=wall, end_acc_per_run=[float(a[-1].mean()) for a in accs],
                                evaluate_ms=dict(mean=float(np.mean(ev_ms)), last_task=float(np.mean(ev_ms[c["n_tasks"] - 1::c["n_tasks"]])))),
                       stream="texture_prototype, %d images per class" % ACC_ASER["n_train"],
                       config="BASELINE.json configs[2] (ER, retrieve ASER, update ASER, k 3, n_smp_cls 1.5, softmax classifier) with mem_size %d: the memory is full "
                              "after 50 of the 200 steps, the other 150 run the Shapley-valued retrieval and replacement" % ACC_ASER["mem_size"])
    out["stream"] = ("%d tasks x %d classes, %d train / %d test images per class; SCR random/random, mem_size 5000, eps_mem_batch 100, temp 0.07, "
                     "NCM classifier; %d runs, seeds = --seed + rank + 100 * run; noise_prototype: class prototype (white noise) blended %.0f%% "
                     "with white noise; smooth_prototype: smooth 4x4-grid prototype, 30%% + 70%% smooth per-image field + pixel noise; "
                     "texture_prototype: luminance plaids whose class (orientations as +-theta pairs, frequency band, waveform) is invariant "
                     "under crop >= 0.2 / flip / colour jitter / grayscale by construction (+ paper_setting: temp 0.1, review trick)"
                     % (c["n_tasks"], c["classes_per_task"], c["n_train"], c["n_test"], len(seeds), 100 * c["blend"]))
    return out, seeds


def accuracy_oracle_worker(seed, kind, threads, workload="scr"):
    """One run of the CPU oracle (identity augmentation) on the stream of (seed, kind); prints the [T, T] accuracy array as JSON."""
    from oracle import ocl_oracle as O
    import random
    c = ACC_CFG
    aser = workload == "aser"
    tasks, tests = accuracy_stream(seed, c["n_tasks"], c["classes_per_task"], ACC_ASER["n_train"] if aser else c["n_train"], c["n_test"], c["blend"], kind=kind)
    cfg = dict(WORKLOADS[workload], seed=seed, tasks=[[0]] * c["n_tasks"], n_train=0, n_test=0)
    if aser:
        cfg["mem_size"] = ACC_ASER["mem_size"]
    np.random.seed(seed)
    random.seed(seed)
    torch.manual_seed(seed)
    torch.set_num_threads(threads)
    t0 = time.perf_counter()
    oa = O.OracleAgent(cfg)
    accs = []
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):
        for (x, y) in tasks:
            oa.train_learner(x, y)
            accs.append(oa.evaluate(tests))
    print(json.dumps(dict(acc=np.array(accs).tolist(), wall_s=time.perf_counter() - t0)))


def accuracy_oracle_start(seeds, threads):
    """The same streams through the CPU oracle (identity augmentation), rank 0 at N = 1 only: one process per (stream kind, seed),
    all at once on the host cores (test infrastructure: the checker, not the thing measured).  Started AFTER every timed GPU leg and
    the cpu_baseline sample, so that the HIP accuracy runs (not timing-critical) overlap the oracle's ~5 minutes."""
    t0 = time.perf_counter()
    # every seed of the texture stream (the spread over the seeds is the yardstick for the augmentation comparison on the stream where the
    # augmentation must not hurt), the first seed of the white-noise and smooth streams (the like-for-like check of the identity-
    # augmentation runs): five concurrent runs
    todo = {"smooth_prototype": list(seeds[:1]), "noise_prototype": list(seeds[:1]), "texture_prototype": list(seeds)}
    procs = {(k, s): subprocess.Popen([sys.executable, os.path.abspath(__file__), "--oracle-accuracy-worker", str(s), k, str(threads)],
                                      stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=dict(os.environ, OMP_NUM_THREADS=str(threads)))
             for k in ACC_STREAMS for s in todo[k]}
    # ER + ASER (configs[2]) on the texture stream, every seed: three more concurrent runs
    todo["aser"] = aser_seeds(seeds)
    for s in todo["aser"]:
        ta = ACC_ASER["oracle_threads"]   # (short runs, half of them Python loops: fewer threads, less contention with the SCR oracles)
        procs[("aser", s)] = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--oracle-accuracy-worker", str(s), "texture_prototype", str(ta), "aser"],
                                              stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=dict(os.environ, OMP_NUM_THREADS=str(ta)))
    return procs, todo, t0, threads


def accuracy_oracle_finish(handle):
    procs, todo, t0, threads = handle
    res = {k: json.loads(p.communicate()[0].strip().splitlines()[-1]) for k, p in procs.items()}
    out = {}
    for kind in ACC_STREAMS + ("aser",):
        accs = np.array([res[(kind, s)]["acc"] for s in todo[kind]])
        out[kind] = dict(summarise_accuracy(accs), runs=len(todo[kind]), seeds=todo[kind], end_acc_per_run=[float(a[-1].mean()) for a in accs],
                         run_wall_s=[res[(kind, s)]["wall_s"] for s in todo[kind]])
    out.update(wall_s=time.perf_counter() - t0, threads_per_run=threads, kind="port (oracle restatement, identity augmentation)")
    return out


def cpu_leg(args):
    """The oracle restatement of the same step on the host cores (bounded sample)."""
    from oracle import ocl_oracle as O
    w = WORKLOADS[args.workload]
    cfg = dict(w, seed=args.seed, tasks=[[0]], n_train=0, n_test=0)
    torch.manual_seed(args.seed)
    np.random.seed(args.seed)
    oa = O.OracleAgent(cfg)
    hw = {"cifar10": 32, "cifar100": 32, "mini_imagenet": 84}[w["data"]]
    ncls = {"cifar10": 10, "cifar100": 100, "mini_imagenet": 100}[w["data"]]
    rng = np.random.default_rng(0)
    n_fill = w["mem_size"]
    for s in range(0, n_fill, 500):
        n = min(500, n_fill - s)
        ys = torch.from_numpy(rng.integers(0, ncls, n).astype(np.int64))
        xs = torch.from_numpy(rng.random((n, 3, hw, hw), dtype=np.float32))
        if w["update"] == "ASER":
            O.aser_update(O.OracleNet(oa.state, head=oa.head, training=True), oa.buf, oa.cache, xs, ys, oa.p)
        else:
            O.reservoir_update(oa.buf, xs, ys)
    n_warm, n_timed = 2, args.cpu_steps
    x, y = synth_u8((n_warm + 4 * 2 + n_timed) * 10, hw, ncls, 4)
    oa.train_learner(x[:n_warm * 10], y[:n_warm * 10])
    # torch's default (one thread per hardware thread) oversubscribes these small ops badly on a 256-thread host: probe a
    # few intra-op thread counts on 2 iterations each and time the bounded sample at the fastest one (`cores` reports it)
    default_threads = torch.get_num_threads()
    probes, pos = {}, n_warm * 10
    for nt in sorted(set([8, 16, 32, min(64, default_threads)])):
        torch.set_num_threads(nt)
        t0 = time.perf_counter()
        oa.train_learner(x[pos:pos + 20], y[pos:pos + 20])
        probes[nt] = (time.perf_counter() - t0) / 2
        pos += 20
    best = min(probes, key=probes.get)
    torch.set_num_threads(best)
    t0 = time.perf_counter()
    oa.train_learner(x[pos:pos + n_timed * 10], y[pos:pos + n_timed * 10])
    dt = time.perf_counter() - t0
    torch.set_num_threads(default_threads)
    return dict(value=n_timed * 10 / dt, unit="stream images/s", cores=best, kind="port",
                sample="%d iterations of the %s step (oracle restatement: torch-CPU ATen ops, the reference's own backend) "
                       "with the replay buffer full, %.1f s at %d intra-op threads (probed %s ms/step); port vs the reference's own agents on the "
                       "same 8 cores, alternating bursts: SCR 0.92x, ASER 1.08x of the reference's ms/step (profiles/r6_cpu_port_vs_reference.txt)"
                       % (n_timed, args.workload.upper(), dt, best, {k: round(v * 1e3) for k, v in probes.items()}),
                ms_per_step=dt / n_timed * 1e3)


def compact(v, sig=5):
    """Floats to `sig` significant digits (the driver stores a bounded tail of the line: every config must stay inside it)."""
    if isinstance(v, float):
        return float("%.*g" % (sig, v)) if v == v and abs(v) != float("inf") else None
    if isinstance(v, dict):
        return {k: compact(x, 7 if k in ("value", "ms_per_step") else sig) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [compact(x, sig) for x in v]
    return v


def main():
    if len(sys.argv) >= 5 and sys.argv[1] == "--oracle-accuracy-worker":
        return accuracy_oracle_worker(int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), sys.argv[5] if len(sys.argv) > 5 else "scr")
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="scr", choices=sorted(WORKLOADS))
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--cpu-steps", type=int, default=150)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the ASER / ER / MIR legs (also.*) of the default SCR run")
    ap.add_argument("--preroll", type=int, default=100,
                    help="real training steps run BEFORE the --warmup steps (reported as preroll_steps / preroll_ms; 0 = none)")
    ap.add_argument("--repeats", type=int, default=5,
                    help="the timed region (exactly --steps iterations between barriers) is run this many times back to back; "
                         "ms_per_step / value are the median repeat, every repeat is listed in ms_per_step_repeats")
    ap.add_argument("--no-accuracy", action="store_true", help="skip the accuracy leg")
    ap.add_argument("--also-steps", type=int, default=100)
    ap.add_argument("--single-stream", action="store_true",
                    help="keep the weight-gradient kernels on the main stream (OCL_SINGLE_STREAM=1): the configuration the roofline "
                         "leg measures kernels in, and the one profiles/*kernel_stats_single_stream* are taken in")
    args = ap.parse_args()
    if args.single_stream:
        os.environ["OCL_SINGLE_STREAM"] = "1"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: launch the N ranks ourselves (one process per GPU, the same command the driver uses);
        # rank 0 of the child job prints the JSON line
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus, "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))))

    import ocl_amd  # noqa: F401
    from ocl_amd import dist as odist
    rank, world, local = odist.init_from_env()
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    # every rank (also the single one) goes to the CPUs of its GPU's NUMA node before the first model is built (dist.pin_to_gpu_numa)
    pinned = odist.pin_to_gpu_numa(local, int(os.environ.get("LOCAL_WORLD_SIZE", str(world))))
    also, acc_res, acc_seeds = {}, None, None
    with contextlib.redirect_stdout(sys.stderr):   # the agents print like the reference ("buffer has N slots"): stdout carries the JSON only
        res = gpu_leg(args, rank, world, local)
        # the other three 1-GPU configurations of BASELINE.json and the accuracy leg belong to the single-GPU record (the scaling runs
        # time the headline step only)
        if args.workload == "scr" and not args.no_also and world == 1:
            for wl in ("aser", "er", "mir"):
                also[wl] = gpu_leg(args, rank, world, local, workload=wl, steps=args.also_steps, warmup=10)
    # weight-gradient time the second stream does not hide: the same timed region in a second process whose engine skips the convolution
    # weight gradients (OCL_DEBUG_SKIP_WGRAD=1, a timing-only debug switch of csrc/net.hip read once per process): step - chain-only step
    exposed = None
    if rank == 0 and world == 1 and "roofline" in res and args.workload == "scr" and not args.single_stream:
        try:
            cmd = [sys.executable, os.path.abspath(__file__), "--workload", args.workload, "--steps", str(min(args.steps, 100)), "--warmup", "5", "--repeats", "3",
                   "--no-roofline", "--no-accuracy", "--no-cpu-baseline", "--no-also", "--seed", str(args.seed)]
            r = subprocess.run(cmd, env=dict(os.environ, OCL_DEBUG_SKIP_WGRAD="1"), capture_output=True, text=True, timeout=300)
            d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
            step_ms = res["elapsed"] / args.steps * 1e3
            exposed = dict(chain_only_ms_per_step=d["ms_per_step"], exposed_ms_per_step=step_ms - d["ms_per_step"],
                           source="second process with OCL_DEBUG_SKIP_WGRAD=1 (the engine skips conv_wgrad_kernel + its reductions; timing only), same "
                                  "two-stream schedule, %d steps x 3 repeats" % min(args.steps, 100))
        except Exception as e:      # (a reported extra: its failure must not take the line with it)
            exposed = dict(error="%s: %s" % (type(e).__name__, str(e)[:160]))
        res["roofline"]["wgrad"]["exposed"] = exposed
    cpu_line, oracle_handle = None, None
    if rank == 0 and world == 1:
        with contextlib.redirect_stdout(sys.stderr):
            # the CPU legs run on the mask the process started with, not on the GPU-local slice the launch loop was pinned to
            if not args.no_cpu_baseline:
                odist.restore_affinity()
                cpu_line = cpu_leg(args)        # (before the oracle's accuracy processes take the host cores)
            if not args.no_accuracy:
                if not args.no_cpu_baseline:
                    oracle_handle = accuracy_oracle_start([odist.run_seed(args.seed, rank) + 100 * i for i in range(ACC_CFG["seeds"])], 8)
                if pinned:
                    odist.pin_to_gpu_numa(local, int(os.environ.get("LOCAL_WORLD_SIZE", str(world))))
                acc_res, acc_seeds = accuracy_leg(args, rank, world, local)
    if rank != 0:
        return
    w = WORKLOADS[args.workload]
    bs = res["bs"]
    value = res["total_steps"] * bs / res["elapsed"]
    line = {
        "metric": "replay-step images/sec (%s, Split-CIFAR100-shaped synthetic stream, Reduced-ResNet18, mem_size %d)"
                  % (args.workload.upper(), w["mem_size"]) if w["data"] != "mini_imagenet" else
                  "replay-step images/sec (%s, Split-Mini-ImageNet-shaped synthetic stream, Reduced-ResNet18, mem_size %d)"
                  % (args.workload.upper(), w["mem_size"]),
        "value": value,
        "unit": "stream images/s",
        "n_gpus": world,
        "per_rank_images_per_s": res["per_rank_images_per_s"],
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": res["elapsed"] / args.steps * 1e3,
        "ms_per_step_max": max(res["repeats_ms_per_step"]),
        "ms_per_step_repeats": res["repeats_ms_per_step"],
        "timing": "median of %d back-to-back repeats of the timed region (each EXACTLY --steps iterations, barrier + synchronize on both "
                  "sides, max over ranks); every repeat listed in ms_per_step_repeats" % len(res["repeats_ms_per_step"]),
        "preroll_steps": res["preroll_steps"], "preroll_ms": res["preroll_ms"],
        "env": res["env"],
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "BASELINE.json configs[%d]: %s" % ({"er": 0, "scr": 1, "aser": 2, "mir": 3}[args.workload],
                                                                  ", ".join("%s=%s" % kv for kv in sorted(w.items()))),
                   "stream_batch": bs, "images_through_network_per_step": {"scr": 220, "aser": 610, "er": 20, "mir": 120}[args.workload],
                   "parallelism": "%d independent stream(s), one per GPU" % world},
    }
    line["config"]["rank_placement"] = ("each rank pinned to a disjoint slice of its GPU's NUMA-local CPUs (rank 0: %d CPUs, first %d)" % (len(pinned), pinned[0])
                                        if pinned else "not pinned (OCL_PIN=0, or sysfs gave no local_cpulist)")
    if "roofline" in res:
        line["roofline"] = res["roofline"]
    if cpu_line is not None:
        line["cpu_baseline"] = cpu_line
    if also:
        names = {"aser": ("ER + ASER retrieve / update, Split-CIFAR100-shaped synthetic stream, mem_size 5000, k 3", 2, 610),
                 "er": ("ER random / random, Split-CIFAR10-shaped synthetic stream, mem_size 1000", 0, 20),
                 "mir": ("ER + MIR retrieve, Split-Mini-ImageNet-shaped 84x84 synthetic stream, mem_size 10000, subsample 50", 3, 120)}
        line["also"] = {}
        for wl, a in also.items():
            wa = WORKLOADS[wl]
            rf = a.get("roofline", {})
            line["also"][wl] = {
                "metric": "replay-step images/sec (%s)" % names[wl][0],
                "workload": "BASELINE.json configs[%d]" % names[wl][1],
                "value": a["total_steps"] * a["bs"] / a["elapsed"], "unit": "stream images/s", "steps": a["steps"],
                "ms_per_step": a["elapsed"] / a["steps"] * 1e3, "ms_per_step_max": max(a["repeats_ms_per_step"]), "ms_per_step_repeats": a["repeats_ms_per_step"],
                "images_through_network_per_step": names[wl][2], "env": {k: a["env"].get(k) for k in ("sclk_mhz", "power_w")},
                # (same kernel class, peak and unit as the headline's roofline object)
                "roofline": {k: rf.get(k) for k in ("bound", "achieved", "frac", "frac_of_calibrated", "avg_launch_us", "launches_per_step",
                                                    "algorithmic_gflop_per_step", "whole_step_frac", "wgrad", "knn_buffer", "per_step_ms",
                                                    "launches_per_step_all")} if rf else None}
    if acc_res is not None:
        if oracle_handle is not None:
            orc = accuracy_oracle_finish(oracle_handle)   # 5 concurrent runs x 8 intra-op threads, started before the HIP accuracy runs
            acc_res["cpu_oracle"] = orc
            for kind in ACC_STREAMS:
                h, hi, o = acc_res[kind]["hip"], acc_res[kind]["hip_identity_augmentation"], orc[kind]
                spread = (max(o["end_acc_per_run"]) - min(o["end_acc_per_run"])) if o["runs"] > 1 else None
                same = [hi["end_acc_per_run"][acc_seeds.index(s)] for s in o["seeds"]]     # the HIP identity runs of the oracle's seeds
                acc_res[kind]["summary"] = dict(
                    abs_diff_avg_end_acc_identity_vs_oracle=abs(float(np.mean(same)) - o["avg_end_acc"]["mean"]),
                    oracle_spread_over_seeds=spread,
                    product_minus_identity_augmentation=h["avg_end_acc"]["mean"] - hi["avg_end_acc"]["mean"])
                if kind == "texture_prototype" and spread is not None:
                    # the stream on which the augmentation must not hurt: product - identity >= -(the oracle's own spread over the seeds)
                    acc_res[kind]["summary"]["augmentation_not_harmful"] = bool(
                        acc_res[kind]["summary"]["product_minus_identity_augmentation"] >= -spread)
            if "aser" in acc_res and "aser" in orc:
                h, o = acc_res["aser"]["hip"], orc["aser"]
                acc_res["aser"]["cpu_oracle"] = dict(avg_end_acc=o["avg_end_acc"], runs=o["runs"], seeds=o["seeds"])
                acc_res["aser"]["summary"] = dict(abs_diff_avg_end_acc_vs_oracle=abs(h["avg_end_acc"]["mean"] - o["avg_end_acc"]["mean"]),
                                                  oracle_spread_over_seeds=(max(o["end_acc_per_run"]) - min(o["end_acc_per_run"])) if o["runs"] > 1 else None,
                                                  # (free-running runs diverge step by step -- fp32 rounding, then other retrievals -- so the two columns agree
                                                  # as distributions, not run by run; the per-step agreement is the co-simulation tests')
                                                  # measured yardstick (profiles/r6_aser_accuracy_chaos.txt): a one-ulp change of the initial weights moves ONE
                                                  # seed's end accuracy over 0.12-0.28 (HIP) / 0.18-0.26 (oracle); 38 HIP runs 0.2008, 20 oracle runs 0.2003
                                                  one_run_std_measured=0.04,
                                                  end_acc_per_seed=dict(hip=[round(v, 4) for v in h["end_acc_per_run"]], cpu_oracle=[round(v, 4) for v in o["end_acc_per_run"]]))
                orc.pop("aser", None)
        if "aser" in acc_res:
            acc_res["aser"]["hip"].pop("end_acc_per_run", None)
            acc_res["aser"]["hip"].pop("wall_s", None)
        # the line keeps means / intervals / the summary; per-run values and wall times have done their job
        for kind in ACC_STREAMS:
            for tag, v in acc_res[kind].items():
                if tag != "summary":
                    v.pop("end_acc_per_run", None)
                    v.pop("wall_s", None)
        if "cpu_oracle" in acc_res:
            for kind in ACC_STREAMS:
                o = acc_res["cpu_oracle"][kind]
                acc_res["cpu_oracle"][kind] = dict(avg_end_acc=o["avg_end_acc"], runs=o["runs"], seeds=o["seeds"])
        line["accuracy"] = acc_res
    print(json.dumps(compact(line), separators=(",", ":")))


if __name__ == "__main__":
    main()
