"""Names the one-off stall of the ASER bench leg (the 5th timed repeat is +0.8 - 1.0 ms per step = one ~85 ms event).

Runs bench.py's ASER workload exactly as its gpu_leg does (same seeds, same pre-roll / warm-up / 5 x 100 timed steps on the same batches),
but with the agent's loop cut into single iterations, each bracketed by torch.cuda.synchronize(), and with OCL_LOG_PLANS=1: the engine
prints a line per plan set it builds (a first-seen batch shape) and per plan-table arena chunk it allocates.  Prints the slow steps (> 3 x the
median) of every repeat with the engine lines that fell into them, torch's allocator counters before / after, and the eval-set sizes seen.

    OCL_LOG_PLANS=1 python scripts/aser_stall_probe.py [--workload aser] > profiles/r6_aser_stall_probe.txt 2>&1"""
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
    ap.add_argument("--workload", default="aser")
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--repeats", type=int, default=5)
    ap.add_argument("--preroll", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    params, model, agent, hw, ncls = bench.build_agent(args.workload, args.seed, dev)
    bs = params.batch
    xw, yw = bench.synth_u8(max(1, args.warmup) * bs, hw, ncls, 1)
    xt, yt = bench.synth_u8(args.steps * bs, hw, ncls, 2)
    xp, yp = bench.synth_u8(args.preroll * bs, hw, ncls, 5)
    to = lambda a: torch.from_numpy(a).to(dev)
    xw_d, xt_d, xp_d = to(xw), to(xt), to(xp)

    # one marker per iteration: the update plugin's update_finish (the loop's one synchronisation point, agents/exp_replay.py) -- or, for
    # update plugins without it, buffer.update -- is wrapped; the interval between two markers is one step (pipelined by one)
    marks = []
    upd = agent.buffer.update_method
    name = "update_finish" if hasattr(upd, "update_finish") else "update"
    inner = getattr(upd, name)

    def hooked(*a, **k):
        r = inner(*a, **k)
        marks.append(time.perf_counter())
        sys.stderr.flush()
        print("[probe] step %d done" % len(marks), file=sys.stderr)
        return r
    setattr(upd, name, hooked)

    # Python's cyclic collector: every collection with its generation and duration (a full collection walks every container alive)
    import gc
    gc_t = {}

    def gc_cb(phase, info):
        if phase == "start":
            gc_t["t"] = time.perf_counter()
        else:
            dt = (time.perf_counter() - gc_t.get("t", time.perf_counter())) * 1e3
            if dt > 1.0 or info["generation"] == 2:
                print("[probe] gc generation %d: %.2f ms, collected %d (after step %d)" % (info["generation"], dt, info["collected"], len(marks)), file=sys.stderr)
    gc.callbacks.append(gc_cb)

    def run(x_d, y, tag):
        del marks[:]
        print("[probe] %s begins" % tag, file=sys.stderr)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        agent.train_learner(x_d, y)
        torch.cuda.synchronize()
        ts = np.diff(np.array([t0] + marks)) * 1e3
        return ts, (time.perf_counter() - t0) * 1e3

    run(xp_d, yp, "preroll")
    run(xw_d, yw, "warmup")
    print("torch allocator after warm-up: %d segments, %.1f MB reserved" % (torch.cuda.memory_stats(dev)["segment.all.current"], torch.cuda.memory_reserved(dev) / 2**20))
    for r in range(args.repeats):
        seg0 = torch.cuda.memory_stats(dev)["segment.all.current"]
        ts, total = run(xt_d, yt, "repeat%d" % r)
        med = float(np.median(ts))
        slow = [(i, float(t)) for i, t in enumerate(ts) if t > 3 * med]
        print("repeat %d: %.3f ms per step over the call (%d markers: median interval %.3f ms, max %.3f ms); intervals over 3 x median: %s; torch segments %d -> %d"
              % (r, total / args.steps, len(ts), med, float(ts.max()), [(i, round(t, 2)) for i, t in slow], seg0, torch.cuda.memory_stats(dev)["segment.all.current"]))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
