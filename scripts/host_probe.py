"""Host-side probe of one workload's training loop: is a step bound by the host (Python + launch issue) or by the GPU?

  python scripts/host_probe.py [workload] [steps]

Prints (a) wall time per step with a synchronise only at the end, (b) the host's own time to issue the same steps (time until
`train_learner` returns, before the final synchronise) and the GPU backlog left at that moment, (c) a cProfile of the issue loop
(top entries by own time).  No oracle, no reference: it only drives the product path."""
import cProfile
import io
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "scr"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    params, model, agent, hw, ncls = bench.build_agent(workload, 0, device)
    bs = params.batch
    xw, yw = bench.synth_u8(20 * bs, hw, ncls, 1)
    agent.train_learner(torch.from_numpy(xw).to(device), yw)
    torch.cuda.synchronize()
    for rep in range(2):
        xt, yt = bench.synth_u8(steps * bs, hw, ncls, 2 + rep)
        xt_d = torch.from_numpy(xt).to(device)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        agent.train_learner(xt_d, yt)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("%s rep %d: wall %.3f ms/step; host issue %.3f ms/step; GPU backlog when the host finished %.3f ms (total)"
              % (workload, rep, (t2 - t0) / steps * 1e3, (t1 - t0) / steps * 1e3, (t2 - t1) * 1e3), flush=True)
    xt, yt = bench.synth_u8(steps * bs, hw, ncls, 9)
    xt_d = torch.from_numpy(xt).to(device)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    agent.train_learner(xt_d, yt)
    pr.disable()
    torch.cuda.synchronize()
    for key in ("tottime", "cumtime"):
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats(key).print_stats(28)
        txt = s.getvalue()
        print("\n".join(l[:170] for l in txt.splitlines()[4:48]))
    print("(cProfile over %d steps; divide by %d for per-step)" % (steps, steps))


if __name__ == "__main__":
    main()
