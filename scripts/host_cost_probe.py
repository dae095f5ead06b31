"""What does the HOST pay per replay step when the GPU is not the bottleneck?  (profiles/r4_host_cost.txt)

  python scripts/host_cost_probe.py [workload]

Boxes of the pool differ by ~14 % on the same tree (profiles/r4_driver_command_repro.txt vs gpurun log r4c) at the same GPU clocks:
the suspicion is the host side (CPU contention on a shared node).  This probe times short bursts of steps that fit into the launch
queue -- the host never blocks on the GPU -- so the burst time is the host's own issue cost; the difference between two burst lengths
removes the per-call set-up of train_learner.  It also times the two C entry points that issue most launches and prints the cProfile
top of the issue loop."""
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
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    params, model, agent, hw, ncls = bench.build_agent(workload, 0, device)
    bs = params.batch
    xw, yw = bench.synth_u8(60 * bs, hw, ncls, 1)
    agent.train_learner(torch.from_numpy(xw).to(device), yw)
    torch.cuda.synchronize()
    from ocl_amd import ffi
    L = ffi.lib()
    acc = {"fwd": [0.0, 0], "bwd": [0.0, 0]}
    f0, b0 = L.ocl_net_forward_segments, L.ocl_net_backward

    def fwd(*a):
        t = time.perf_counter(); r = f0(*a); acc["fwd"][0] += time.perf_counter() - t; acc["fwd"][1] += 1; return r

    def bwd(*a):
        t = time.perf_counter(); r = b0(*a); acc["bwd"][0] += time.perf_counter() - t; acc["bwd"][1] += 1; return r

    res = {}
    for k in (3, 6, 3, 6, 3, 6):
        xt, yt = bench.synth_u8(k * bs, hw, ncls, 2 + k)
        xt_d = torch.from_numpy(xt).to(device)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        agent.train_learner(xt_d, yt)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        res.setdefault(k, []).append((t1 - t0, t2 - t0))
    h3 = min(r[0] for r in res[3]); h6 = min(r[0] for r in res[6])
    g3 = min(r[1] for r in res[3]); g6 = min(r[1] for r in res[6])
    print("%s: host issue of a 3-step burst %.3f ms, 6-step burst %.3f ms -> %.3f ms of pure host time per step (GPU: %.3f ms per step)"
          % (workload, h3 * 1e3, h6 * 1e3, (h6 - h3) / 3 * 1e3, (g6 - g3) / 3 * 1e3), flush=True)
    L.ocl_net_forward_segments, L.ocl_net_backward = fwd, bwd
    try:
        xt, yt = bench.synth_u8(6 * bs, hw, ncls, 99)
        xt_d = torch.from_numpy(xt).to(device)
        torch.cuda.synchronize()
        agent.train_learner(xt_d, yt)
        torch.cuda.synchronize()
    finally:
        L.ocl_net_forward_segments, L.ocl_net_backward = f0, b0
    for k, (t, n) in acc.items():
        print("  C entry %s: %d calls in the 6-step burst, %.1f us per call" % (k, n, t / max(n, 1) * 1e6))
    xt, yt = bench.synth_u8(6 * bs, hw, ncls, 77)
    xt_d = torch.from_numpy(xt).to(device)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    agent.train_learner(xt_d, yt)
    pr.disable()
    torch.cuda.synchronize()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
    print("\n".join(l[:160] for l in s.getvalue().splitlines()[4:40]))
    print("(cProfile over 6 steps)")


if __name__ == "__main__":
    if os.environ.get("OCL_PROBE_STREAM") == "1":   # (hipGraph capture, OCL_GRAPH=1, needs a non-default stream)
        with torch.cuda.stream(torch.cuda.Stream()):
            main()
    else:
        main()
