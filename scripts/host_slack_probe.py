"""Is the SCR step host-bound?  Add X us of host busy-wait per step: a GPU-bound loop absorbs it, a host-bound one slows down by X."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

w = sys.argv[1] if len(sys.argv) > 1 else "scr"
n = 200
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
params, model, agent, hw, ncls = bench.build_agent(w, 0, dev)
xw, yw = bench.synth_u8(300, hw, ncls, 1)
agent.train_learner(torch.from_numpy(xw).to(dev), yw)
torch.cuda.synchronize()
orig = agent.buffer.update
spin_us = [0]


def slow_update(*a, **k):
    t = time.perf_counter() + spin_us[0] * 1e-6
    while time.perf_counter() < t:
        pass
    return orig(*a, **k)


agent.buffer.update = slow_update
um = agent.buffer.update_method
if hasattr(um, "update_begin"):   # the pipelined ASER loop bypasses buffer.update (agents/exp_replay.py): slow its entry point as well
    orig_begin = um.update_begin

    def slow_begin(*a, **k):
        t = time.perf_counter() + spin_us[0] * 1e-6
        while time.perf_counter() < t:
            pass
        return orig_begin(*a, **k)
    um.update_begin = slow_begin
for us in (0, 200, 400, 800, 0):
    spin_us[0] = us
    x, y = bench.synth_u8(n * 10, hw, ncls, 2)
    xd = torch.from_numpy(x).to(dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    agent.train_learner(xd, y)
    torch.cuda.synchronize()
    print("%s: +%4d us host spin per step -> %.3f ms/step" % (w, us, (time.perf_counter() - t0) / n * 1e3))
