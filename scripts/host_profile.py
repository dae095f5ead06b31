"""cProfile of the host side of the replay step on the GPU box (where does the Python time go?)."""
import cProfile
import pstats
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

w = sys.argv[1] if len(sys.argv) > 1 else "scr"
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
params, model, agent, hw, ncls = bench.build_agent(w, 0, dev)
xw, yw = bench.synth_u8(100, hw, ncls, 1)
agent.train_learner(torch.from_numpy(xw).to(dev), yw)
torch.cuda.synchronize()
x, y = bench.synth_u8(500, hw, ncls, 2)
xd = torch.from_numpy(x).to(dev)
pr = cProfile.Profile()
import time
t0 = time.perf_counter()
pr.enable()
agent.train_learner(xd, y)
pr.disable()
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("50 steps: host-side enqueue %.1f ms/step, with final sync %.1f ms/step" % (t_host / 50 * 1e3, t_all / 50 * 1e3))
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
