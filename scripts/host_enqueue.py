"""Host-side enqueue time per step vs wall time per step (is the GPU ever waiting for Python?)."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

w = sys.argv[1] if len(sys.argv) > 1 else "scr"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
params, model, agent, hw, ncls = bench.build_agent(w, 0, dev)
xw, yw = bench.synth_u8(200, hw, ncls, 1)
agent.train_learner(torch.from_numpy(xw).to(dev), yw)
torch.cuda.synchronize()
for rep in range(3):
    x, y = bench.synth_u8(n * 10, hw, ncls, 2 + rep)
    xd = torch.from_numpy(x).to(dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    agent.train_learner(xd, y)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print("%s %d steps: host-side enqueue %.3f ms/step, with final sync %.3f ms/step" % (w, n, t_host / n * 1e3, t_all / n * 1e3))
