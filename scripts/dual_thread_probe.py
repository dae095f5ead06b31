"""Upper bound on what interleaving two independent kernel chains on two HIP streams can recover: two SCR agents in one process,
one thread + one stream each, against one agent alone."""
import os
import sys
import threading
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
import ocl_amd  # noqa: F401
from ocl_amd import ops
tl = threading.local()
Ring = ops._PinnedRing


class TLRing(object):
    def upload(self, t, device):
        if not hasattr(tl, "ring"):
            tl.ring = Ring()
        return tl.ring.upload(t, device)


ops._ring = TLRing()
n = 300
agents = []
for i in range(2):
    params, model, agent, hw, ncls = bench.build_agent("scr", i, dev)
    xw, yw = bench.synth_u8(300, hw, ncls, 1 + i)
    agent.train_learner(torch.from_numpy(xw).to(dev), yw)
    x, y = bench.synth_u8(n * 10, hw, ncls, 5 + i)
    agents.append((agent, torch.from_numpy(x).to(dev), y))
torch.cuda.synchronize()


def run(k, stream):
    torch.cuda.set_device(dev)
    with torch.cuda.stream(stream):
        agents[k][0].train_learner(agents[k][1], agents[k][2])
        stream.synchronize()


t0 = time.perf_counter()
run(0, torch.cuda.current_stream())
t1 = time.perf_counter() - t0
print("one agent: %.3f ms/step (%.0f img/s)" % (t1 / n * 1e3, n * 10 / t1))
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
for rep in range(2):
    th = [threading.Thread(target=run, args=(k, streams[k])) for k in range(2)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    t2 = time.perf_counter() - t0
    print("two agents, two threads/streams: %.3f ms per pair of steps (%.0f img/s total, x%.2f)" % (t2 / n * 1e3, 2 * n * 10 / t2, (2 * n * 10 / t2) / (n * 10 / t1)))
