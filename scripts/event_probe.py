"""How long does Event.synchronize()/query() take on an event that completed long ago (never observed by a sync) while the stream
is busy with later work?"""
import time
import torch
dev = torch.device("cuda", 0)
a = torch.randn(8192, 8192, device=dev)
small = torch.zeros(16, device=dev)
pin = torch.empty(1024, dtype=torch.uint8).pin_memory()
torch.cuda.synchronize()
for timing in (False, True):
    for mode in ("after_kernel", "after_h2d"):
        for _ in range(3):
            b = a @ a
        torch.cuda.synchronize()
        ev = torch.cuda.Event(enable_timing=timing)
        if mode == "after_kernel":
            small.add_(1)
        else:
            d = pin.to(dev, non_blocking=True)
        ev.record()
        time.sleep(0.05)       # the GPU finishes the tiny op; no HIP call observes it
        t0 = time.perf_counter()
        for _ in range(20):
            b = a @ a          # ~140 ms of queued work behind it
        t1 = time.perf_counter()
        q = ev.query()
        t2 = time.perf_counter()
        ev.synchronize()
        t3 = time.perf_counter()
        torch.cuda.synchronize()
        t4 = time.perf_counter()
        print("timing=%s %s: enqueue %.2f ms, query()=%s %.3f ms, synchronize() %.3f ms, drain %.1f ms" % (
            timing, mode, (t1 - t0) * 1e3, q, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3))
