"""SupCon loss + gradient (ocl_supcon_fwd_bwd: supcon_rows + supcon_grad) at the SCR step's size, timed alone: HIP events around a burst of calls.
OCL_LIB=<other libocl_hip.so> runs the same against another build (A/B on one box: scripts/gpu_r6bc.sh)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocl_amd import ops  # noqa: E402

torch.manual_seed(0)
A, dim = 220, 128
f = torch.nn.functional.normalize(torch.randn(A, dim, device="cuda"), dim=1)
y = torch.randint(0, 100, (A // 2,), device="cuda")
for _ in range(20):
    loss, df = ops.supcon(f, y, 2, 0.07)
torch.cuda.synchronize()
best = 1e9
for rep in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        loss, df = ops.supcon(f, y, 2, 0.07)
    e1.record()
    torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 200 * 1e3)
print("supcon A=%d dim=%d: %.2f us per call (both launches, best of 5 bursts of 200); loss %.7f |df| %.7f lib=%s"
      % (A, dim, best, float(loss), float(df.abs().sum()), os.environ.get("OCL_LIB", "tree")))
