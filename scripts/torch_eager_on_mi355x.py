"""TEST INFRASTRUCTURE -- a BASELINE, NOT THE TARGET and not the product path.

The oracle's plain-torch restatement of the replay step (oracle/ocl_oracle.py: scr_step / er_step / aser_er_step, the same ATen calls as the
reference's agents) run by stock PyTorch-ROCm EAGER on the MI355X: tensors moved to cuda:0, nothing else changed (MIOpen / rocBLAS kernels,
one launch per ATen op, autograd tape).  It answers "what does the reference get on this node if one only adds .cuda()" -- the same-node
comparator VERDICT r5 asked for next to the CPU baseline.  Same workloads, buffer full, batch 10, as bench.py.

    python scripts/torch_eager_on_mi355x.py [--steps 100] [--repeats 3] > profiles/r6_torch_eager_on_mi355x.txt"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ocl_oracle as O          # noqa: E402

WORKLOADS = {
    "scr": dict(agent="SCR", retrieve="random", update="random", data="cifar100", mem_size=5000, eps_mem_batch=100, temp=0.07, head="mlp"),
    "er": dict(agent="ER", retrieve="random", update="random", data="cifar100", mem_size=5000, eps_mem_batch=10),
    "aser": dict(agent="ER", retrieve="ASER", update="ASER", data="cifar100", mem_size=5000, eps_mem_batch=10, k=3, n_smp_cls=1.5, aser_type="asvm"),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--repeats", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workloads", default="scr,er,aser")
    args = ap.parse_args()
    dev = torch.device(os.environ.get("OCL_EAGER_DEVICE", "cuda:0"))      # (cpu: a dry run of this script where there is no GPU)
    print("# BASELINE, NOT TARGET: the oracle's torch code under stock PyTorch-ROCm eager on %s (torch %s), batch 10, memory of 5000 slots full"
          % (torch.cuda.get_device_name(0) if dev.type == "cuda" else "cpu", torch.__version__))
    out = {}
    for name in args.workloads.split(","):
        w = WORKLOADS[name]
        cfg = dict(w, seed=0, tasks=[[0]], n_train=0, n_test=0)
        torch.manual_seed(0); np.random.seed(0)
        oa = O.OracleAgent(cfg)
        for k in list(oa.state):
            t = oa.state[k].detach().to(dev)
            oa.state[k] = t.requires_grad_(True) if k in oa.names else t
        rng = np.random.default_rng(0)
        try:
            if w["update"] == "ASER":     # the class cache is filled by the plugin itself (CPU), then the memory moves to the device
                for s in range(0, 5000, 500):
                    cpu_state = {k: v.detach().cpu() for k, v in oa.state.items()}
                    O.aser_update(O.OracleNet(cpu_state, head=None, training=True), oa.buf, oa.cache,
                                  torch.from_numpy(rng.random((500, 3, 32, 32), dtype=np.float32)), torch.from_numpy(rng.integers(0, 100, 500).astype(np.int64)), oa.p)
            else:
                oa.buf.img[:] = torch.from_numpy(rng.random((5000, 3, 32, 32), dtype=np.float32))
                oa.buf.label[:] = torch.from_numpy(rng.integers(0, 100, 5000).astype(np.int64))
                oa.buf.current_index, oa.buf.n_seen_so_far = 5000, 20000
            oa.buf.img, oa.buf.label = oa.buf.img.to(dev), oa.buf.label.to(dev)
            n = (args.warmup + args.steps) * 10
            x = torch.from_numpy(rng.random((n, 3, 32, 32), dtype=np.float32)).to(dev)
            y = torch.from_numpy(rng.integers(0, 100, n).astype(np.int64)).to(dev)

            def step(i):
                bx, by = x[i * 10:(i + 1) * 10], y[i * 10:(i + 1) * 10]
                if name == "scr":
                    O.scr_step(oa.state, oa.names, oa.buf, bx, by, oa.p)
                elif name == "er":
                    O.er_step(oa.state, oa.names, oa.buf, bx, by, oa.p, "random")
                else:
                    O.aser_er_step(oa.state, oa.names, oa.buf, oa.cache, bx, by, oa.p)
            for i in range(args.warmup):
                step(i)
            reps = []
            for r in range(args.repeats):
                if dev.type == "cuda":
                    torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(args.steps):
                    step(args.warmup + i)
                if dev.type == "cuda":
                    torch.cuda.synchronize()
                reps.append((time.perf_counter() - t0) / args.steps * 1e3)
            out[name] = dict(ms_per_step=[round(v, 3) for v in reps], median_ms=round(float(np.median(reps)), 3),
                             images_per_s=round(10 / (float(np.median(reps)) * 1e-3), 1))
            print("%-5s torch eager on the GPU: %s ms/step (median %.3f) = %.0f stream images/s" % (name, out[name]["ms_per_step"], out[name]["median_ms"], out[name]["images_per_s"]))
        except Exception as e:      # the oracle is CPU code: a plugin that reaches for numpy on a device tensor is reported, not patched
            out[name] = dict(error="%s: %s" % (type(e).__name__, str(e)[:200]))
            print("%-5s does not run on the device unchanged: %s" % (name, out[name]["error"]))
        sys.stdout.flush()
    print("JSON " + json.dumps(out))


if __name__ == "__main__":
    main()
