"""CPU-only: does ClassBalancedRandomSampling.draw_fast get slower as update_cache churns the class sets (the ASER step drifts
2.82 -> 3.00 ms over 1500 steps, profiles/r6_aser_drift_probe.txt)?  5000 slots / 100 classes, 8 slots moved per step (as an ASER update does),
three draws per step (plain, with 100 exclusions, plain); prints per 100 steps the time per draw and the hash-table sizes of the sets.

    python scripts/cbrs_drift_cpu.py > profiles/r6_cbrs_drift_cpu.txt"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib
pkg = importlib.import_module("ocl_amd") if importlib.util.find_spec("ocl_amd") else None
from ocl_amd.plugins import buffer_utils as B   # noqa: E402


def table_size(s):
    # sys.getsizeof(set) = header + table bytes (16 per entry) once the table left the inline small table
    return max(8, (sys.getsizeof(s) - 216) // 16)


def main():
    C = B.ClassBalancedRandomSampling
    rng = np.random.default_rng(0)
    labels = rng.integers(0, 100, 5000).astype(np.int64)
    C.class_index_cache = None
    C.update_cache(labels, 100)
    C.verify_every = int(os.environ.get("OCL_CBRS_VERIFY_EVERY", "64"))   # (1 = verify every draw: the behaviour before round 6)
    print("# verify_every = %d" % C.verify_every)
    torch.manual_seed(0)
    for rep in range(16):
        t_plain = t_excl = 0.0
        for step in range(100):
            t = time.perf_counter(); cand = C.draw_fast(1); t_plain += time.perf_counter() - t
            ex = set(cand.tolist())
            t = time.perf_counter(); C.draw_fast(1, ex); t_excl += time.perf_counter() - t
            t = time.perf_counter(); C.draw_fast(1); t_plain += time.perf_counter() - t
            slots = rng.choice(5000, 8, replace=False)
            new = rng.integers(0, 100, 8).astype(np.int64)
            C.update_cache(labels, 100, new_y=new, ind=slots.tolist())
            labels[slots] = new
        sizes = [table_size(s) for s in C.class_index_cache.values()]
        used = [len(s) for s in C.class_index_cache.values()]
        print("steps %4d: plain draw %.1f us, draw with exclusions %.1f us | set tables: mean %.0f entries (max %d) for %.0f members on average"
              % ((rep + 1) * 100, t_plain / 200 * 1e6, t_excl / 100 * 1e6, np.mean(sizes), max(sizes), np.mean(used)))


if __name__ == "__main__":
    main()
