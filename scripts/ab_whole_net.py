"""Whole-network A/B of planner switches: runs tests/test_gpu_ring.py's network script once per environment setting (each in its own
interpreter: the switches are read once per process) and prints, per pair of runs, the tensors that differ most.
   python scripts/ab_whole_net.py "OCL_CONV_S=0" "OCL_CONV_S=1" "OCL_CONV_S=1" ..."""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
src = open(os.path.join(ROOT, "tests", "test_gpu_ring.py")).read()
script = src[src.index('_SCRIPT_S = r"""') + len('_SCRIPT_S = r"""'):]
script = script[:script.index('"""')] % {"root": ROOT}
runs = []
tmp = tempfile.mkdtemp()
for i, e in enumerate(sys.argv[1:]):
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    for kv in e.split():
        if "=" in kv:
            k, v = kv.split("=", 1); env[k] = v
    f = os.path.join(tmp, "r%d.npz" % i)
    r = subprocess.run([sys.executable, "-c", script, f], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    runs.append((e, dict(np.load(f))))
for i in range(len(runs)):
    for j in range(i + 1, len(runs)):
        a, b = runs[i][1], runs[j][1]
        errs = sorted(((float(np.abs(a[k] - b[k]).max() / (1e-12 + np.abs(a[k]).max())), k) for k in a), reverse=True)
        print("[%s] vs [%s]: " % (runs[i][0], runs[j][0]) + "; ".join("%s %.2e" % (k, e) for e, k in errs[:4]))
