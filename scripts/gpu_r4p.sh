# round 4: does the process's CPU placement explain the slow leases?  allowed CPUs, the GPU's local CPUs, bench with and without pinning.   gpurun --timeout 600 -- 'bash scripts/gpu_r4p.sh r4p'
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
T=${1:-r4p}
python - <<'PY' > gpurun_out/${T}_placement.txt 2>&1
import os, torch, sys
sys.path.insert(0, os.getcwd())
import ocl_amd
from ocl_amd import dist as odist
print("allowed CPUs:", len(os.sched_getaffinity(0)), sorted(os.sched_getaffinity(0))[:4], "...", sorted(os.sched_getaffinity(0))[-4:])
d = odist._pci_dir(0)
print("GPU pci dir:", d)
for f in ("numa_node", "local_cpulist"):
    print(f, (odist._read(os.path.join(d, f)) or "").strip())
import glob
for n in sorted(glob.glob("/sys/devices/system/node/node*")):
    print(os.path.basename(n), (odist._read(n + "/cpulist") or "").strip())
print("this process runs on CPU", os.sched_getcpu() if hasattr(os, "sched_getcpu") else "?")
print("pinned to", (lambda m: (len(m), m[0], m[-1]) if m else None)(odist.pin_to_gpu_numa(0, 1)))
PY
cat gpurun_out/${T}_placement.txt
Q="--no-cpu-baseline --no-accuracy --no-also"
line() { python -c "
import json,sys
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    r=d.get('roofline',{})
    print('$1', 'ms %.4f' % d['ms_per_step'], [round(x,4) for x in d['ms_per_step_repeats']], {k: round(v,4) for k,v in (r.get('per_step_ms') or {}).items()}, d['config'].get('rank_placement','')[:60])
"; }
for wl in scr er; do for v in 0 1 0 1; do
  OCL_PIN=$v timeout 200 python bench.py --workload $wl --steps 200 --warmup 20 --repeats 3 $Q 2>gpurun_out/${T}_err.log | line "$wl pin=$v"
done; done 2>&1 | tee gpurun_out/${T}_ab.txt
# the far node on purpose (if there is one): taskset to the CPUs that are NOT local to the GPU
python - <<'PY' > /tmp/far.txt
import os, sys
sys.path.insert(0, os.getcwd())
import torch, ocl_amd
from ocl_amd import dist as odist
d = odist._pci_dir(0)
txt = odist._read(os.path.join(d, "local_cpulist")) or ""
loc = set()
for part in txt.strip().split(","):
    if part:
        a, _, b = part.partition("-"); loc |= set(range(int(a), int(b or a) + 1))
far = sorted(set(os.sched_getaffinity(0)) - loc)
print(",".join(map(str, far[:32])))
PY
FAR=$(cat /tmp/far.txt)
if [ -n "$FAR" ]; then for wl in scr er; do
  OCL_PIN=0 timeout 200 taskset -c $FAR python bench.py --workload $wl --steps 200 --warmup 20 --repeats 3 $Q 2>gpurun_out/${T}_err.log | line "$wl taskset to the far node"
done; else echo "no far CPUs in this container's set"; fi 2>&1 | tee -a gpurun_out/${T}_ab.txt
