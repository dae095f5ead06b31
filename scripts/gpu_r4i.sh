# round 4, call 9: weight-gradient planner sweep in kbench (pixel tile, split target), MIR side-extra threshold by pixels.   gpurun --timeout 900 -- 'bash scripts/gpu_r4i.sh r4i'
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
T=${1:-r4i}
K=online-continual-learning_amd/csrc/kbench
{
for kp in 128 64; do for tgt in 256 512 768 1024; do
  echo "## OCL_WGRAD_KP=$kp OCL_WGRAD_TARGET=$tgt"
  OCL_WGRAD_KP=$kp OCL_WGRAD_TARGET=$tgt timeout 60 $K 220 2 32 wgrad 2>&1 | grep " wgrad " | cut -c1-260
done; done
echo "## OCL_WGRAD_KP=128 OCL_WGRAD_TARGET=512 OCL_WGRAD_ENOUGH=256"
OCL_WGRAD_ENOUGH=256 timeout 60 $K 220 2 32 wgrad 2>&1 | grep " wgrad " | cut -c1-260
} > gpurun_out/${T}_wgrad_sweep.txt 2>&1
python - "$T" <<'PY' | tee gpurun_out/${T}_wgrad_sweep_summary.txt
import re, sys
T = sys.argv[1]
cur = None; res = {}; order = []
for l in open("gpurun_out/%s_wgrad_sweep.txt" % T):
    if l.startswith("##"):
        cur = l.strip()[3:]; res[cur] = {}; order.append(cur); continue
    us = re.findall(r"([0-9.]+) us", l)
    if cur and len(us) >= 2:
        res[cur][l.split()[0]] = (float(us[0]), float(us[1]), "MISMATCH" in l)
names = list(res[order[0]].keys()) if order else []
print("%-22s" % "layer" + "".join("%16s" % k.replace("OCL_WGRAD_", "").replace("TARGET", "T")[:15] for k in order))
for n in names:
    print("%-22s" % n + "".join("%16s" % ("%.1f+%.1f%s" % (res[k][n][0], res[k][n][1], "!" if res[k][n][2] else "") if n in res[k] else "-") for k in order))
print("%-22s" % "sum wgrad+reduce" + "".join("%16.1f" % sum(a + b for a, b, _ in res[k].values()) for k in order))
PY
Q="--no-cpu-baseline --no-accuracy --no-also --no-roofline"
line() { python -c "
import json,sys
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('$1', 'ms %.4f' % d['ms_per_step'], [round(x,4) for x in d['ms_per_step_repeats']], d.get('env',{}).get('sclk_mhz'))
"; }
for v in -1 98304 -1 98304; do
  if [ $v = -1 ]; then timeout 200 python bench.py --workload mir --steps 200 --warmup 20 --repeats 3 $Q 2>gpurun_out/${T}_err.log | line "mir side-extra by images (default)"; else OCL_SIDE_EXTRA_MIN_PIX=$v timeout 200 python bench.py --workload mir --steps 200 --warmup 20 --repeats 3 $Q 2>gpurun_out/${T}_err.log | line "mir side-extra from $v pixels"; fi
done 2>&1 | tee gpurun_out/${T}_mir_ab.txt
