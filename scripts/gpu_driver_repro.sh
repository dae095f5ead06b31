# The driver's exact bench command on a fresh lease, next to the builder's usual one, with the GPU's clocks and power logged
# before / during / after.  gpurun --timeout 1200 -- 'bash scripts/gpu_driver_repro.sh r4a'
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
T=${1:-repro}
O=gpurun_out/${T}_driver_command_repro.txt
smi() { echo "--- rocm-smi ($1) $(date +%T.%N)"; rocm-smi --showclocks --showpower --showperflevel --showmaxpower 2>&1 | grep -v '^=*$' | head -40; }
{
  echo "### box"; rocm-smi --showproductname 2>&1 | grep -i -E 'series|model|sku|gfx' | head -5; nproc
  for f in /sys/class/drm/card*/device/pp_dpm_sclk /sys/class/drm/card*/device/pp_dpm_mclk /sys/class/drm/card*/device/power_dpm_force_performance_level; do echo "$f:"; cat $f 2>&1 | head -12; done
  ls /sys/class/drm/card*/device/hwmon/hwmon*/ 2>&1 | head -40
  smi "before anything"
} > $O 2>&1
# background sampler: sysfs current sclk + power every 50 ms
( while true; do
    s=$(grep -h '\*' /sys/class/drm/card*/device/pp_dpm_sclk 2>/dev/null | tr '\n' ' ')
    m=$(grep -h '\*' /sys/class/drm/card*/device/pp_dpm_mclk 2>/dev/null | tr '\n' ' ')
    p=$(cat /sys/class/drm/card*/device/hwmon/hwmon*/power1_average 2>/dev/null | tr '\n' ' ')
    f=$(cat /sys/class/drm/card*/device/hwmon/hwmon*/freq1_input 2>/dev/null | tr '\n' ' ')
    echo "$(date +%s.%N) sclk[$s] mclk[$m] power_uW[$p] freq1[$f]"
    sleep 0.05
  done ) > gpurun_out/${T}_clock_samples.txt 2>&1 &
SAMPLER=$!
Q="--no-cpu-baseline --no-accuracy"
run() { echo "### $(date +%T.%N) python bench.py $*" >> $O; python bench.py "$@" 2>gpurun_out/${T}_last.err | python -c "
import json,sys
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    r=d.get('roofline',{}); a=d.get('also',{}).get('aser',{})
    print(json.dumps(dict(ms_per_step=d['ms_per_step'], value=d['value'], steps=d['steps'], warmup=d['warmup'], conv_frac=r.get('frac'), per_step_ms=r.get('per_step_ms'), aser_ms=a.get('ms_per_step'), env=d.get('env'), repeats=d.get('ms_per_step_repeats'), preroll_ms=d.get('preroll_ms'))))
" >> $O; echo "### end $(date +%T.%N)" >> $O; smi "after run" >> $O 2>&1; }
run --gpus 1 --steps 20 --warmup 5 $Q
run --gpus 1 --steps 20 --warmup 5 $Q
run --gpus 1 --steps 20 --warmup 5 $Q
run --gpus 1 --steps 200 --warmup 20 $Q
run --gpus 1 --steps 20 --warmup 5 $Q --no-also
run --gpus 1 --steps 200 --warmup 20 $Q --no-also
run --gpus 1 --steps 20 --warmup 5 --workload er $Q
run --gpus 1 --steps 200 --warmup 20 --workload er $Q
kill $SAMPLER
# summarise the samples: distribution of sclk while the power says the GPU is busy
python - "$T" >> $O <<'PY'
import re, sys, collections
T = sys.argv[1]
rows = [l for l in open("gpurun_out/%s_clock_samples.txt" % T)]
print("### %d clock samples; first 3 / last 3" % len(rows)); [print(r.strip()) for r in rows[:3] + rows[-3:]]
c = collections.Counter()
for r in rows:
    m = re.search(r"sclk\[([^\]]*)\]", r)
    c[m.group(1).strip() if m else "?"] += 1
print("sclk level histogram:", dict(c.most_common(12)))
PY
tail -c 3000 $O
