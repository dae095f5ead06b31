# Round 6: pixel-split target of the layers inside the merged weight-gradient launch (OCL_WGRAD_MULTI_TARGET: 0 = 512 as a launch of its own)
# gpurun --timeout 1500 -- 'bash scripts/gpu_r6u.sh r6u'
T=${1:-r6u}
mkdir -p gpurun_out
N=online-continual-learning_amd/csrc/netcheck
{
  for cfg in "20 2 32 0" "10 1 32 0"; do
    for t in 0 192 128 96 64 48 32; do
      echo "### netcheck $cfg OCL_WGRAD_MULTI_TARGET=$t"
      OCL_WGRAD_MULTI_TARGET=$t timeout 60 $N $cfg write /tmp/ref.bin | head -1
    done
  done
} > gpurun_out/${T}_netcheck.txt 2>&1
cat gpurun_out/${T}_netcheck.txt
for wl in er; do
  for t in 0 192 128 96 64 48 32; do
    OCL_WGRAD_MULTI_TARGET=$t timeout 300 python bench.py --workload $wl --steps 100 --warmup 5 --no-roofline --no-accuracy --no-cpu-baseline --no-also > gpurun_out/${T}_${wl}_t${t}.json 2> gpurun_out/${T}_${wl}_t${t}.err
    python - $wl $t gpurun_out/${T}_${wl}_t${t}.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[3]) if l.startswith("{")][-1])
print(sys.argv[1], "OCL_WGRAD_MULTI_TARGET=%s" % sys.argv[2], "ms_per_step %.4f max %.4f" % (d["ms_per_step"], d.get("ms_per_step_max",0)), d["ms_per_step_repeats"])
PY
  done
done 2>&1 | tee gpurun_out/${T}_wgrad_multi_target.txt
