# Round 6: SupConResNet's 'mlp' head in one launch each way (OCL_HEAD_FUSED=0: the separate launches): netcheck A/B (rounding only), SCR bench A/B, the SCR tests.
T=${1:-r6ab}
mkdir -p gpurun_out
N=online-continual-learning_amd/csrc/netcheck
{
  for cfg in "220 2 32 1" "20 2 32 1" "13 1 32 1" "6 2 84 1"; do
    echo "### netcheck $cfg"
    OCL_DETERMINISTIC=1 OCL_HEAD_FUSED=0 timeout -k 5 60 $N $cfg write /tmp/ref.bin | head -1
    OCL_DETERMINISTIC=1 OCL_HEAD_FUSED=1 timeout -k 5 60 $N $cfg compare /tmp/ref.bin | grep -E "netcheck|beyond|differ|reldiff" | head -12
  done
} > gpurun_out/${T}_netcheck.txt 2>&1
cat gpurun_out/${T}_netcheck.txt | cut -c1-200
run() {  # workload, label, env...
  wl=$1; lab=$2; shift 2
  env "$@" timeout -k 10 300 python bench.py --workload $wl --steps 100 --warmup 5 --no-roofline --no-accuracy --no-cpu-baseline --no-also > gpurun_out/${T}_${wl}_${lab}.json 2> gpurun_out/${T}_${wl}_${lab}.err
  python - $wl "$lab" gpurun_out/${T}_${wl}_${lab}.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[3]) if l.startswith("{")][-1])
print(sys.argv[1], sys.argv[2], "ms_per_step %.4f max %.4f" % (d["ms_per_step"], d.get("ms_per_step_max",0)), d["ms_per_step_repeats"])
PY
}
{
run scr separate OCL_HEAD_FUSED=0
run scr fused X=1
run scr separate OCL_HEAD_FUSED=0
run scr fused X=1
} 2>&1 | tee gpurun_out/${T}_head_fused_ab.txt
timeout -k 10 1200 python -m pytest tests/test_gpu_net.py tests/test_gpu_steps.py tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | tail -4
