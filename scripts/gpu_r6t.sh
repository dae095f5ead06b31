# Round 6: every weight gradient of a replay-sized pass in one launch (conv_wgrad_multi_kernel, OCL_WGRAD_MULTI: 0 = per-layer launches).
# netcheck: outputs + gradients of the whole pass against the per-layer launches (bit-identical expected), whole-pass time; the bench legs.
# gpurun --timeout 1500 -- 'bash scripts/gpu_r6t.sh r6t'
T=${1:-r6t}
mkdir -p gpurun_out
N=online-continual-learning_amd/csrc/netcheck
{
  for cfg in "20 2 32 0" "20 1 32 0" "10 1 32 0" "13 1 32 0" "20 2 32 1" "47 1 32 0" "8 1 84 0"; do
    echo "### netcheck $cfg"
    OCL_WGRAD_MULTI=0 timeout 60 $N $cfg write /tmp/ref.bin | head -2
    OCL_WGRAD_MULTI=1 timeout 60 $N $cfg compare /tmp/ref.bin | grep -E "netcheck|beyond|differ"
  done
} > gpurun_out/${T}_netcheck.txt 2>&1
cat gpurun_out/${T}_netcheck.txt
for wl in er aser mir; do
  for m in 0 1 0 1; do
    OCL_WGRAD_MULTI=$m timeout 300 python bench.py --workload $wl --steps 100 --warmup 5 --no-roofline --no-accuracy --no-cpu-baseline --no-also > gpurun_out/${T}_${wl}_m${m}.json 2> gpurun_out/${T}_${wl}_m${m}.err
    python - $wl $m gpurun_out/${T}_${wl}_m${m}.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[3]) if l.startswith("{")][-1])
print(sys.argv[1], "OCL_WGRAD_MULTI=%s" % sys.argv[2], "ms_per_step %.4f max %.4f" % (d["ms_per_step"], d.get("ms_per_step_max",0)), d["ms_per_step_repeats"])
PY
  done
done 2>&1 | tee gpurun_out/${T}_wgrad_multi_bench_ab.txt
timeout 1200 python -m pytest tests/test_gpu_steps.py tests/test_gpu_net.py -x -q -m gpu 2>&1 | tail -5
timeout 600 python -m pytest tests/test_gpu_parity2.py -x -q -m gpu -k "reproducible" 2>&1 | tail -3
