# Round 5, call 17: the merged ER step hands the engine's backward one dL/dlogits buffer (no autograd slice / add launches): the
# schedule-only test, the ER co-simulations and golden free runs, ER bench line before / after is the next bundle's business.
# gpurun --timeout 900 -- 'bash scripts/gpu_r5p.sh r5q'
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
export TMPDIR=/tmp
T=${1:-r5q}
O=gpurun_out/${T}_out.txt; : > $O
timeout 700 python -m pytest tests/test_gpu_steps.py tests/test_gpu_parity2.py -m gpu -q --tb=short -p no:cacheprovider -k "er_ or _er or merged or free_running or reproducible" > gpurun_out/${T}_tests.log 2>&1; echo "tests rc=$?" >> $O
for i in 1 2; do
  timeout 300 python bench.py --workload er --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline --no-accuracy --no-also 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('er', d['ms_per_step'], d['ms_per_step_repeats'], d['roofline']['launches_per_step_all'])" >> $O
done
cat $O; grep -E "^FAILED|passed|failed|Error" gpurun_out/${T}_tests.log | tail -5
