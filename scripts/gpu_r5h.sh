# Round 5, call 9: ASER passes without readers as forward_stats_only + one kNN launch per retrieval: the schedule-only test, the ASER / MIR
# co-simulations, a quiet bench line.
# gpurun --timeout 900 -- 'bash scripts/gpu_r5h.sh r5h'
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
export TMPDIR=/tmp
T=${1:-r5h}
O=gpurun_out/${T}_out.txt; : > $O
timeout 600 python -m pytest tests/test_gpu_net.py tests/test_gpu_parity2.py tests/test_gpu_steps.py tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "same_weights or aser or mir or knn or readers" > gpurun_out/${T}_tests.log 2>&1; echo "tests rc=$?" >> $O
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-accuracy > gpurun_out/${T}_bench.log 2>gpurun_out/${T}_bench.err; echo "bench rc=$?" >> $O
cat $O; grep -E "^FAILED|passed|failed|Error" gpurun_out/${T}_tests.log | tail -5
tail -1 gpurun_out/${T}_bench.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); a=d.get('also',{})
print(len(json.dumps(d)), json.dumps(dict(scr=d['ms_per_step'], roof=d.get('roofline',{}).get('frac'), aser=a.get('aser',{}).get('ms_per_step'), aser_rep=a.get('aser',{}).get('ms_per_step_repeats'), er=a.get('er',{}).get('ms_per_step'), mir=a.get('mir',{}).get('ms_per_step'), aser_l=a.get('aser',{}).get('roofline',{}).get('launches_per_step_all'))))"
