# Round 5, call 2: the winners of call 1 as defaults (OCL_WGRAD_Q, OCL_WGRAD_XCD, OCL_REDUCE_GROUP=4, stage-2 BatchNorm epilogue by pass size)
# through the full GPU suite, smoke, and a quiet bench line (no accuracy / cpu baseline legs).
# gpurun --timeout 1500 -- 'bash scripts/gpu_r5b.sh r5b'
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
export TMPDIR=/tmp
T=${1:-r5b}
L=gpurun_out/${T}_info.log; : > $L
timeout 1000 python -m pytest tests/ -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/${T}_tests.log 2>&1; echo "tests rc=$?" >> $L
timeout 200 python __graft_entry__.py smoke > gpurun_out/${T}_smoke.log 2>&1; echo "smoke rc=$?" >> $L
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-accuracy > gpurun_out/${T}_bench.log 2>gpurun_out/${T}_bench.err; echo "bench rc=$?" >> $L
cat $L; grep -E "^FAILED|passed|failed|Error" gpurun_out/${T}_tests.log | tail -5; tail -1 gpurun_out/${T}_smoke.log
tail -1 gpurun_out/${T}_bench.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); a=d.get('also',{})
print(json.dumps(dict(scr=d['ms_per_step'], repeats=d.get('ms_per_step_repeats'), roof=d.get('roofline',{}).get('frac'), aser=a.get('aser',{}).get('ms_per_step'), er=a.get('er',{}).get('ms_per_step'), mir=a.get('mir',{}).get('ms_per_step'))))"
