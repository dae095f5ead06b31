# round 4: weight-gradient reduction with four output groups per workgroup: parity (network + steps), bench.   gpurun --timeout 900 -- 'bash scripts/gpu_r4t.sh r4t'
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
T=${1:-r4t}
timeout 500 python -m pytest tests/test_gpu_net.py tests/test_gpu_steps.py -x -q > gpurun_out/${T}_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/${T}_tests.log
Q="--no-cpu-baseline --no-accuracy --no-also"
line() { python -c "
import json,sys
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    r=d.get('roofline',{})
    print('$1', 'ms %.4f' % d['ms_per_step'], [round(x,4) for x in d['ms_per_step_repeats']], {k: round(v,4) for k,v in (r.get('per_step_ms') or {}).items()})
"; }
for wl in er scr aser mir; do for rep in 1 2; do
  timeout 200 python bench.py --workload $wl --steps 200 --warmup 20 --repeats 3 $Q 2>gpurun_out/${T}_err.log | line "$wl"
done; done 2>&1 | tee gpurun_out/${T}_bench.txt
