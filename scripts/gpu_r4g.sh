# round 4, call 7: integer sums (plain batched replica loads) vs fp64; launch replay under a side stream.   gpurun --timeout 900 -- 'bash scripts/gpu_r4g.sh r4g'
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
T=${1:-r4g}
Q="--no-cpu-baseline --no-accuracy --no-also"
FP=$PWD/online-continual-learning_amd/libocl_hip_fp64.so
line() { python -c "
import json,sys
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    r=d.get('roofline',{})
    print('$1', 'ms %.4f' % d['ms_per_step'], [round(x,4) for x in d['ms_per_step_repeats']], {k: round(v,4) for k,v in (r.get('per_step_ms') or {}).items()}, d.get('env',{}).get('sclk_mhz'))
"; }
for wl in scr er; do
  for rep in 1 2; do
    timeout 200 python bench.py --workload $wl --steps 200 --warmup 20 --repeats 3 $Q 2>gpurun_out/${T}_err.log | line "$wl int-sums"
    OCL_LIB=$FP timeout 200 python bench.py --workload $wl --steps 200 --warmup 20 --repeats 3 $Q 2>gpurun_out/${T}_err.log | line "$wl fp64-sums"
  done
done 2>&1 | tee gpurun_out/${T}_ab.txt
timeout 400 python -m pytest tests/test_gpu_ring.py -x -q -s -k "replay" > gpurun_out/${T}_replay_test.log 2>&1; echo "replay test rc=$?"; tail -15 gpurun_out/${T}_replay_test.log | cut -c1-250
for g in 0 1; do OCL_PROBE_STREAM=1 OCL_GRAPH=$g timeout 100 python scripts/host_cost_probe.py scr 2>&1 | grep -E "pure host|C entry"; OCL_PROBE_STREAM=1 OCL_GRAPH=$g timeout 100 python scripts/host_cost_probe.py er 2>&1 | grep -E "pure host|C entry"; done | tee gpurun_out/${T}_graph_probe.txt
