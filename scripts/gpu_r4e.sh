# round 4, call 5: launch-sequence replay (OCL_GRAPH=1) and integer vs fp64 batch sums, A/B on one box.   gpurun --timeout 1200 -- 'bash scripts/gpu_r4e.sh r4e'
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
T=${1:-r4e}
OCL_GRAPH=1 timeout 500 python -m pytest tests/test_gpu_steps.py tests/test_gpu_net.py tests/test_gpu_parity2.py -x -q -k "not sharded and not bench_gpus and not free_running" > gpurun_out/${T}_tests_graph.log 2>&1; echo "tests (OCL_GRAPH=1) rc=$?"; tail -3 gpurun_out/${T}_tests_graph.log
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
    timeout 200 python bench.py --workload $wl --steps 200 --warmup 20 --repeats 3 $Q 2>gpurun_out/${T}_err.log | line "$wl int-sums graph=0"
    OCL_LIB=$FP timeout 200 python bench.py --workload $wl --steps 200 --warmup 20 --repeats 3 $Q 2>gpurun_out/${T}_err.log | line "$wl fp64-sums graph=0"
    OCL_GRAPH=1 timeout 200 python bench.py --workload $wl --steps 200 --warmup 20 --repeats 3 $Q 2>gpurun_out/${T}_err.log | line "$wl int-sums graph=1"
  done
done 2>&1 | tee gpurun_out/${T}_ab.txt
for wl in aser mir; do
  for g in 0 1 0 1; do
    OCL_GRAPH=$g timeout 200 python bench.py --workload $wl --steps 200 --warmup 20 --repeats 3 $Q --no-roofline 2>gpurun_out/${T}_err.log | line "$wl int-sums graph=$g"
  done
done 2>&1 | tee -a gpurun_out/${T}_ab.txt
for wl in scr er; do OCL_GRAPH=1 timeout 100 python scripts/host_cost_probe.py $wl 2>&1 | head -4; done | tee gpurun_out/${T}_host_cost_graph.txt
