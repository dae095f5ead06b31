# round 4, call 6: integer batch sums with batched replica loads vs fp64; ER data-stream overlap; replay diagnostics.   gpurun --timeout 900 -- 'bash scripts/gpu_r4f.sh r4f'
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
T=${1:-r4f}
timeout 300 python -m pytest tests/test_gpu_steps.py tests/test_gpu_parity2.py -x -q -k "data_stream or bit_reproducible or single_run_driver" > gpurun_out/${T}_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/${T}_tests.log
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
for v in 0 1 0 1; do OCL_DATA_STREAM=$v timeout 200 python bench.py --workload er --steps 200 --warmup 20 --repeats 3 $Q --no-roofline 2>gpurun_out/${T}_err.log | line "er data_stream=$v"; done 2>&1 | tee -a gpurun_out/${T}_ab.txt
for wl in aser mir; do timeout 200 python bench.py --workload $wl --steps 200 --warmup 20 --repeats 3 $Q 2>gpurun_out/${T}_err.log | line "$wl int-sums"; done 2>&1 | tee -a gpurun_out/${T}_ab.txt
OCL_GRAPH=1 OCL_GRAPH_VERBOSE=1 timeout 100 python scripts/host_cost_probe.py scr 2>&1 | grep -v "^ \|^$" | head -30 | cut -c1-200 | tee gpurun_out/${T}_graph_probe.txt
