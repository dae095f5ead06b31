# round 4, call 8: launch replay (agents on their own stream) on the four workloads; default tree check.   gpurun --timeout 900 -- 'bash scripts/gpu_r4h.sh r4h'
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
T=${1:-r4h}
timeout 400 python -m pytest tests/test_gpu_ring.py tests/test_gpu_steps.py tests/test_gpu_parity2.py -x -q -s -k "replay or data_stream or bit_reproducible" > gpurun_out/${T}_tests.log 2>&1; echo "tests rc=$?"; grep -E "captured|passed|failed" gpurun_out/${T}_tests.log | head -30 | cut -c1-200
Q="--no-cpu-baseline --no-accuracy --no-also --no-roofline"
line() { python -c "
import json,sys
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('$1', 'ms %.4f' % d['ms_per_step'], [round(x,4) for x in d['ms_per_step_repeats']], d.get('env',{}).get('sclk_mhz'))
"; }
for wl in aser er mir scr; do
  for g in 0 1 0 1; do
    OCL_GRAPH=$g timeout 200 python bench.py --workload $wl --steps 200 --warmup 20 --repeats 3 $Q 2>gpurun_out/${T}_err.log | line "$wl graph=$g"
  done
done 2>&1 | tee gpurun_out/${T}_ab.txt
