# round 4: ocl_gather_rows_pair on the retrieval paths: step parity (ER / SCR / ASER / MIR co-simulations), ASER bench A/B against the previous commit is not possible in one tree -> before/after by lease.   gpurun --timeout 900 -- 'bash scripts/gpu_r4u.sh r4u'
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
T=${1:-r4u}
timeout 600 python -m pytest tests/test_gpu_steps.py tests/test_gpu_f4.py tests/test_gpu_kernels.py -x -q > gpurun_out/${T}_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/${T}_tests.log
timeout 300 python -m pytest tests/test_gpu_parity2.py -x -q -k "aser or mir or evaluate or review" > gpurun_out/${T}_tests2.log 2>&1; echo "tests2 rc=$?"; tail -3 gpurun_out/${T}_tests2.log
Q="--no-cpu-baseline --no-accuracy --no-also --no-roofline"
line() { python -c "
import json,sys
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('$1', 'ms %.4f' % d['ms_per_step'], [round(x,4) for x in d['ms_per_step_repeats']])
"; }
for wl in aser aser aser er scr mir; do timeout 200 python bench.py --workload $wl --steps 200 --warmup 20 --repeats 3 $Q 2>gpurun_out/${T}_err.log | line "$wl"; done 2>&1 | tee gpurun_out/${T}_bench.txt
timeout 100 python scripts/host_cost_probe.py aser 2>&1 | grep -E "pure host|C entry" | tee -a gpurun_out/${T}_bench.txt
