# round 4, call 4: whole suite on the deterministic batch sums + segment forward; host cost probe.   gpurun --timeout 1200 -- 'bash scripts/gpu_r4d.sh r4d'
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
T=${1:-r4d}
for wl in scr er aser; do timeout 120 python scripts/host_cost_probe.py $wl; done > gpurun_out/${T}_host_cost.txt 2>&1; head -5 gpurun_out/${T}_host_cost.txt
timeout 200 python -m pytest tests/test_gpu_parity2.py -x -q -k "bit_reproducible or single_run_driver" > gpurun_out/${T}_repro.log 2>&1; echo "reproducibility rc=$?"; tail -4 gpurun_out/${T}_repro.log
timeout 800 python -m pytest tests -m gpu -x -q > gpurun_out/${T}_suite.log 2>&1; echo "suite rc=$?"; tail -4 gpurun_out/${T}_suite.log
Q="--no-cpu-baseline --no-accuracy --no-also"
for wl in scr er; do
  timeout 200 python bench.py --workload $wl --steps 200 --warmup 20 --repeats 3 $Q 2>gpurun_out/${T}_err.log | python -c "
import json,sys
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    r=d.get('roofline',{})
    print('$wl', 'ms %.4f' % d['ms_per_step'], [round(x,4) for x in d['ms_per_step_repeats']], 'conv %.3f cal %.1f' % (r.get('frac') or 0, r.get('calibrated_peak') or 0), {k: round(v,4) for k,v in (r.get('per_step_ms') or {}).items()}, d.get('env',{}).get('sclk_mhz'))
"
done 2>&1 | tee gpurun_out/${T}_bench.txt
