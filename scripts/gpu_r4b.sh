# round 4, call 2: EPI_BNB correctness (A/B test, whole -m gpu suite) + A/B bench lines + the default bench line.   gpurun --timeout 1500 -- 'bash scripts/gpu_r4b.sh r4b'
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
T=${1:-r4b}
timeout 400 python -m pytest tests/test_gpu_ring.py -x -q -k "bn_backward" -s > gpurun_out/${T}_bnb_ab.log 2>&1; echo "bnb A/B rc=$?"; tail -5 gpurun_out/${T}_bnb_ab.log
timeout 700 python -m pytest tests -m gpu -x -q > gpurun_out/${T}_suite.log 2>&1; echo "suite rc=$?"; tail -5 gpurun_out/${T}_suite.log
Q="--no-cpu-baseline --no-accuracy --no-also"
for wl in scr er aser mir; do
  for v in 0 1 0 1; do
    OCL_BNB_EPI=$v timeout 200 python bench.py --workload $wl --steps 200 --warmup 20 --repeats 3 $Q 2>gpurun_out/${T}_err.log | python -c "
import json,sys
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    r=d.get('roofline',{})
    print('$wl bnb=$v', 'ms %.4f' % d['ms_per_step'], [round(x,4) for x in d['ms_per_step_repeats']], 'conv %.3f cal %.1f' % (r.get('frac') or 0, r.get('calibrated_peak') or 0), {k: round(v,4) for k,v in (r.get('per_step_ms') or {}).items()}, r.get('launches_per_step_all'), d.get('env',{}).get('sclk_mhz'), d.get('env',{}).get('power_w'))
"
  done
done 2>&1 | tee gpurun_out/${T}_ab.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${T}_bench_default.json 2> gpurun_out/${T}_bench_default.err; echo "default bench rc=$?"; python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/${T}_bench_default.json").read().strip().splitlines()[-1])
    print({k: d[k] for k in ("value", "ms_per_step", "ms_per_step_repeats", "preroll_ms", "env")})
    print({k: (v["ms_per_step"], v["roofline"]["frac"]) for k, v in d["also"].items()})
    for k, v in d["accuracy"].items():
        if isinstance(v, dict) and "summary" in v: print(k, v["summary"], {t: v[t]["avg_end_acc"]["mean"] for t in v if t.startswith(("hip", "paper"))})
    print(d.get("cpu_baseline"))
except Exception as e:
    print("parse failed", e)
PY
