# Round 6: the tree as committed -- full GPU suite, smoke, the driver's bench command with its wall time.
T=${1:-r6v2}
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
if [ -z "$SKIP_TESTS" ]; then timeout -k 10 1500 python -m pytest tests/ -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/${T}_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/${T}_tests.log; fi
timeout -k 10 300 python __graft_entry__.py smoke > gpurun_out/${T}_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/${T}_smoke.log
t0=$(date +%s); timeout -k 10 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${T}_bench_scr.log 2> gpurun_out/${T}_bench_scr.err; echo "bench rc=$? wall $(( $(date +%s) - t0 )) s"
python - $T <<'PY'
import json,sys
d=json.loads([l for l in open("gpurun_out/%s_bench_scr.log" % sys.argv[1]) if l.startswith("{")][-1])
print("scr", d["ms_per_step"], d["value"], "frac", d["roofline"]["frac"], "exposed", d["roofline"]["wgrad"].get("exposed"))
for k,a in d["also"].items(): print(k, a["ms_per_step"], a.get("ms_per_step_max"))
print("line bytes", len(json.dumps(d)))
PY
