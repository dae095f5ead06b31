# Round 6: the class-balanced draw with exclusions from a simulation of CPython's set table (OCL_CBRS_EMULATE=0: builds one set per class): ASER A/B,
# the drift probe, the ASER tests (index sequences against the oracle: the draws must be the reference's).
T=${1:-r6aq}
mkdir -p gpurun_out
run() {  # workload, label, repeats, env...
  wl=$1; lab=$2; rep=$3; shift 3
  env "$@" timeout -k 10 600 python bench.py --workload $wl --steps 100 --warmup 5 --repeats $rep --no-roofline --no-accuracy --no-cpu-baseline --no-also > gpurun_out/${T}_${wl}_${lab}.json 2> gpurun_out/${T}_${wl}_${lab}.err
  python - $wl "$lab" gpurun_out/${T}_${wl}_${lab}.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[3]) if l.startswith("{")][-1])
print(sys.argv[1], sys.argv[2], "ms_per_step %.4f max %.4f" % (d["ms_per_step"], d.get("ms_per_step_max",0)), d["ms_per_step_repeats"])
PY
}
{
run aser builds_sets 5 OCL_CBRS_EMULATE=0
run aser simulated 5 X=1
run aser builds_sets 5 OCL_CBRS_EMULATE=0
run aser simulated 5 X=1
run aser simulated_15_repeats 15 X=1
} 2>&1 | tee gpurun_out/${T}_cbrs_emulate_ab.txt
timeout -k 10 600 python scripts/aser_drift_probe.py --repeats 12 2>/dev/null | cut -c1-150 | tee gpurun_out/${T}_aser_drift.txt
timeout -k 10 1200 python -m pytest tests -x -q -m gpu -k "aser or ASER" 2>&1 | tail -3
