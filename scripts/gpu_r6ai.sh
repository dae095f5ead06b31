# Round 6: the ASER combined pass without autograd between loss and backward (bench A/B through `_force_autograd` is not possible from the
# command line: compare against the previous call's numbers on the same command), the two new schedule-only tests, all ASER tests.
T=${1:-r6ai}
mkdir -p gpurun_out
run() {  # workload, label, env...
  wl=$1; lab=$2; shift 2
  env "$@" timeout -k 10 300 python bench.py --workload $wl --steps 100 --warmup 5 --no-roofline --no-accuracy --no-cpu-baseline --no-also > gpurun_out/${T}_${wl}_${lab}.json 2> gpurun_out/${T}_${wl}_${lab}.err
  python - $wl "$lab" gpurun_out/${T}_${wl}_${lab}.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[3]) if l.startswith("{")][-1])
print(sys.argv[1], sys.argv[2], "ms_per_step %.4f max %.4f" % (d["ms_per_step"], d.get("ms_per_step_max",0)), d["ms_per_step_repeats"])
PY
}
{
run aser autograd OCL_ASER_AUTOGRAD=1
run aser direct X=1
run aser autograd OCL_ASER_AUTOGRAD=1
run aser direct X=1
run aser autograd OCL_ASER_AUTOGRAD=1
run aser direct X=1
} 2>&1 | tee gpurun_out/${T}_aser_direct_ab.txt
timeout -k 10 1200 python -m pytest tests -x -q -m gpu -k "aser or ASER" 2>&1 | tail -4
