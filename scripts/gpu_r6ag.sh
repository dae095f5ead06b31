# Round 6: ASER retrieval with the batch + candidate feature pass issued BEFORE the host's cooperative draw (OCL_ASER_SPLIT=0: one pass after both draws).
T=${1:-r6ag}
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
run aser one_pass OCL_ASER_SPLIT=0
run aser split X=1
run aser one_pass OCL_ASER_SPLIT=0
run aser split X=1
run aser one_pass OCL_ASER_SPLIT=0
run aser split X=1
} 2>&1 | tee gpurun_out/${T}_aser_split_ab.txt
timeout -k 10 1200 python -m pytest tests -x -q -m gpu -k "aser or ASER" 2>&1 | tail -4
