# Round 6: ASER's split feature pass again, now that the host draws are 3 x cheaper (OCL_ASER_SPLIT=0: one pass after both draws).
T=${1:-r6ar}
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
run aser one_pass 5 OCL_ASER_SPLIT=0
run aser split 5 X=1
run aser one_pass 5 OCL_ASER_SPLIT=0
run aser split 5 X=1
run aser one_pass 5 OCL_ASER_SPLIT=0
run aser split 5 X=1
} 2>&1 | tee gpurun_out/${T}_aser_split_ab2.txt
