# Round 6 (experiment): does the young-generation collector show in the ASER / ER step?  OCL_GC_THRESHOLD: generation-0 threshold (default 700), 0 = off.
T=${1:-r6am}
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
run aser default X=1
run aser gc_off OCL_GC_THRESHOLD=0
run aser gc_100000 OCL_GC_THRESHOLD=100000
run aser default X=1
run aser gc_off OCL_GC_THRESHOLD=0
run aser gc_100000 OCL_GC_THRESHOLD=100000
run er default X=1
run er gc_off OCL_GC_THRESHOLD=0
} 2>&1 | tee gpurun_out/${T}_gc_threshold_ab.txt
