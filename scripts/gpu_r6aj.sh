# Round 6, last call (an experiment): the training loop on a HIGH-priority stream of its own (OCL_LOOP_PRIO=-1) against the default stream --
# does the dependent chain get the CUs sooner against the lowest-priority weight-gradient stream?  SCR (two streams), ER and ASER for control.
T=${1:-r6aj}
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
run scr default_stream X=1
run scr high_priority_stream OCL_LOOP_PRIO=-1
run scr normal_priority_own_stream OCL_LOOP_PRIO=0
run scr default_stream X=1
run scr high_priority_stream OCL_LOOP_PRIO=-1
run er default_stream X=1
run er high_priority_stream OCL_LOOP_PRIO=-1
run aser default_stream X=1
run aser high_priority_stream OCL_LOOP_PRIO=-1
} 2>&1 | tee gpurun_out/${T}_loop_prio_ab.txt
