# Round 6, last call (an experiment): is the ASER loop's pipelining (next batch's statistics pass issued before the update's synchronisation,
# OCL_ASER_PIPELINE, round 4) still worth anything now that the host half behind the synchronisation is 3 x cheaper?
T=${1:-r6as}
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
run aser pipelined 5 X=1
run aser not_pipelined 5 OCL_ASER_PIPELINE=0
run aser pipelined 5 X=1
run aser not_pipelined 5 OCL_ASER_PIPELINE=0
} 2>&1 | tee gpurun_out/${T}_aser_pipeline_ab.txt
