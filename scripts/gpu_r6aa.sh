# Round 6: what the forward's side-stream extras (projection shortcuts, head weight gradients beside the chain, from 96 images on) cost / give at 220 views
# (OCL_SIDE_EXTRA_MIN=100000: off), with and without the weight gradients; + the bench's exposed-weight-gradient leg after the import fix.
T=${1:-r6aa}
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
run scr default X=1
run scr no_side_extras OCL_SIDE_EXTRA_MIN=100000
run scr default X=1
run scr no_side_extras OCL_SIDE_EXTRA_MIN=100000
run scr default_no_wgrad OCL_DEBUG_SKIP_WGRAD=1
run scr no_side_extras_no_wgrad OCL_SIDE_EXTRA_MIN=100000 OCL_DEBUG_SKIP_WGRAD=1
run mir default X=1
run mir no_side_extras OCL_SIDE_EXTRA_MIN=100000
} 2>&1 | tee gpurun_out/${T}_side_extra_ab.txt
timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-accuracy --no-cpu-baseline --no-also 2> gpurun_out/${T}_bench.err | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('exposed leg:', d['roofline']['wgrad'].get('exposed'))"
