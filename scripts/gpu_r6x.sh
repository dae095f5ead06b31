# Round 6: the two-stream backward without ring hand-backs (one dL/dy buffer per layer, OCL_DY_KEEP) and with fewer hand-overs to the
# weight-gradient stream (OCL_WGRAD_FLUSH layers per event pair).  SCR step, with and without the weight gradients.
# gpurun --timeout 1500 -- 'bash scripts/gpu_r6x.sh r6x'
T=${1:-r6x}
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
run scr ring OCL_DY_KEEP=0
run scr keep OCL_DY_KEEP=1
run scr keep_flush2 OCL_WGRAD_FLUSH=2
run scr keep_flush3 OCL_WGRAD_FLUSH=3
run scr keep_flush5 OCL_WGRAD_FLUSH=5
run scr keep_flush10 OCL_WGRAD_FLUSH=10
run scr keep_flush21 OCL_WGRAD_FLUSH=21
run scr ring_no_wgrad OCL_DY_KEEP=0 OCL_DEBUG_SKIP_WGRAD=1
run scr keep_no_wgrad OCL_DEBUG_SKIP_WGRAD=1
run scr keep_flush5_no_wgrad OCL_WGRAD_FLUSH=5 OCL_DEBUG_SKIP_WGRAD=1
run mir ring OCL_DY_KEEP=0
run mir keep OCL_DY_KEEP=1
run mir keep_flush3 OCL_WGRAD_FLUSH=3
} 2>&1 | tee gpurun_out/${T}_dy_keep.txt
