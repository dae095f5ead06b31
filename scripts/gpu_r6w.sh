# Round 6: (1) weight-gradient time the second stream does not hide: SCR step with and without the weight gradients (OCL_DEBUG_SKIP_WGRAD=1,
# timing only), two streams and one; (2) the ER / ASER / MIR legs with the merged launch at its default split; (3) GPU tests of the step + net.
# gpurun --timeout 1500 -- 'bash scripts/gpu_r6w.sh r6w'
T=${1:-r6w}
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
run scr two_streams X=1
run scr two_streams_no_wgrad OCL_DEBUG_SKIP_WGRAD=1
run scr one_stream OCL_SINGLE_STREAM=1
run scr one_stream_no_wgrad OCL_SINGLE_STREAM=1 OCL_DEBUG_SKIP_WGRAD=1
run er default X=1
run er no_wgrad OCL_DEBUG_SKIP_WGRAD=1
run aser default X=1
run aser per_layer OCL_WGRAD_MULTI=0
run aser default X=1
run mir default X=1
} 2>&1 | tee gpurun_out/${T}_exposed_wgrad.txt
timeout -k 10 1200 python -m pytest tests/test_gpu_steps.py tests/test_gpu_net.py tests/test_gpu_netcheck.py -x -q -m gpu 2>&1 | tail -4
