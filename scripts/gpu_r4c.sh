# round 4, call 3: weight gradients of replay-sized passes beside the dependent chain (OCL_TWO_STREAM_MIN_PIX) + the segment forward.   gpurun --timeout 1200 -- 'bash scripts/gpu_r4c.sh r4c'
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
T=${1:-r4c}
OCL_TWO_STREAM_MIN_PIX=0 timeout 500 python -m pytest tests/test_gpu_steps.py tests/test_gpu_net.py tests/test_gpu_f4.py -x -q > gpurun_out/${T}_tests_two_stream.log 2>&1; echo "tests (two-stream small passes) rc=$?"; tail -3 gpurun_out/${T}_tests_two_stream.log
timeout 400 python -m pytest tests/test_gpu_parity2.py tests/test_gpu_kernels.py -x -q > gpurun_out/${T}_tests_rest.log 2>&1; echo "tests (rest, default) rc=$?"; tail -3 gpurun_out/${T}_tests_rest.log
Q="--no-cpu-baseline --no-accuracy --no-also --no-roofline"
for wl in er aser mir scr; do
  for v in 1000000000 0 49152 0 1000000000; do
    OCL_TWO_STREAM_MIN_PIX=$v timeout 200 python bench.py --workload $wl --steps 200 --warmup 20 --repeats 3 $Q 2>gpurun_out/${T}_err.log | python -c "
import json,sys
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('$wl min_pix=$v', 'ms %.4f' % d['ms_per_step'], [round(x,4) for x in d['ms_per_step_repeats']], d.get('env',{}).get('sclk_mhz'), d.get('env',{}).get('power_w'))
"
  done
done 2>&1 | tee gpurun_out/${T}_ab.txt
