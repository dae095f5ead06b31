# round 4: weight gradients of replay-sized passes beside the chain, at EQUAL stream priority.   gpurun --timeout 600 -- 'bash scripts/gpu_r4s.sh r4s'
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
T=${1:-r4s}
Q="--no-cpu-baseline --no-accuracy --no-also --no-roofline"
line() { python -c "
import json,sys
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('$1', 'ms %.4f' % d['ms_per_step'], [round(x,4) for x in d['ms_per_step_repeats']])
"; }
for wl in er aser; do for rep in 1 2; do
  timeout 200 python bench.py --workload $wl --steps 200 --warmup 20 --repeats 3 $Q 2>gpurun_out/${T}_err.log | line "$wl default (one stream)"
  OCL_TWO_STREAM_MIN_PIX=0 timeout 200 python bench.py --workload $wl --steps 200 --warmup 20 --repeats 3 $Q 2>gpurun_out/${T}_err.log | line "$wl two streams, side stream lowest priority"
  OCL_TWO_STREAM_MIN_PIX=0 OCL_SIDE_PRIO=0 timeout 200 python bench.py --workload $wl --steps 200 --warmup 20 --repeats 3 $Q 2>gpurun_out/${T}_err.log | line "$wl two streams, equal priority"
done; done 2>&1 | tee gpurun_out/${T}_ab.txt
