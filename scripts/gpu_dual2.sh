export PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_net.py -m gpu -q --tb=short -p no:cacheprovider -x -k "dual_chain or SCR" > gpurun_out/d2_tests.log 2>&1; echo "tests rc=$?"; tail -12 gpurun_out/d2_tests.log
for m in 0 1 2; do
  OCL_DUAL_CHAIN=$m timeout 300 python bench.py --workload scr --steps 200 --warmup 30 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('scr dual mode $m: %.3f ms/step  %.0f img/s' % (d['ms_per_step'], d['value']))"
done
