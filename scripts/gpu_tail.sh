export PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gpu_net.py tests/test_gpu_steps.py -m gpu -q --tb=short -p no:cacheprovider -x -k "SCR or scr" > gpurun_out/tail_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/tail_tests.log
for rep in 1 2; do for f in 1 0; do
  OCL_TAIL_MAIN=$f timeout 300 python bench.py --workload scr --steps 300 --warmup 30 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('scr OCL_TAIL_MAIN=$f: %.3f ms/step  %.0f img/s' % (d['ms_per_step'], d['value']))"
done; done
