export PYTHONDONTWRITEBYTECODE=1
B="python bench.py --workload scr --steps 300 --warmup 30 --no-cpu-baseline --no-roofline"
echo "single:"; timeout 300 $B 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms/step %.0f img/s' % (d['ms_per_step'], d['value']))"
echo "two processes on one GPU:"
(timeout 300 $B 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('A %.3f ms/step %.0f img/s' % (d['ms_per_step'], d['value']))") &
(timeout 300 $B 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('B %.3f ms/step %.0f img/s' % (d['ms_per_step'], d['value']))") &
wait
echo "single, OCL_SINGLE_STREAM=1:"; OCL_SINGLE_STREAM=1 timeout 300 $B 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms/step %.0f img/s' % (d['ms_per_step'], d['value']))"
echo "two processes, OCL_SINGLE_STREAM=1:"
(OCL_SINGLE_STREAM=1 timeout 300 $B 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('A %.3f ms/step %.0f img/s' % (d['ms_per_step'], d['value']))") &
(OCL_SINGLE_STREAM=1 timeout 300 $B 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('B %.3f ms/step %.0f img/s' % (d['ms_per_step'], d['value']))") &
wait
