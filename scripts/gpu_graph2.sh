export PYTHONDONTWRITEBYTECODE=1
for w in scr er aser mir; do for m in 0 1 2; do
  OCL_GRAPH_MODE=$m timeout 300 python bench.py --workload $w --steps 150 --warmup 30 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$w mode $m: %.3f ms/step  %.0f img/s' % (d['ms_per_step'], d['value']))"
done; done
