# CPU: the oracle's side of accuracy.aser over ten seeds (sequential, 8 threads)
for s in 0 100 200 300 400 500 600 700 800 900; do
  python bench.py --oracle-accuracy-worker $s texture_prototype 8 aser 2>/dev/null | python -c "
import json,sys,numpy as np
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); a=np.array(d['acc']); print('seed $s end_acc %.4f avg_acc %.4f' % (a[-1].mean(), np.mean([a[i,:i+1].mean() for i in range(len(a))])))"
done
