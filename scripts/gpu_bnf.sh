export PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out
(cd online-continual-learning_amd/csrc && timeout 60 ./kbench 220 2 32 bn 0 2>&1 | grep "^bn" )
timeout 400 python -m pytest tests/test_gpu_net.py -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/bnf_tests.log 2>&1; echo "net tests rc=$?"; tail -3 gpurun_out/bnf_tests.log
for f in 1 0; do
  OCL_BN_FUSED=$f timeout 300 python bench.py --workload scr --steps 200 --warmup 30 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('scr OCL_BN_FUSED=$f: %.3f ms/step  %.0f img/s' % (d['ms_per_step'], d['value']))"
done
OCL_BN_FUSED=1 OCL_SINGLE_STREAM=1 timeout 300 python bench.py --workload scr --steps 200 --warmup 30 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('scr fused single-stream: %.3f ms/step' % (d['ms_per_step']))"
OCL_BN_FUSED=0 OCL_SINGLE_STREAM=1 timeout 300 python bench.py --workload scr --steps 200 --warmup 30 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('scr unfused single-stream: %.3f ms/step' % (d['ms_per_step']))"
for w in er; do for f in 1 0; do
  OCL_BN_FUSED=$f timeout 300 python bench.py --workload $w --steps 100 --warmup 20 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$w OCL_BN_FUSED=$f: %.3f ms/step' % (d['ms_per_step']))"
done; done
