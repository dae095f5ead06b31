# First GPU call of round 5: the two switches round 4 left OFF for lack of GPU time.
#   step 1 (~30 s, no torch): csrc/netcheck, whole training pass, each switch against the default -- OCL_BNB_EPI2=1 has NEVER run;
#          it must agree with the default within rounding on every tensor (exit code of `compare`), on SCR / ER / odd / 84 x 84 shapes;
#   step 2: the full -m gpu suite under both switches;
#   step 3: bench lines (quiet: no accuracy / cpu baseline) default / Q / Q + EPI2, alternating, SCR + also legs.
# gpurun --timeout 2400 -- 'bash scripts/gpu_r5_first.sh r5a'
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
T=${1:-r5a}
O=gpurun_out/${T}_switches.txt
N=online-continual-learning_amd/csrc/netcheck
{
  for cfg in "220 2 32 1" "20 1 32 0" "13 1 32 0" "20 1 84 0" "64 2 32 3"; do
    echo "### netcheck $cfg   (n groups hw head)"
    OCL_DETERMINISTIC=1 timeout 60 $N $cfg write /tmp/ref.bin | head -1
    for E in "OCL_BNB_EPI2=1" "OCL_WGRAD_Q=1" "OCL_BNB_EPI2=1 OCL_WGRAD_Q=1"; do
      echo "# $E (order-independent sums)"; env OCL_DETERMINISTIC=1 $E timeout 60 $N $cfg compare /tmp/ref.bin; echo "rc=$?"
    done
    echo "# pass time, default sums: default / EPI2 / Q / both"
    timeout 60 $N $cfg write /tmp/ref2.bin | head -1
    for E in "OCL_BNB_EPI2=1" "OCL_WGRAD_Q=1" "OCL_BNB_EPI2=1 OCL_WGRAD_Q=1"; do env $E timeout 60 $N $cfg compare /tmp/ref2.bin | grep -E "netcheck|beyond"; done
  done
} > $O 2>&1
grep -E "^###|rc=|MISMATCH|NaN|netcheck" $O | cut -c1-160
for E in "OCL_WGRAD_Q=1" "OCL_WGRAD_Q=1 OCL_BNB_EPI2=1"; do
  tag=$(echo $E | tr -c 'A-Za-z0-9' '_')
  env $E timeout 1500 python -m pytest tests/ -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/${T}_tests_${tag}.log 2>&1
  echo "pytest [$E] rc=$?  $(tail -1 gpurun_out/${T}_tests_${tag}.log)" | tee -a $O
done
Q="--no-cpu-baseline --no-accuracy"
for rep in 1 2; do for E in "OCL_NONE=1" "OCL_WGRAD_Q=1" "OCL_WGRAD_Q=1 OCL_BNB_EPI2=1"; do
  echo "### bench [$E] rep $rep" >> $O
  env $E timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 $Q 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    a=d.get('also',{})
    print(json.dumps(dict(scr=d['ms_per_step'], repeats=d.get('ms_per_step_repeats'), aser=a.get('aser',{}).get('ms_per_step'), er=a.get('er',{}).get('ms_per_step'), mir=a.get('mir',{}).get('ms_per_step'), conv_frac=d.get('roofline',{}).get('frac'))))
" >> $O
done; done
tail -8 $O
