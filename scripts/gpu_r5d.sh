# Round 5, call 4: full GPU suite on the tree with OCL_FWD_SAME_WEIGHTS, the unused ASER loss kernels gone and 80-channel weight-gradient
# blocks on the large layers; the NTW = 5 gate on mini-ImageNet shapes; a quiet bench line.
# gpurun --timeout 1500 -- 'bash scripts/gpu_r5d.sh r5d'
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
export TMPDIR=/tmp
T=${1:-r5d}
O=gpurun_out/${T}_out.txt
N=online-continual-learning_amd/csrc/netcheck
K=online-continual-learning_amd/csrc/kbench
{
  for cfg in "20 1 84 0" "50 1 84 0" "100 2 32 1" "64 2 32 3"; do
    echo "### netcheck $cfg: NTW = 5 gate at 3000 pixels (default) / off / everywhere"
    timeout 60 $N $cfg write /tmp/ref.bin | head -1
    for E in "OCL_WGRAD_NT5_MINK=1000000000" "OCL_WGRAD_NT5_MINK=0" "OCL_NONE=1"; do echo -n "[$E] "; env $E timeout 60 $N $cfg compare /tmp/ref.bin | grep -E "netcheck|beyond" | tr '\n' ' '; echo; done
  done
  for cfg in "20 1 84" "100 2 32"; do for E in "OCL_WGRAD_NT5_MINK=1000000000" "OCL_WGRAD_NT5_MINK=0"; do
    echo "### [$E] kbench $cfg wgrad"; env $E timeout 60 $K $cfg wgrad 2>&1 | grep -E "wgrad |MISMATCH|rror" | cut -c1-20,96-240
  done; done
} > $O 2>&1
timeout 1000 python -m pytest tests/ -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/${T}_tests.log 2>&1; echo "tests rc=$?" >> $O
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-accuracy > gpurun_out/${T}_bench.log 2>gpurun_out/${T}_bench.err; echo "bench rc=$?" >> $O
grep -E "^###|netcheck|rc=" $O | cut -c1-220; grep -E "^FAILED|passed|failed|Error" gpurun_out/${T}_tests.log | tail -5
tail -1 gpurun_out/${T}_bench.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); a=d.get('also',{})
print(len(json.dumps(d)), json.dumps(dict(scr=d['ms_per_step'], repeats=d.get('ms_per_step_repeats'), roof=d.get('roofline',{}).get('frac'), aser=a.get('aser',{}).get('ms_per_step'), er=a.get('er',{}).get('ms_per_step'), mir=a.get('mir',{}).get('ms_per_step'))))"
