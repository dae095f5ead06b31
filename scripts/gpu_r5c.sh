# Round 5, call 3: the cleaned-up tree (wgrad TAB=0 / PD=2 forms and ~15 switches deleted) through the tests that name the remaining
# switches, whole-pass times (netcheck), and a planner experiment: 80-channel output blocks (NTW = 5) for layers 3 - 4.
# gpurun --timeout 900 -- 'bash scripts/gpu_r5c.sh r5c'
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
T=${1:-r5c}
O=gpurun_out/${T}_out.txt
N=online-continual-learning_amd/csrc/netcheck
K=online-continual-learning_amd/csrc/kbench
{
  for cfg in "220 2 32 1" "20 1 32 0" "20 1 84 0"; do
    echo "### netcheck $cfg"
    timeout 60 $N $cfg write /tmp/ref.bin | head -1
    for E in "OCL_WGRAD_NTMAX=5" "OCL_WGRAD_NTMAX=5 OCL_WGRAD_LDS_KB=100"; do echo -n "[$E] "; env $E timeout 60 $N $cfg compare /tmp/ref.bin | grep -E "netcheck|beyond" | tr '\n' ' '; echo; done
    timeout 60 $N $cfg write /tmp/ref.bin | head -1
  done
  for E in "OCL_NONE=1" "OCL_WGRAD_NTMAX=5" "OCL_WGRAD_NTMAX=5 OCL_WGRAD_LDS_KB=100" "OCL_WGRAD_NTMAX=4"; do
    echo "### [$E] kbench 220 2 32 wgrad"; env $E timeout 60 $K 220 2 32 wgrad 2>&1 | grep -E "wgrad |MISMATCH|rror" | cut -c1-20,96-240
  done
  for E in "OCL_NONE=1" "OCL_WGRAD_NTMAX=5"; do
    echo "### [$E] kbench 20 1 32 wgrad"; env $E timeout 60 $K 20 1 32 wgrad 2>&1 | grep -E "wgrad |MISMATCH|rror" | cut -c1-20,96-240
  done
} > $O 2>&1
timeout 700 python -m pytest tests/test_gpu_netcheck.py tests/test_gpu_ring.py tests/test_gpu_steps.py -m gpu -q --tb=short -p no:cacheprovider -k "netcheck or ring or bn_backward or conv_s or replay or data_stream or 4x4x1" > gpurun_out/${T}_tests.log 2>&1; echo "tests rc=$?" >> $O
grep -E "^###|netcheck|rc=" $O | cut -c1-220; tail -3 gpurun_out/${T}_tests.log
