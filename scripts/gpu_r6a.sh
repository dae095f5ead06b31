# Round 6, call 1 (~1 min of box time, no torch): the per-phase s_memtime timeline of every convolution launch of the 220-view pass on the
# round-5 tree, the planner's per-launch times of this lease, and the whole-pass times netcheck sees.
# gpurun --timeout 420 -- 'bash scripts/gpu_r6a.sh r6a'
T=${1:-r6a}
mkdir -p gpurun_out
K=online-continual-learning_amd/csrc/kbench
N=online-continual-learning_amd/csrc/netcheck
KBENCH_TRACE=1 timeout 200 $K 220 2 32 conv 0 > gpurun_out/${T}_trace220.txt 2>&1; echo "rc=$?"
timeout 100 $K 20 1 32 conv 0 > gpurun_out/${T}_conv20.txt 2>&1; echo "rc=$?"
{
  for cfg in "220 2 32 1" "20 1 32 0" "100 1 32 0" "64 2 32 3"; do
    echo "### netcheck $cfg"
    timeout 60 $N $cfg write /tmp/ref.bin | head -2
    timeout 60 $N $cfg compare /tmp/ref.bin | grep -E "netcheck|beyond"
  done
} > gpurun_out/${T}_netcheck.txt 2>&1
grep -E "us " gpurun_out/${T}_trace220.txt | head -60
cat gpurun_out/${T}_netcheck.txt
