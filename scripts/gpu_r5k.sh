# Round 5, call 12: the one-pass BatchNorm backward (grid-wide arrival: 12 - 13 us per launch on a 20-image pass) against the reduce + apply
# pair (OCL_BN_FUSED=0) through the whole pass, small passes only.
# gpurun --timeout 300 -- 'bash scripts/gpu_r5k.sh r5k'
mkdir -p gpurun_out
T=${1:-r5k}
O=gpurun_out/${T}_out.txt
N=online-continual-learning_amd/csrc/netcheck
{
  for cfg in "20 1 32 0" "13 1 32 0" "20 1 84 0" "10 1 32 0" "64 2 32 3"; do
    echo "### netcheck $cfg: default / OCL_BN_FUSED=0 / default / OCL_BN_FUSED=0"
    for i in 1 2; do
      timeout 60 $N $cfg write /tmp/ref.bin | head -1
      OCL_BN_FUSED=0 timeout 60 $N $cfg compare /tmp/ref.bin | grep -E "netcheck|beyond" | tr '\n' ' '; echo
    done
  done
} > $O 2>&1
cat $O | cut -c1-220
