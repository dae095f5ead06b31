# Round 5, call 1 (~2 min of box time, no torch): every switch round 4 wrote and never ran, through kbench + netcheck only.
#   gpu_r5_wgrad.sh: wgradtrace, OCL_WGRAD_XCD / OCL_WGRAD_PD / OCL_REDUCE_GROUP per layer and through the whole pass;
#   step 1 of gpu_r5_first.sh: OCL_BNB_EPI2 / OCL_WGRAD_Q through the whole pass on five shapes.
# gpurun --timeout 420 -- 'bash scripts/gpu_r5a.sh r5a'
T=${1:-r5a}
bash scripts/gpu_r5_wgrad.sh ${T} > /dev/null 2>&1
mkdir -p gpurun_out
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
grep -E "^###|rc=|MISMATCH|NaN|netcheck" $O | cut -c1-200
