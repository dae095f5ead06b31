# Round 6, call 4: conv_w_kernel after the prologue change, with the start of the second half of the workgroup staggered / prioritised
# (OCL_CW_SLEEP in units of 64 cycles, OCL_CW_PRIO 1 = waves 0-3, 2 = waves 4-7), on one launch of each of layers 1 - 3 of the 220-view pass.
# gpurun --timeout 300 -- 'bash scripts/gpu_r6d.sh r6d'
T=${1:-r6d}
mkdir -p gpurun_out
K=online-continual-learning_amd/csrc/kbench
{
for E in "OCL_CW_SLEEP=0" "OCL_CW_SLEEP=32" "OCL_CW_SLEEP=64" "OCL_CW_SLEEP=96" "OCL_CW_PRIO=1" "OCL_CW_PRIO=2" "OCL_CW_SLEEP=48 OCL_CW_PRIO=1" "OCL_CW_SLEEP=48 OCL_CW_PRIO=2"; do
  echo "### $E"
  for L in layer2.1.conv1 layer3.1.conv1 layer1.1.conv1; do
    env $E KBENCH_ONLY=$L timeout 100 $K 220 2 32 conv 0 | grep -E "^layer|conv_w"
  done
done
echo "### trace, OCL_CW_SLEEP=48"
for L in layer2.1.conv1 layer3.1.conv1; do
  OCL_CW_SLEEP=48 KBENCH_ONLY=$L KBENCH_TRACE=1 timeout 100 $K 220 2 32 conv 0 | grep -E "^layer|conv_w|wave"
done
} > gpurun_out/${T}_stagger.txt 2>&1
cut -c1-260 gpurun_out/${T}_stagger.txt
