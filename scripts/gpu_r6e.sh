# Round 6, call 5: conv_w_kernel with the priority raised outside the K loop (OCL_CW_PRIO=3), with and without the staggered start.
T=${1:-r6e}
mkdir -p gpurun_out
K=online-continual-learning_amd/csrc/kbench
{
for E in "OCL_CW_SLEEP=0" "OCL_CW_PRIO=3" "OCL_CW_SLEEP=48 OCL_CW_PRIO=3"  "OCL_CW_SLEEP=96 OCL_CW_PRIO=3"; do
  echo "### $E"
  for L in layer2.1.conv1 layer3.1.conv1 layer1.1.conv1; do
    env $E KBENCH_ONLY=$L timeout 100 $K 220 2 32 conv 0 | grep -E "^layer|conv_w"
  done
done
echo "### trace, OCL_CW_SLEEP=48 OCL_CW_PRIO=3"
for L in layer2.1.conv1 layer3.1.conv1; do
  OCL_CW_SLEEP=48 OCL_CW_PRIO=3 KBENCH_ONLY=$L KBENCH_TRACE=1 timeout 100 $K 220 2 32 conv 0 | grep -E "^layer|conv_w|wave"
done
} > gpurun_out/${T}_prio.txt 2>&1
cut -c1-260 gpurun_out/${T}_prio.txt
