# Round 6, call 8: conv_w_kernel, start of waves 4-7 delayed by OCL_CW_SLEEP/8 x s_sleep(8) (one s_sleep(8) measured ~2000 cycles, call 6).
T=${1:-r6h}
mkdir -p gpurun_out
K=online-continual-learning_amd/csrc/kbench
{
for E in "OCL_CW_SLEEP=0" "OCL_CW_SLEEP=8" "OCL_CW_SLEEP=16" "OCL_CW_SLEEP=24" "OCL_CW_SLEEP=16 OCL_CW_PRIO=3"; do
  echo "### $E"
  for L in layer2.1.conv1 layer3.1.conv1 layer1.1.conv1; do
    env $E KBENCH_ONLY=$L timeout 100 $K 220 2 32 conv 0 | grep -E "conv_w"
  done
done
echo "### trace, OCL_CW_SLEEP=16"
for L in layer2.1.conv1 layer3.1.conv1; do
  OCL_CW_SLEEP=16 KBENCH_ONLY=$L KBENCH_TRACE=1 timeout 100 $K 220 2 32 conv 0 | grep -E "^layer|conv_w|wave"
done
} > gpurun_out/${T}_stagger.txt 2>&1
cut -c1-300 gpurun_out/${T}_stagger.txt
