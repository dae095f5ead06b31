# Round 6, call 6: finer stamps inside conv_w_kernel's prologue.
T=${1:-r6f}
mkdir -p gpurun_out
K=online-continual-learning_amd/csrc/kbench
for L in layer2.1.conv1 layer3.1.conv1; do
  KBENCH_ONLY=$L KBENCH_TRACE=1 timeout 100 $K 220 2 32 conv 0 | grep -E "^layer|conv_w|wave"
done > gpurun_out/${T}_trace.txt 2>&1
cut -c1-330 gpurun_out/${T}_trace.txt
