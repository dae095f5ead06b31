# Round 6, call 14: conv_wx_kernel: thirds of the K loop (trace build)
T=${1:-r6l}
mkdir -p gpurun_out
K=online-continual-learning_amd/csrc/kbench
for L in layer2.1.conv1 layer3.1.conv1; do
  KBENCH_ONLY=$L KBENCH_TRACE=1 timeout 100 $K 220 2 32 conv 0 | grep -E "^layer|conv_w|wave|conv_wx"
done > gpurun_out/${T}_trace.txt 2>&1
cut -c1-400 gpurun_out/${T}_trace.txt
timeout 60 online-continual-learning_amd/csrc/mfma_probe > gpurun_out/${T}_probe.txt; cat gpurun_out/${T}_probe.txt
