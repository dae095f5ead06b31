# Round 6, call 12: conv_wx_kernel (K loop without VALU, requests spread over the rounds) against the planner's choice on every launch of the
# 220-view pass; phase traces of layers 1 - 4.
T=${1:-r6k}
mkdir -p gpurun_out
K=online-continual-learning_amd/csrc/kbench
timeout 200 $K 220 2 32 conv 0 > gpurun_out/${T}_conv220.txt 2>&1; echo "rc=$?"
for L in layer2.1.conv1 layer3.1.conv1 layer1.1.conv1 layer4.1.conv1; do
  KBENCH_ONLY=$L KBENCH_TRACE=1 timeout 100 $K 220 2 32 conv 0 | grep -E "^layer|conv_w|wave"
done > gpurun_out/${T}_trace.txt 2>&1
grep -E "^layer|^conv1|us " gpurun_out/${T}_conv220.txt | cut -c1-200
cut -c1-400 gpurun_out/${T}_trace.txt
