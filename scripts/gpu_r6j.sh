# Round 6, call 11: conv_w_kernel v2 with the weight copy freed of its per-piece load + wait; and the K loop without its address arithmetic
# (kbench_fake: -DOCL_CW_FAKE_B=1, WRONG results: is the K loop's VALU what it costs?).
T=${1:-r6j}
mkdir -p gpurun_out
for B in kbench kbench_fake; do
echo "### $B"
for L in layer2.1.conv1 layer3.1.conv1 layer1.1.conv1 layer4.1.conv1; do
  KBENCH_ONLY=$L KBENCH_TRACE=1 timeout 100 online-continual-learning_amd/csrc/$B 220 2 32 conv 0 | grep -E "^layer|conv_w"
done; done > gpurun_out/${T}_trace.txt 2>&1
cut -c1-330 gpurun_out/${T}_trace.txt
