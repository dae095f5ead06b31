# Round 6, call 17: what do the staging slots of conv_wx_kernel's K loop cost?  kbench_slot0: no slot, slot1: LDS stores only, slot2: loads only (WRONG results), kbench: both
T=${1:-r6m}
mkdir -p gpurun_out
for B in kbench_slot0 kbench_slot1 kbench_slot2 kbench; do
echo "### $B"
for L in layer2.1.conv1 layer3.1.conv1 layer1.1.conv1 layer4.1.conv1; do
  KBENCH_ONLY=$L timeout 100 online-continual-learning_amd/csrc/$B 220 2 32 conv 0 | grep -E "conv_w"
done; done > gpurun_out/${T}_slots.txt 2>&1
cut -c1-300 gpurun_out/${T}_slots.txt
