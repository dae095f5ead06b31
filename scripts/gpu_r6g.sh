# Round 6, call 7: conv_w_kernel's weight copy started at a per-workgroup rotation (OCL_CW_ROT=1) against the common start.
T=${1:-r6g}
mkdir -p gpurun_out
K=online-continual-learning_amd/csrc/kbench
for E in "OCL_CW_ROT=0" "OCL_CW_ROT=2"; do
echo "### $E"
for L in layer2.1.conv1 layer3.1.conv1; do
  env $E KBENCH_ONLY=$L KBENCH_TRACE=1 timeout 100 $K 220 2 32 conv 0 | grep -E "^layer|conv_w"
done; done > gpurun_out/${T}_rot.txt 2>&1
cut -c1-330 gpurun_out/${T}_rot.txt
