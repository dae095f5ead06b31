# Round 6, call 3: phase trace of conv_w_kernel on the layer-2 / layer-3 / layer-1 launches of the 220-view pass.
# gpurun --timeout 300 -- 'bash scripts/gpu_r6c.sh r6c'
T=${1:-r6c}
mkdir -p gpurun_out
K=online-continual-learning_amd/csrc/kbench
for L in layer2.1.conv1 layer3.1.conv1 layer1.1.conv1; do
  KBENCH_ONLY=$L KBENCH_TRACE=1 timeout 100 $K 220 2 32 conv 0
done > gpurun_out/${T}_trace.txt 2>&1
grep -E "conv_w|^layer|wg " gpurun_out/${T}_trace.txt | cut -c1-330
