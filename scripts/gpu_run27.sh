mkdir -p gpurun_out
K=online-continual-learning_amd/csrc/kbench
timeout 300 $K 220 2 32 conv 0 > gpurun_out/r27_kbench.log 2>&1
grep -E "^conv1|layer1.0.conv1|layer2.0.conv2|layer3.0.conv2|layer4.0.conv2|layer4.0.conv1 +fwd" gpurun_out/r27_kbench.log | cut -c1-150
