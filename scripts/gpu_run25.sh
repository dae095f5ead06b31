mkdir -p gpurun_out
K=online-continual-learning_amd/csrc/kbench
timeout 600 $K 220 2 32 wgrad 1 > gpurun_out/r25_wgrad.log 2>&1; echo "wgrad rc=$?"
timeout 120 $K 220 2 32 bn 0 > gpurun_out/r25_bn.log 2>&1; cat gpurun_out/r25_bn.log
