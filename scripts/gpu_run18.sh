mkdir -p gpurun_out
timeout 120 online-continual-learning_amd/csrc/kbench 220 2 32 peak 0 > gpurun_out/r18_peak.log 2>&1; cat gpurun_out/r18_peak.log
