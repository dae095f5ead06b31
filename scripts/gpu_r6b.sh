# Round 6, call 2: conv_w_kernel (convw.hip) against the planner's choice, per convolution launch, at the SCR batch and a replay-sized batch.
# gpurun --timeout 420 -- 'bash scripts/gpu_r6b.sh r6b'
T=${1:-r6b}
mkdir -p gpurun_out
K=online-continual-learning_amd/csrc/kbench
timeout 200 $K 220 2 32 conv 0 > gpurun_out/${T}_conv220.txt 2>&1; echo "rc=$?"
timeout 100 $K 20 1 32 conv 0 > gpurun_out/${T}_conv20.txt 2>&1; echo "rc=$?"
grep -E "^layer|^conv1|us " gpurun_out/${T}_conv220.txt | cut -c1-200
