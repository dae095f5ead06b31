# Round 5, call 13: kbench tiling sweep of every convolution launch at the SCR batch and at a replay-sized batch (is any forced tiling
# better than the planner's choice on this tree?).
# gpurun --timeout 600 -- 'bash scripts/gpu_r5l.sh r5l'
mkdir -p gpurun_out
T=${1:-r5l}
K=online-continual-learning_amd/csrc/kbench
timeout 240 $K 220 2 32 conv 1 > gpurun_out/${T}_sweep220.txt 2>&1; echo "rc=$?"
timeout 120 $K 20 1 32 conv 1 > gpurun_out/${T}_sweep20.txt 2>&1; echo "rc=$?"
wc -l gpurun_out/${T}_sweep220.txt gpurun_out/${T}_sweep20.txt
