# Round 6: host side of the ASER step (scripts/host_cost_probe.py aser): pure host time per step, cProfile top of the issue loop.
T=${1:-r6af}
mkdir -p gpurun_out
timeout -k 10 300 python scripts/host_cost_probe.py aser > gpurun_out/${T}_host_cost_aser.txt 2>&1; echo "rc=$?"
head -60 gpurun_out/${T}_host_cost_aser.txt | cut -c1-200
