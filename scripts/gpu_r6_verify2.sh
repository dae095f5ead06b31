# Round 6, second session, last verification of the committed tree: the whole GPU suite, smoke(), the driver's bench command.
# gpurun --timeout 2400 -- 'bash scripts/gpu_r6_verify2.sh r6j'
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
T=${1:-r6j}
L=gpurun_out/${T}_info.log; : > $L
timeout -k 10 1500 python -m pytest tests/ -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/${T}_tests.log 2>&1; echo "tests rc=$?" >> $L
timeout -k 10 300 python __graft_entry__.py smoke > gpurun_out/${T}_smoke.log 2>&1; echo "smoke rc=$?" >> $L
timeout -k 10 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${T}_bench_scr.log 2>gpurun_out/${T}_bench_scr.err; echo "bench (driver's command) rc=$?" >> $L
cat $L; grep -E "^FAILED|passed|failed|Error" gpurun_out/${T}_tests.log | tail -5; tail -1 gpurun_out/${T}_smoke.log; tail -1 gpurun_out/${T}_bench_scr.log | cut -c1-300
