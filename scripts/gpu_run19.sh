mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
K=online-continual-learning_amd/csrc/kbench
timeout 600 $K 220 2 32 conv 1 > gpurun_out/r19_kbench.log 2>&1; echo "kbench rc=$?"
grep -c MISMATCH gpurun_out/r19_kbench.log
timeout 600 python -m pytest tests/test_gpu_net.py -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/r19_net.log 2>&1; echo "net rc=$?"
grep -E "^FAILED|passed|failed|Error" gpurun_out/r19_net.log | tail -5
timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/r19_bench.log 2>&1; tail -1 gpurun_out/r19_bench.log | cut -c1-1500
