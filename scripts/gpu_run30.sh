mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
K=online-continual-learning_amd/csrc/kbench
timeout 300 $K 220 2 32 conv 0 > gpurun_out/r30_kbench.log 2>&1; grep -A1 -E "^layer1.0.conv1|^layer1.1.conv2" gpurun_out/r30_kbench.log | cut -c1-160
timeout 600 python -m pytest tests/test_gpu_net.py -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/r30_tests.log 2>&1; echo "tests rc=$?"; grep -E "^FAILED|passed|failed|Error" gpurun_out/r30_tests.log | tail -3
timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/r30_bench.log 2>&1; tail -1 gpurun_out/r30_bench.log | cut -c1-300
