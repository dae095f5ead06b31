mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
K=online-continual-learning_amd/csrc/kbench
timeout 300 $K 220 2 32 wgrad 0 > gpurun_out/r29_kbench.log 2>&1; cut -c1-190 gpurun_out/r29_kbench.log | grep -E "layer1.0.conv1|layer2.0.conv2|layer3.0.conv2|layer4.0.conv2"
timeout 600 python -m pytest tests/test_gpu_net.py -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/r29_tests.log 2>&1; echo "tests rc=$?"; grep -E "^FAILED|passed|failed|Error" gpurun_out/r29_tests.log | tail -3
timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/r29_bench.log 2>&1; tail -1 gpurun_out/r29_bench.log | cut -c1-300
