mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
L=gpurun_out/r8_info.log; : > $L
timeout 300 online-continual-learning_amd/csrc/kbench 220 2 32 all 1 > gpurun_out/r8_kbench.log 2>&1; echo "kbench rc=$?" >> $L
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_net.py -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/r8_net.log 2>&1; echo "kernels+net rc=$?" >> $L
timeout 900 python -m pytest tests/test_gpu_steps.py -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/r8_steps.log 2>&1; echo "steps rc=$?" >> $L
timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/r8_bench.log 2>&1; echo "bench rc=$?" >> $L
cat $L; grep -E "^FAILED|passed|failed|Error" gpurun_out/r8_net.log | tail -8; grep -E "^FAILED|passed|failed|Error" gpurun_out/r8_steps.log | tail -8; tail -2 gpurun_out/r8_bench.log | cut -c1-1500
