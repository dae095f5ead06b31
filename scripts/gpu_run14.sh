mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
export TMPDIR=/tmp
L=gpurun_out/r14_info.log; : > $L
K=online-continual-learning_amd/csrc/kbench
timeout 300 $K 220 2 32 wgrad 0 > gpurun_out/r14_kbench.log 2>&1; echo "kbench rc=$?" >> $L
timeout 300 $K 220 2 32 bn 0 >> gpurun_out/r14_kbench.log 2>&1
timeout 900 python -m pytest tests/test_gpu_net.py tests/test_gpu_steps.py -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/r14_tests.log 2>&1; echo "tests rc=$?" >> $L
timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/r14_bench.log 2>&1; echo "bench rc=$?" >> $L
cat $L; grep -E "^FAILED|passed|failed|Error" gpurun_out/r14_tests.log | tail -8; tail -1 gpurun_out/r14_bench.log | cut -c1-1800; cat gpurun_out/r14_kbench.log | cut -c1-200
