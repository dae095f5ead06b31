mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
export TMPDIR=/tmp
K=online-continual-learning_amd/csrc/kbench
timeout 600 $K 220 2 32 conv 1 > gpurun_out/r12_kbench.log 2>&1; echo "kbench rc=$?"
grep -c MISMATCH gpurun_out/r12_kbench.log
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_net.py -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/r12_net.log 2>&1; echo "kernels+net rc=$?"
grep -E "^FAILED|passed|failed|Error" gpurun_out/r12_net.log | tail -8
grep -v "^    MT" gpurun_out/r12_kbench.log | cut -c1-170
