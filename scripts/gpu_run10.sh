mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
export TMPDIR=/tmp
K=online-continual-learning_amd/csrc/kbench
timeout 300 $K 220 2 32 conv 1 > gpurun_out/r10_kbench.log 2>&1; echo "kbench rc=$?"
grep -c MISMATCH gpurun_out/r10_kbench.log
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_net.py -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/r10_net.log 2>&1; echo "kernels+net rc=$?"
timeout 900 python -m pytest tests/test_gpu_steps.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/r10_steps.log 2>&1; echo "steps rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/r10_pmc1 -o k -- $K 220 2 32 conv 0 > gpurun_out/r10_pmc1.log 2>&1; echo "pmc1 rc=$?"
grep -E "^FAILED|passed|failed|Error" gpurun_out/r10_net.log | tail -8; grep -E "^FAILED|passed|failed|Error" gpurun_out/r10_steps.log | tail -8
grep -v "^    MT" gpurun_out/r10_kbench.log | cut -c1-170
