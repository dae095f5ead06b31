# Round 6, second session, call 3: SupCon kernels with 16-byte loads / four chains (rows) and sixteen chains + a tree for the loss (grad),
# against the library of the commit before (csrc/base/libocl_hip.so through OCL_LIB); the SupCon golden test and the SCR step tests.
# gpurun --timeout 900 -- 'bash scripts/gpu_r6bc.sh r6bc'
T=${1:-r6bc}
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
O=gpurun_out/${T}_out.txt
: > $O
for i in 1 2; do
  OCL_LIB=$PWD/online-continual-learning_amd/csrc/base/libocl_hip.so timeout 120 python scripts/supcon_time.py 2>&1 | tail -1 >> $O
  timeout 120 python scripts/supcon_time.py 2>&1 | tail -1 >> $O
done
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_steps.py -m gpu -q --tb=short -p no:cacheprovider -k "supcon or scr" > gpurun_out/${T}_tests.log 2>&1; echo "tests rc=$?" >> $O
cat $O; tail -3 gpurun_out/${T}_tests.log
