# Round 5, call 7: few-slab weight-gradient reduction (S <= 8: one thread per output) against the library of the commit before
# (csrc/base/libocl_hip.so, picked up through LD_LIBRARY_PATH): bit-identical by construction -> checked; pass times; SupCon tests.
# gpurun --timeout 600 -- 'bash scripts/gpu_r5f.sh r5f'
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
T=${1:-r5f}
O=gpurun_out/${T}_out.txt
C=online-continual-learning_amd/csrc
{
  for cfg in "20 1 32 0" "13 1 32 0" "20 1 84 0" "220 2 32 1" "64 2 32 3"; do
    echo "### netcheck $cfg: base library -> file; this tree compared (order-independent sums: must be bit-identical)"
    LD_LIBRARY_PATH=$C/base OCL_DETERMINISTIC=1 timeout 60 $C/netcheck $cfg write /tmp/ref.bin | head -1
    OCL_DETERMINISTIC=1 timeout 60 $C/netcheck $cfg compare /tmp/ref.bin | tail -2
    echo "# pass time, default sums: base / this tree / base / this tree"
    for i in 1 2; do
      LD_LIBRARY_PATH=$C/base timeout 60 $C/netcheck $cfg write /tmp/ref2.bin | head -1
      timeout 60 $C/netcheck $cfg write /tmp/ref3.bin | head -1
    done
  done
} > $O 2>&1
timeout 400 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_steps.py -m gpu -q --tb=short -p no:cacheprovider -k "supcon or scr" > gpurun_out/${T}_tests.log 2>&1; echo "tests rc=$?" >> $O
cat $O | cut -c1-200; tail -3 gpurun_out/${T}_tests.log
