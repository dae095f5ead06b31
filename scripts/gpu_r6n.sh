# Round 6, call 19: conv_w_kernel in the product (planner default): whole training pass through netcheck against OCL_CONV_W=0 (tensor by tensor,
# order-independent sums), pass times, then the GPU test suite.
T=${1:-r6n}
mkdir -p gpurun_out
N=online-continual-learning_amd/csrc/netcheck
{
  for cfg in "220 2 32 1" "100 1 32 0" "128 2 32 3"; do
    echo "### netcheck $cfg"
    OCL_CONV_W=0 OCL_DETERMINISTIC=1 timeout 60 $N $cfg write /tmp/ref.bin | head -1
    OCL_DETERMINISTIC=1 timeout 60 $N $cfg compare /tmp/ref.bin; echo "rc=$?"
    echo "# pass time: OCL_CONV_W=0 / default / default"
    OCL_CONV_W=0 timeout 60 $N $cfg write /tmp/ref2.bin | head -1
    timeout 60 $N $cfg compare /tmp/ref2.bin | grep -E "netcheck|beyond"
    timeout 60 $N $cfg compare /tmp/ref2.bin | grep -E "netcheck"
  done
} > gpurun_out/${T}_netcheck.txt 2>&1
cat gpurun_out/${T}_netcheck.txt | cut -c1-200
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/${T}_pytest.txt 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/${T}_pytest.txt
