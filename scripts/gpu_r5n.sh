# Round 5, call 15: the projection shortcut's BatchNorm inside the block's last BatchNorm launch (z = relu(bn2(y2) + bn_s(ys)): -3 launches
# per train-mode forward) against the library of the commit before: bit-identical by construction -> checked; pass times; the network tests.
# gpurun --timeout 900 -- 'bash scripts/gpu_r5n.sh r5n'
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
T=${1:-r5n}
O=gpurun_out/${T}_out.txt
C=online-continual-learning_amd/csrc
{
  for cfg in "20 1 32 0" "13 1 32 0" "20 1 84 0" "220 2 32 1" "64 2 32 3" "100 2 32 1"; do
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
timeout 600 python -m pytest tests/test_gpu_net.py tests/test_gpu_ring.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/${T}_tests.log 2>&1; echo "tests rc=$?" >> $O
grep -E "###|differ|us per pass|rc=" $O | cut -c1-160; tail -3 gpurun_out/${T}_tests.log
