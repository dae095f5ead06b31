# Round 6, second session, call 4: the mlp head's backward on two streams -- both layers' parameter gradients (and the zero fill of the unused
# encoder.linear gradient) handed to the side stream behind ONE event after the chain's own launches (was: fill on the chain, one event per layer)
# -- against the library of the commit before (csrc/base2), netcheck 220 2 32 1, alternating; bit-for-bit compare under deterministic sums;
# the two-stream / netcheck A/B tests.
# gpurun --timeout 900 -- 'bash scripts/gpu_r6bd.sh r6bd'
T=${1:-r6bd}
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
C=online-continual-learning_amd/csrc
N=$C/netcheck
O=gpurun_out/${T}_out.txt
: > $O
for cfg in "220 2 32 1" "128 2 32 1"; do
  echo "### netcheck $cfg, deterministic sums: base2 library -> file; the tree compared" >> $O
  LD_LIBRARY_PATH=$C/base2 OCL_DETERMINISTIC=1 timeout 60 $N $cfg write /tmp/ref.bin 2>&1 | head -1 >> $O
  OCL_DETERMINISTIC=1 timeout 60 $N $cfg compare /tmp/ref.bin 2>&1 | tail -2 >> $O
  echo "# pass time, default sums, two streams: base2 / tree, four times" >> $O
  for i in 1 2 3 4; do
    LD_LIBRARY_PATH=$C/base2 timeout 60 $N $cfg write /tmp/ref2.bin 2>&1 | head -1 >> $O
    timeout 60 $N $cfg write /tmp/ref3.bin 2>&1 | head -1 >> $O
  done
done
timeout 600 python -m pytest tests/test_gpu_ring.py tests/test_gpu_netcheck.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/${T}_tests.log 2>&1; echo "tests rc=$?" >> $O
cat $O; tail -3 gpurun_out/${T}_tests.log
