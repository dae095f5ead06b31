# First GPU call of round 3: is the three-buffer weight ring (conv_t_kernel<..., PIPE>) correct, and what does it buy?
#   gpurun --timeout 1500 -- 'bash scripts/gpu_ring_ab.sh r3a'
# Every step runs under its own timeout (an experimental kernel with barriers: a hang must not take the box with it).  Order: cheapest
# and most informative first -- the per-layer A/B against the reference kernel, then the clean MFMA calibration, then the parity
# tests and the bench with the ring on.
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
export TMPDIR=/tmp
T=${1:-ring}
L=gpurun_out/${T}_info.log; : > $L
K=online-continual-learning_amd/csrc/kbench
# 1. per layer: two-buffer plan, ring plan (" ring" lines), both against the reference kernel (maxdiff / statdiff / MISMATCH), SCR batch
timeout 120 $K 220 2 32 conv 0 > gpurun_out/${T}_kbench_conv_220.txt 2>&1; echo "kbench conv 220 rc=$?" >> $L
grep -c MISMATCH gpurun_out/${T}_kbench_conv_220.txt >> $L
#    replay-sized and eval-mode batches (the ASER step's passes)
timeout 120 $K 20 2 32 conv 0 > gpurun_out/${T}_kbench_conv_20.txt 2>&1; echo "kbench conv 20 rc=$?" >> $L
timeout 120 $K 410 1 32 conv 0 > gpurun_out/${T}_kbench_conv_410.txt 2>&1; echo "kbench conv 410 rc=$?" >> $L
timeout 120 $K 15 1 84 conv 0 > gpurun_out/${T}_kbench_conv_84.txt 2>&1; echo "kbench conv 15x84 rc=$?" >> $L
#    the weight-gradient kernels against kbench's reference (reldiff / MISMATCH): the baseline for work on their k loop
timeout 300 $K 220 2 32 wgrad 0 > gpurun_out/${T}_kbench_wgrad_220.txt 2>&1; echo "kbench wgrad 220 rc=$?" >> $L
#    the VGPR-bank probe (make -C online-continual-learning_amd/csrc kbench_brot beforehand): every layer, both schedules, operands of each MFMA in different banks
[ -x ${K}_brot ] && { timeout 180 ${K}_brot 220 2 32 conv 0 > gpurun_out/${T}_kbench_conv_220_brot.txt 2>&1; echo "kbench_brot conv 220 rc=$?" >> $L; }
# 2. the MFMA calibration with a clean loop (1 / 2 / 4 / 8 accumulators: the price of a dependent issue)
timeout 120 $K 220 2 32 peak > gpurun_out/${T}_kbench_peak.txt 2>&1; echo "kbench peak rc=$?" >> $L
# 3. phase trace of the ring on the staged layers
KBENCH_TRACE=1 timeout 120 $K 220 2 32 conv 0 > gpurun_out/${T}_kbench_trace.txt 2>&1; echo "kbench trace rc=$?" >> $L
# 4. parity with the ring on: network forward / backward and whole steps against the oracle
OCL_CONV_PIPE=1 timeout 900 python -m pytest tests/ -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/${T}_tests_ring.log 2>&1; echo "tests (ring) rc=$?" >> $L
#    forward outputs and gradients bit-identical with the ring on / off (whole network, five batch shapes)
OCL_TEST_RING=1 timeout 900 python -m pytest tests/test_gpu_ring.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/${T}_tests_ring_equal.log 2>&1; echo "ring == two-buffer rc=$?" >> $L
# 5. the step, ring off / on
Q="--no-cpu-baseline --no-also --no-accuracy"
for w in scr aser er mir; do
  timeout 600 python bench.py --workload $w --steps 100 --warmup 10 $Q > gpurun_out/${T}_bench_${w}_off.log 2>&1; echo "bench $w off rc=$?" >> $L
  OCL_CONV_PIPE=1 timeout 600 python bench.py --workload $w --steps 100 --warmup 10 $Q > gpurun_out/${T}_bench_${w}_ring.log 2>&1; echo "bench $w ring rc=$?" >> $L
done
cat $L
grep -h " ring\|(auto)      MT" gpurun_out/${T}_kbench_conv_220.txt | cut -c1-170 | head -60
grep -E "passed|failed" gpurun_out/${T}_tests_ring.log | tail -2
for f in gpurun_out/${T}_bench_*.log; do echo $f; tail -1 $f | cut -c1-200; done
