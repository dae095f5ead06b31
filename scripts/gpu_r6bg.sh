# Round 6, second session, call 8: bn_bwd_chan_kernel (BatchNorm backward of a small map partitioned by channel quad: no atomics, no grid-wide arrival)
# on layers 3 - 4 of replay-sized passes, against the library of the commit before (csrc/base): per-kernel averages under rocprofv3, netcheck
# compare (tolerance 1e-4 of a tensor's largest entry: the sums are taken in another order) and pass times; then the parity suites.
# gpurun --timeout 1500 -- 'bash scripts/gpu_r6bg.sh r6bg'
T=${1:-r6bg}
mkdir -p gpurun_out
export TMPDIR=/tmp
export PYTHONDONTWRITEBYTECODE=1
C=online-continual-learning_amd/csrc
N=$C/netcheck
O=gpurun_out/${T}_out.txt
: > $O
for V in base tree; do
  D=$C/base; [ $V = tree ] && D=online-continual-learning_amd
  rm -rf /tmp/prof_$V
  LD_LIBRARY_PATH=$D timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/prof_$V -o p -- $N 20 2 32 0 write /tmp/x_$V.bin > /tmp/log_$V.txt 2>&1
  DB=$(find /tmp/prof_$V -name "*_results.db" | head -1)
  python scripts/rocpd_stats.py $DB /tmp/stats_$V.csv > /dev/null 2>&1
  echo "### $V, netcheck 20 2 32 0 under rocprofv3 (name, calls, total ns, average ns):" >> $O
  grep "bn_bwd_fused\|bn_bwd_chan" /tmp/stats_$V.csv | cut -d, -f1-6 >> $O
done
for cfg in "20 2 32 0" "20 1 32 0" "13 1 32 0" "6 1 84 0" "64 2 32 3"; do
  echo "### netcheck $cfg, deterministic sums: base library -> file; the tree compared" >> $O
  LD_LIBRARY_PATH=$C/base OCL_DETERMINISTIC=1 timeout 60 $N $cfg write /tmp/ref.bin 2>&1 | head -1 >> $O
  OCL_DETERMINISTIC=1 timeout 60 $N $cfg compare /tmp/ref.bin 2>&1 | tail -3 >> $O
  echo "# pass time, default sums: base / tree, three times" >> $O
  for i in 1 2 3; do
    LD_LIBRARY_PATH=$C/base timeout 60 $N $cfg write /tmp/ref2.bin 2>&1 | head -1 >> $O
    timeout 60 $N $cfg write /tmp/ref3.bin 2>&1 | head -1 >> $O
  done
done
timeout 1200 python -m pytest tests/test_gpu_net.py tests/test_gpu_ring.py tests/test_gpu_netcheck.py tests/test_gpu_steps.py -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/${T}_tests.log 2>&1; echo "tests rc=$?" >> $O
cat $O; tail -5 gpurun_out/${T}_tests.log
