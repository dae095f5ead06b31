# Round 6, second session, call 7: bn_bwd_fused_kernel's grid-wide arrival on ONE counter for grids of <= 40 workgroups (replay-sized passes:
# 8 - 33) instead of sub-counter -> master, against the library of the commit before (csrc/base): netcheck pass times, bit-for-bit compare,
# per-kernel averages under rocprofv3.
# gpurun --timeout 600 -- 'bash scripts/gpu_r6bf.sh r6bf'
T=${1:-r6bf}
mkdir -p gpurun_out
export TMPDIR=/tmp
C=online-continual-learning_amd/csrc
N=$C/netcheck
O=gpurun_out/${T}_out.txt
: > $O
for V in base tree; do
  D=$C/base; [ $V = tree ] && D=online-continual-learning_amd
  rm -rf /tmp/prof_$V
  LD_LIBRARY_PATH=$D timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/prof_$V -o p -- $N 20 1 32 0 write /tmp/x_$V.bin > /tmp/log_$V.txt 2>&1
  DB=$(find /tmp/prof_$V -name "*_results.db" | head -1)
  python scripts/rocpd_stats.py $DB /tmp/stats_$V.csv > /dev/null 2>&1
  echo "### $V, netcheck 20 1 32 0 under rocprofv3 (name, calls, total ns, average ns):" >> $O
  grep bn_bwd_fused /tmp/stats_$V.csv | cut -d, -f1-6 >> $O
done
for cfg in "20 1 32 0" "13 1 32 0" "64 2 32 3" "6 1 84 0"; do
  echo "### netcheck $cfg, deterministic sums: base library -> file; the tree compared" >> $O
  LD_LIBRARY_PATH=$C/base OCL_DETERMINISTIC=1 timeout 60 $N $cfg write /tmp/ref.bin 2>&1 | head -1 >> $O
  OCL_DETERMINISTIC=1 timeout 60 $N $cfg compare /tmp/ref.bin 2>&1 | tail -2 >> $O
  echo "# pass time, default sums: base / tree, three times" >> $O
  for i in 1 2 3; do
    LD_LIBRARY_PATH=$C/base timeout 60 $N $cfg write /tmp/ref2.bin 2>&1 | head -1 >> $O
    timeout 60 $N $cfg write /tmp/ref3.bin 2>&1 | head -1 >> $O
  done
done
cat $O
