# Round 6, second session, call 5: pack_weights_kernel walking the packs (coalesced stores, gathered reads) instead of the OIHW tensors (scattered
# 4-byte stores into both packs), against the library of the commit before (csrc/base): per-kernel time under rocprofv3, bit-for-bit compare, pass times.
# gpurun --timeout 600 -- 'bash scripts/gpu_r6be.sh r6be'
T=${1:-r6be}
mkdir -p gpurun_out
export TMPDIR=/tmp
C=online-continual-learning_amd/csrc
N=$C/netcheck
O=gpurun_out/${T}_out.txt
: > $O
for V in base tree; do
  D=$C/base; [ $V = tree ] && D=online-continual-learning_amd
  for cfg in "220 2 32 1" "20 1 32 0"; do
    rm -rf /tmp/prof_$V
    LD_LIBRARY_PATH=$D OCL_SINGLE_STREAM=1 timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/prof_$V -o p -- $N $cfg write /tmp/x_$V.bin > /tmp/log_$V.txt 2>&1
    DB=$(find /tmp/prof_$V -name "*_results.db" | head -1)
    python scripts/rocpd_stats.py $DB /tmp/stats_$V.csv > /dev/null 2>&1
    echo "### $V, netcheck $cfg under rocprofv3: $(grep pack_weights /tmp/stats_$V.csv | cut -d'"' -f3 | cut -d, -f2-4) (calls, total ns, average ns of pack_weights_kernel)" >> $O
  done
done
for cfg in "220 2 32 1" "20 1 32 0" "6 1 84 0"; do
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
