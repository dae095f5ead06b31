# Round 6, second session, call 1: three prologue changes, each as a library variant of its own (csrc/var_*/libocl_hip.so, built with -D
# switches of conv.hip; base = the library of commit 28fddec) through netcheck under rocprofv3 (single stream: per-kernel durations of the
# kernel alone), 220-view SCR pass; then pass times of the 220-view and 20-image passes, base against the tree, and the bit-for-bit compare.
#   s  : conv_s_kernel's input-transform table for the tile's own BatchNorm group only, two replica loads in flight (was: every group, one)
#   f2 / f4 : bn_fwd_kernel requests a thread's first 2 / 4 units before its table;  a2 / a4 : the same in bn_bwd_apply_e_kernel
# gpurun --timeout 900 -- 'bash scripts/gpu_r6ba.sh r6ba'
T=${1:-r6ba}
mkdir -p gpurun_out
export TMPDIR=/tmp
C=online-continual-learning_amd/csrc
N=$C/netcheck
O=gpurun_out/${T}_out.txt
: > $O
for V in base s f2 f4 a2 a4 tree; do
  D=$C/var_$V; [ $V = base ] && D=$C/base; [ $V = tree ] && D=online-continual-learning_amd
  rm -rf /tmp/prof_$V
  LD_LIBRARY_PATH=$D OCL_SINGLE_STREAM=1 timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/prof_$V -o p -- $N 220 2 32 1 write /tmp/x_$V.bin > gpurun_out/${T}_$V.log 2>&1
  DB=$(find /tmp/prof_$V -name "*_results.db" | head -1)
  python scripts/rocpd_stats.py $DB gpurun_out/${T}_$V.csv > /dev/null 2>&1
  echo "### $V: $(grep -h 'forward + backward' gpurun_out/${T}_$V.log | head -1)" >> $O
  grep -E "conv_s_kernel<1, false, false, false>|bn_fwd_kernel|bn_bwd_apply_e_kernel" gpurun_out/${T}_$V.csv | cut -d, -f1-4 | sed 's/"//g' >> $O
done
for cfg in "220 2 32 1" "20 1 32 0" "64 2 32 3"; do
  echo "### netcheck $cfg, deterministic sums: base library -> file; the tree compared" >> $O
  LD_LIBRARY_PATH=$C/base OCL_DETERMINISTIC=1 timeout 60 $N $cfg write /tmp/ref.bin 2>&1 | head -1 >> $O
  OCL_DETERMINISTIC=1 timeout 60 $N $cfg compare /tmp/ref.bin 2>&1 | tail -2 >> $O
  echo "# pass time, default sums, two streams: base / tree, three times" >> $O
  for i in 1 2 3; do
    LD_LIBRARY_PATH=$C/base timeout 60 $N $cfg write /tmp/ref2.bin 2>&1 | head -1 >> $O
    timeout 60 $N $cfg write /tmp/ref3.bin 2>&1 | head -1 >> $O
  done
done
cut -c1-200 $O
