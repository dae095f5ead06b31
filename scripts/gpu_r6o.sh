# Round 6, call 20: per-kernel times of the whole 220-view pass (netcheck under rocprofv3), planner default (conv_w on) against OCL_CONV_W=0,
# single stream (OCL_SINGLE_STREAM=1) and the product's two streams.
T=${1:-r6o}
mkdir -p gpurun_out
export TMPDIR=/tmp
N=online-continual-learning_amd/csrc/netcheck
for W in 0 1; do for S in 1 0; do
  rm -rf /tmp/prof_$W$S
  OCL_CONV_W=$W OCL_SINGLE_STREAM=$S timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/prof_$W$S -o p -- $N 220 2 32 1 write /tmp/x.bin > gpurun_out/${T}_w${W}_s${S}.log 2>&1
  DB=$(find /tmp/prof_$W$S -name "*_results.db" | head -1)
  python scripts/rocpd_stats.py $DB gpurun_out/${T}_w${W}_single${S}.csv
done; done
grep -h "forward + backward" gpurun_out/${T}_w*.log
