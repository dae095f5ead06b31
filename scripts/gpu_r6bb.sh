# Round 6, second session, call 2: UPPER BOUNDS of two launch folds the reviews asked for (items 3, 4), by leaving the launches out
# (timing only, results wrong): OCL_DEBUG_SKIP_BN2FWD=1 = no bn_fwd_kernel behind conv2 of the seven non-final blocks;
# OCL_DEBUG_SKIP_SHORTCUT=1 = no projection-shortcut convolution / data gradient / weight gradient.  netcheck (whole training pass through the
# C-ABI, product schedule: two streams on the 220-view pass), alternating, three rounds.
# gpurun --timeout 600 -- 'bash scripts/gpu_r6bb.sh r6bb'
T=${1:-r6bb}
mkdir -p gpurun_out
N=online-continual-learning_amd/csrc/netcheck
O=gpurun_out/${T}_out.txt
: > $O
for cfg in "220 2 32 1" "20 1 32 0" "64 2 32 3" "6 1 84 0"; do
  echo "### netcheck $cfg" >> $O
  for i in 1 2 3; do
    for E in "X=0" "OCL_DEBUG_SKIP_BN2FWD=1" "OCL_DEBUG_SKIP_SHORTCUT=1" "OCL_DEBUG_SKIP_BN2FWD=1 OCL_DEBUG_SKIP_SHORTCUT=1" "OCL_SINGLE_STREAM=1" "OCL_SINGLE_STREAM=1 OCL_DEBUG_SKIP_BN2FWD=1" "OCL_SINGLE_STREAM=1 OCL_DEBUG_SKIP_SHORTCUT=1"; do
      echo "$E: $(env $E timeout 60 $N $cfg write /tmp/x.bin 2>&1 | head -1 | sed 's/.*forward + backward//')" >> $O
    done
  done
done
cat $O
