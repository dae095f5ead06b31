# Planner knobs of the weight gradient under the table-driven staging, whole-pass time (netcheck): largest pixel tile x workgroup target.
# gpurun --timeout 60 -- 'bash scripts/gpu_r4z5.sh'
mkdir -p gpurun_out
cd online-continual-learning_amd/csrc
O=../../gpurun_out/r4z5_wgrad_knobs_netcheck.txt
{
  for cfg in "220 2 32 1" "20 1 32 0"; do
    for KP in 128 64; do for T in 384 512 768; do
      echo -n "KP=$KP T=$T  "
      OCL_WGRAD_KP=$KP OCL_WGRAD_TARGET=$T timeout 20 ./netcheck $cfg write /tmp/x.bin | head -1
    done; done
  done
} > $O 2>&1
cat $O
