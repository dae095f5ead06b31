# The table-driven weight-gradient staging as the DEFAULT against OCL_WGRAD_TAB=0 over more shapes (whole pass, bit for bit, C-ABI only).
# gpurun --timeout 120 -- 'bash scripts/gpu_r4z4.sh'
mkdir -p gpurun_out
cd online-continual-learning_amd/csrc
O=../../gpurun_out/r4z4_wgrad_tab_shapes.txt
{
  for cfg in "1 1 32 0" "3 1 32 0" "7 1 32 0" "50 1 32 0" "64 2 32 1" "220 2 32 3" "10 1 84 0" "50 1 84 0" "6 2 84 1"; do
    echo "### netcheck $cfg   (n groups hw head): OCL_WGRAD_TAB=0 -> file; default compared"
    OCL_DETERMINISTIC=1 OCL_WGRAD_TAB=0 timeout 30 ./netcheck $cfg write /tmp/ref.bin | head -1
    OCL_DETERMINISTIC=1 timeout 30 ./netcheck $cfg compare /tmp/ref.bin
  done
  echo "### OCL_WGRAD_Q=1 (gated by tiles per workgroup): 220 views use it, 20 images of 84 x 84 do not"
  timeout 30 ./netcheck 220 2 32 1 write /tmp/ref.bin | head -1
  OCL_WGRAD_Q=1 timeout 30 ./netcheck 220 2 32 1 compare /tmp/ref.bin | grep -E "netcheck|differ"
  timeout 30 ./netcheck 20 1 84 0 write /tmp/ref.bin | head -1
  OCL_WGRAD_Q=1 timeout 30 ./netcheck 20 1 84 0 compare /tmp/ref.bin | grep -E "netcheck|differ"
} > $O 2>&1
cat $O | cut -c1-200
