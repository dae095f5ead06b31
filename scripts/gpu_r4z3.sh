# Weight-gradient staging with precomputed unit tables (OCL_WGRAD_TAB=1) against the default form: per layer in kbench (reference
# kernel + time), and through the whole pass in netcheck (bit-for-bit in the order-independent sum mode, + pass time).
# gpurun --timeout 150 -- 'bash scripts/gpu_r4z3.sh'
mkdir -p gpurun_out
cd online-continual-learning_amd/csrc
O=../../gpurun_out/r4z3_wgrad_tab.txt
{
  for E in 0 1; do
    echo "### OCL_WGRAD_TAB=$E ./kbench 220 2 32 wgrad"
    OCL_WGRAD_TAB=$E timeout 40 ./kbench 220 2 32 wgrad 2>&1 | grep -E "wgrad |MISMATCH|rror" | cut -c1-20,96-240
  done
  echo "### OCL_WGRAD_TAB=1 OCL_WGRAD_Q=1 ./kbench 220 2 32 wgrad"
  OCL_WGRAD_TAB=1 OCL_WGRAD_Q=1 timeout 40 ./kbench 220 2 32 wgrad 2>&1 | grep -E "^layer1|MISMATCH|rror" | cut -c1-20,96-240
  for cfg in "220 2 32 1" "20 1 32 0" "13 1 32 0" "20 1 84 0"; do
    echo "### netcheck $cfg   (n groups hw head): default -> file; OCL_WGRAD_TAB=1 compared"
    OCL_DETERMINISTIC=1 timeout 30 ./netcheck $cfg write /tmp/ref.bin | head -1
    OCL_DETERMINISTIC=1 OCL_WGRAD_TAB=1 timeout 30 ./netcheck $cfg compare /tmp/ref.bin
    echo "# default sums, pass time: default / OCL_WGRAD_TAB=1 / OCL_WGRAD_TAB=1 OCL_WGRAD_Q=1"
    timeout 30 ./netcheck $cfg write /tmp/ref2.bin | head -1
    OCL_WGRAD_TAB=1 timeout 30 ./netcheck $cfg compare /tmp/ref2.bin | grep -E "netcheck|MISMATCH|NaN|beyond"
    OCL_WGRAD_TAB=1 OCL_WGRAD_Q=1 timeout 30 ./netcheck $cfg compare /tmp/ref2.bin | grep -E "netcheck|MISMATCH|NaN|beyond"
  done
} > $O 2>&1
cat $O | cut -c1-200
