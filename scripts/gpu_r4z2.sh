# The experimental 4x4x1 weight-gradient form through the WHOLE pass, C-ABI only (netcheck: no torch, ~1 s per run):
# gradient / output / running statistics of a training forward + backward with OCL_WGRAD_Q=1 against the default form, tensor by tensor,
# in the order-independent sum mode (so every bit that differs is the weight-gradient form's), plus the pass time of both.
# gpurun --timeout 120 -- 'bash scripts/gpu_r4z2.sh'
mkdir -p gpurun_out
cd online-continual-learning_amd/csrc
O=../../gpurun_out/r4z2_netcheck_wgrad_q.txt
{
  for cfg in "220 2 32 1" "20 1 32 0" "13 1 32 0" "20 1 84 0"; do
    echo "### netcheck $cfg   (n groups hw head)"
    OCL_DETERMINISTIC=1 timeout 30 ./netcheck $cfg write /tmp/ref.bin
    OCL_DETERMINISTIC=1 timeout 30 ./netcheck $cfg compare /tmp/ref.bin | tail -2
    echo "# OCL_WGRAD_Q=1:"
    OCL_DETERMINISTIC=1 OCL_WGRAD_Q=1 timeout 30 ./netcheck $cfg compare /tmp/ref.bin
    echo "# default sums, pass time: default / OCL_WGRAD_Q=1"
    timeout 30 ./netcheck $cfg write /tmp/ref2.bin | head -1
    OCL_WGRAD_Q=1 timeout 30 ./netcheck $cfg compare /tmp/ref2.bin | grep -E "netcheck|MISMATCH|NaN|beyond"
  done
  echo "### kbench wgrad, OCL_WGRAD_Q=1, other shapes"
  OCL_WGRAD_Q=1 timeout 40 ./kbench 13 1 32 wgrad 2>&1 | grep -E "^layer1|MISMATCH|rror"
  OCL_WGRAD_Q=1 timeout 40 ./kbench 20 1 84 wgrad 2>&1 | grep -E "^layer1|MISMATCH|rror"
  timeout 40 ./kbench 20 1 84 wgrad 2>&1 | grep -E "^layer1.0.conv1|MISMATCH|rror"
} > $O 2>&1
cat $O | cut -c1-200
