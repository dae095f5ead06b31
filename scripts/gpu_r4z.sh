# Shots at the end of round 4 (~16 s of box time each): the experimental 4x4x1 weight-gradient form against the reference kernel of kbench,
# next to the default form, layer 1 + stem at the SCR batch.  gpurun --timeout 120 -- 'bash scripts/gpu_r4z.sh'
mkdir -p gpurun_out
cd online-continual-learning_amd/csrc
O=../../gpurun_out/r4z_wgrad_q.txt
F='^conv1|^layer1.0.conv1|MISMATCH|rror'
{
  echo "### default ./kbench 220 2 32 wgrad"
  timeout 40 ./kbench 220 2 32 wgrad 2>&1 | grep -E "$F"
  for T in 128 256 512; do for R in 2 3; do
    echo "### OCL_WGRAD_Q=1 OCL_WGRAD_Q_TARGET=$T OCL_WGRAD_Q_RGW=$R"
    OCL_WGRAD_Q=1 OCL_WGRAD_Q_TARGET=$T OCL_WGRAD_Q_RGW=$R timeout 40 ./kbench 220 2 32 wgrad 2>&1 | grep -E "$F"
  done; done
  echo "### OCL_WGRAD_Q=1 at 20 images"
  OCL_WGRAD_Q=1 timeout 40 ./kbench 20 1 32 wgrad 2>&1 | grep -E "$F"
  echo "### default at 20 images"
  timeout 40 ./kbench 20 1 32 wgrad 2>&1 | grep -E "$F"
} > $O 2>&1
cat $O
