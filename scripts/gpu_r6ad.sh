# Round 6: larger row blocks (MTW) in the weight gradient of the 220-view pass -- fewer, longer workgroups with fewer LDS reads per MFMA -- through
# the planner's "enough workgroups" bound (OCL_WGRAD_ENOUGH, default 384): per layer (kbench wgrad) and through the whole pass (netcheck, two streams).
T=${1:-r6ad}
mkdir -p gpurun_out
K=online-continual-learning_amd/csrc/kbench
N=online-continual-learning_amd/csrc/netcheck
{
for e in 384 256 192 128; do
  echo "### OCL_WGRAD_ENOUGH=$e kbench 220 2 32 wgrad"
  OCL_WGRAD_ENOUGH=$e timeout -k 5 100 $K 220 2 32 wgrad 2>&1 | grep -E "wgrad|total" | cut -c1-230
  echo "### OCL_WGRAD_ENOUGH=$e netcheck 220 2 32 1"
  OCL_WGRAD_ENOUGH=$e timeout -k 5 60 $N 220 2 32 1 write /tmp/x.bin | head -1
  OCL_WGRAD_ENOUGH=$e timeout -k 5 60 $N 220 2 32 1 write /tmp/x.bin | head -1
done
} > gpurun_out/${T}_wgrad_mtw.txt 2>&1
grep -E "###|netcheck n|layer3.1.conv1|layer4.1.conv1|layer2.1.conv1|layer1.1.conv1" gpurun_out/${T}_wgrad_mtw.txt | cut -c1-220
