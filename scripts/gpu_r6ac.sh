# Round 6: where a workgroup's time goes in the weight gradient of a 20-image pass, split as for a launch of its own (512 workgroups) and as inside
# the merged launch (96): kbench 20 2 32 wgradtrace, ticks of the 100 MHz counter.
T=${1:-r6ac}
mkdir -p gpurun_out
K=online-continual-learning_amd/csrc/kbench
{
for t in 0 96 32; do
  echo "### KBENCH_WG_TARGET=$t kbench 20 2 32 wgradtrace"
  KBENCH_WG_TARGET=$t timeout -k 5 100 $K 20 2 32 wgradtrace 2>&1 | grep -v "^#"
done
} > gpurun_out/${T}_wgradtrace20.txt 2>&1
cut -c1-330 gpurun_out/${T}_wgradtrace20.txt
