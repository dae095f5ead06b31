# Round 6: accuracy.aser -- HIP runs lower than the oracle's distribution (0.155 vs 0.217 over five seeds).  Which switch, if any?
T=${1:-r6at}
mkdir -p gpurun_out
{
for e in "X=1" "OCL_ASER_SPLIT=0" "OCL_ASER_AUTOGRAD=1" "OCL_CBRS_VERIFY_EVERY=1" "OCL_CBRS_EMULATE=0" "OCL_ASER_PIPELINE=0" "OCL_WGRAD_MULTI=0" "OCL_CONV_W=0" "OCL_ASER_SPLIT=0 OCL_ASER_AUTOGRAD=1 OCL_CBRS_VERIFY_EVERY=1 OCL_CBRS_EMULATE=0 OCL_ASER_PIPELINE=0 OCL_WGRAD_MULTI=0 OCL_CONV_W=0" "OCL_DETERMINISTIC=1"; do
  echo "### $e"
  env $e timeout -k 10 300 python scripts/aser_accuracy_probe.py 2>/dev/null | tail -1
done
} 2>&1 | tee gpurun_out/${T}_aser_accuracy_switches.txt
