# Round 6, last call: the HIP side of accuracy.aser at TEN seeds (the oracle's ten: scripts/aser_accuracy_oracle10.sh on the CPU), default
# sums and order-independent sums -- paired with the oracle's per-seed values and with the oracle's own fixed-seed spread under a one-ulp
# perturbation (scripts/aser_oracle_chaos_probe.py).
T=${1:-r6bi}
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
S=0,100,200,300,400,500,600,700,800,900
O=gpurun_out/${T}_aser_acc10.txt
: > $O
for E in "OCL_NONE=1" "OCL_DETERMINISTIC=1" "OCL_NONE=2"; do
  echo "### $E seeds $S" >> $O
  env $E PROBE_SEEDS=$S timeout -k 5 150 python scripts/aser_accuracy_probe.py 2>/dev/null >> $O; echo "rc=$?" >> $O
done
cat $O
