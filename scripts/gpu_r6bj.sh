# Round 6: the HIP side's spread of accuracy.aser's end accuracy at a FIXED seed under a one-ulp perturbation of the initial weights
# (ten perturbations at seeds 0 and 100; run 0 of each seed is unperturbed) -- counterpart of scripts/aser_oracle_chaos_probe.py.
T=${1:-r6bj}
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
O=gpurun_out/${T}_aser_fixed_seed_spread.txt
: > $O
for S in 0 100; do
  echo "### seed $S, perturbations 0..9 (0 = none)" >> $O
  PROBE_SEEDS=$S PROBE_PERT=10 timeout -k 5 150 python scripts/aser_accuracy_probe.py 2>/dev/null >> $O; echo "rc=$?" >> $O
done
cat $O
