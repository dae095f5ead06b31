# Round 5, call 14: MFMA-pipe utilisation of the product's kernels by hardware counters (one PMC pass over the SCR bench, single stream;
# counters alone with --kernel-trace, as the pool requires), summarised per kernel by scripts/pmc_mfma.py.
# gpurun --timeout 600 -- 'bash scripts/gpu_r5m.sh r5m'
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
export TMPDIR=/tmp
T=${1:-r5m}
Q="--no-cpu-baseline --no-also --no-accuracy --no-roofline --preroll 0 --repeats 1 --single-stream"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d gpurun_out/${T}_pmc -o p -- python bench.py --steps 10 --warmup 3 $Q > gpurun_out/${T}_pmc.log 2>&1; echo "pmc rc=$?"
f=$(find gpurun_out/${T}_pmc -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python scripts/pmc_mfma.py $(dirname $f) gpurun_out/${T}_pmc_mfma.txt
rm -rf gpurun_out/${T}_pmc
grep -E "conv_|wgrad" gpurun_out/${T}_pmc_mfma.txt | cut -c1-260 | head -40
