# Round 6: timeline of one ASER step on the product schedule (where the GPU idles: the step's one synchronisation, host-side plugin logic).
T=${1:-r6ae}
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
Q="--no-cpu-baseline --no-also --no-accuracy --no-roofline --preroll 0 --repeats 1"
timeout -k 10 240 rocprofv3 --kernel-trace -d gpurun_out/${T}_prof -o aser -- python bench.py --workload aser --steps 40 --warmup 10 $Q > gpurun_out/${T}_prof.log 2>&1; echo "prof rc=$?"
DB=$(find gpurun_out/${T}_prof -name "*_results.db" | head -1)
if [ -n "$DB" ]; then python scripts/rocpd_timeline.py "$DB" pack_weights_kernel 3 > gpurun_out/${T}_aser_step_timeline.txt; fi
rm -rf gpurun_out/${T}_prof
grep -c . gpurun_out/${T}_aser_step_timeline.txt; tail -4 gpurun_out/${T}_aser_step_timeline.txt
