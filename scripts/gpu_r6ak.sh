# Round 6: MIR leg -- kernel sum against wall time (rocprofv3, product schedule), host cost probe, step timeline.
T=${1:-r6ak}
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
Q="--no-cpu-baseline --no-also --no-accuracy --no-roofline --preroll 0 --repeats 1"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/${T}_prof -o mir -- python bench.py --workload mir --steps 40 --warmup 10 $Q > gpurun_out/${T}_prof.log 2>&1; echo "prof rc=$?"
DB=$(find gpurun_out/${T}_prof -name "*_results.db" | head -1)
if [ -n "$DB" ]; then python scripts/rocpd_stats.py "$DB" gpurun_out/${T}_mir_kernel_stats.csv; python scripts/rocpd_timeline.py "$DB" pack_weights_kernel 3 > gpurun_out/${T}_mir_step_timeline.txt; fi
rm -rf gpurun_out/${T}_prof
timeout -k 10 300 python scripts/host_cost_probe.py mir > gpurun_out/${T}_host_cost_mir.txt 2>&1; echo "rc=$?"
head -30 gpurun_out/${T}_host_cost_mir.txt | cut -c1-180
