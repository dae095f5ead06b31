# Round 6: timeline of one SCR step on the product's two streams (which kernels run after the dependent chain has ended).
T=${1:-r6z}
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
Q="--no-cpu-baseline --no-also --no-accuracy --no-roofline --preroll 0 --repeats 1"
timeout -k 10 240 rocprofv3 --kernel-trace -d gpurun_out/${T}_prof -o scr -- python bench.py --steps 30 --warmup 10 $Q > gpurun_out/${T}_prof.log 2>&1; echo "prof rc=$?"
DB=$(find gpurun_out/${T}_prof -name "*_results.db" | head -1)
if [ -n "$DB" ]; then python scripts/rocpd_timeline.py "$DB" pack_weights_kernel 3 > gpurun_out/${T}_scr_step_timeline.txt; python scripts/rocpd_stats.py "$DB" gpurun_out/${T}_scr_kernel_stats.csv; fi
rm -rf gpurun_out/${T}_prof
tail -40 gpurun_out/${T}_scr_step_timeline.txt
