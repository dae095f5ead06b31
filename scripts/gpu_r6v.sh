# Round 6: rocprofv3 kernel stats (single stream) of the ER and ASER legs with the merged weight-gradient launch.
# gpurun --timeout 900 -- 'bash scripts/gpu_r6v.sh r6v'
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
export TMPDIR=/tmp
T=${1:-r6v}
Q="--no-cpu-baseline --no-also --no-accuracy --no-roofline --preroll 0 --repeats 1"
for wl in er aser; do
  timeout -k 10 240 rocprofv3 --kernel-trace --stats -d gpurun_out/${T}_prof_$wl -o $wl -- python bench.py --workload $wl --steps 50 --warmup 10 $Q --single-stream > gpurun_out/${T}_prof_$wl.log 2>&1; echo "prof $wl rc=$?"
  DB=$(find gpurun_out/${T}_prof_$wl -name "*_results.db" | head -1)
  if [ -n "$DB" ]; then python scripts/rocpd_stats.py "$DB" gpurun_out/${T}_${wl}_kernel_stats_single_stream.csv; head -16 gpurun_out/${T}_${wl}_kernel_stats_single_stream.csv | cut -c1-170; fi
  rm -rf gpurun_out/${T}_prof_$wl
done
