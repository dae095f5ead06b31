# single-stream SCR kernel stats under different environment settings:  bash scripts/gpu_prof_ab.sh TAG "ENV1=.. ENV2=.." "ENV.."   (one run per argument; "-" = no settings)
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
T=$1; shift
Q="--no-cpu-baseline --no-also --no-accuracy --no-roofline --single-stream"
i=0
for e in "$@"; do
  [ "$e" = "-" ] && e=""
  env $e timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/${T}_p$i -o scr -- python bench.py --steps 30 --warmup 10 $Q > gpurun_out/${T}_p$i.log 2>&1
  python scripts/rocpd_stats.py $(find gpurun_out/${T}_p$i -name "scr_results.db") gpurun_out/${T}_stats_$i.csv
  rm -rf gpurun_out/${T}_p$i
  echo "== run $i: $e"; grep -E "conv_[stq]_kernel" gpurun_out/${T}_stats_$i.csv | cut -c1-130
  i=$((i+1))
done
