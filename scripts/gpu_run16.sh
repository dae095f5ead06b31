mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
export TMPDIR=/tmp
for w in aser er; do timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r16_prof_$w -o $w -- python bench.py --workload $w --steps 30 --warmup 10 --no-cpu-baseline --no-roofline > gpurun_out/r16_prof_$w.log 2>&1; echo "prof $w rc=$?"; done
timeout 300 python scripts/host_profile.py aser > gpurun_out/r16_hostprof_aser.log 2>&1; echo "hostprof rc=$?"
timeout 300 python scripts/host_profile.py scr > gpurun_out/r16_hostprof_scr.log 2>&1; echo "hostprof rc=$?"
tail -40 gpurun_out/r16_hostprof_aser.log
