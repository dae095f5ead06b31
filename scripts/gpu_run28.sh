mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
export TMPDIR=/tmp
L=gpurun_out/r28_info.log; : > $L
timeout 900 python -m pytest tests/ -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/r28_tests.log 2>&1; echo "tests rc=$?" >> $L
timeout 300 python __graft_entry__.py smoke > gpurun_out/r28_smoke.log 2>&1; echo "smoke rc=$?" >> $L
for w in scr aser er mir; do timeout 600 python bench.py --workload $w --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/r28_bench_$w.log 2>&1; echo "bench $w rc=$?" >> $L; done
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r28_prof -o scr -- python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-roofline > gpurun_out/r28_prof.log 2>&1; echo "prof rc=$?" >> $L
cat $L; grep -E "^FAILED|passed|failed|Error" gpurun_out/r28_tests.log | tail -5; tail -1 gpurun_out/r28_smoke.log; for f in gpurun_out/r28_bench_*.log; do tail -1 $f | cut -c1-330; done
