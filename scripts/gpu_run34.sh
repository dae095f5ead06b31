mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
L=gpurun_out/r34_info.log; : > $L
timeout 900 python -m pytest tests/ -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/r34_tests.log 2>&1; echo "tests rc=$?" >> $L
for w in scr aser er mir; do timeout 600 python bench.py --workload $w --steps 100 --warmup 10 --no-cpu-baseline --no-roofline > gpurun_out/r34_bench_$w.log 2>&1; echo "bench $w rc=$?" >> $L; done
cat $L; grep -E "^FAILED|passed|failed|Error" gpurun_out/r34_tests.log | tail -5; for f in gpurun_out/r34_bench_*.log; do tail -1 $f | cut -c1-300; done
