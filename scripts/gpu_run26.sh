mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
L=gpurun_out/r26_info.log; : > $L
timeout 900 python -m pytest tests/test_gpu_net.py tests/test_gpu_steps.py -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/r26_tests.log 2>&1; echo "tests rc=$?" >> $L
for w in scr aser er mir; do timeout 600 python bench.py --workload $w --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/r26_bench_$w.log 2>&1; echo "bench $w rc=$?" >> $L; done
cat $L; grep -E "^FAILED|passed|failed|Error" gpurun_out/r26_tests.log | tail -5; for f in gpurun_out/r26_bench_*.log; do tail -1 $f | cut -c1-330; done
