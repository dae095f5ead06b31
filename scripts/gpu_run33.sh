mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
L=gpurun_out/r33_info.log; : > $L
timeout 900 python -m pytest tests/ -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/r33_tests.log 2>&1; echo "tests rc=$?" >> $L
timeout 300 python __graft_entry__.py smoke > gpurun_out/r33_smoke.log 2>&1; echo "smoke rc=$?" >> $L
timeout 900 python bench.py > gpurun_out/r33_bench_default.log 2>&1; echo "bench default rc=$?" >> $L
timeout 600 python bench.py --workload aser --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/r33_bench_aser.log 2>&1; echo "bench aser rc=$?" >> $L
cat $L; grep -E "^FAILED|passed|failed|Error" gpurun_out/r33_tests.log | tail -5; tail -1 gpurun_out/r33_smoke.log; tail -1 gpurun_out/r33_bench_default.log | cut -c1-2500; tail -1 gpurun_out/r33_bench_aser.log | cut -c1-300
