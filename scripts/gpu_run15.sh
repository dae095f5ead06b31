mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
export TMPDIR=/tmp
L=gpurun_out/r15_info.log; : > $L
timeout 900 python -m pytest tests/test_gpu_steps.py -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/r15_tests.log 2>&1; echo "tests rc=$?" >> $L
timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r15_bench.log 2>&1; echo "bench rc=$?" >> $L
for w in er aser mir; do timeout 600 python bench.py --workload $w --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/r15_bench_$w.log 2>&1; echo "bench $w rc=$?" >> $L; done
cat $L; grep -E "^FAILED|passed|failed|Error" gpurun_out/r15_tests.log | tail -8; for f in gpurun_out/r15_bench*.log; do tail -1 $f | cut -c1-420; done
