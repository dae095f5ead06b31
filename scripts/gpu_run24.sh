mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
export TMPDIR=/tmp
L=gpurun_out/r24_info.log; : > $L
timeout 900 python -m pytest tests/test_gpu_net.py tests/test_gpu_steps.py -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/r24_tests.log 2>&1; echo "tests rc=$?" >> $L
timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/r24_bench_scr.log 2>&1; echo "bench rc=$?" >> $L
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r24_prof -o scr -- python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-roofline > gpurun_out/r24_prof.log 2>&1; echo "prof rc=$?" >> $L
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r24_prof_aser -o aser -- python bench.py --workload aser --steps 30 --warmup 10 --no-cpu-baseline --no-roofline > gpurun_out/r24_prof_aser.log 2>&1; echo "prof aser rc=$?" >> $L
cat $L; grep -E "^FAILED|passed|failed|Error" gpurun_out/r24_tests.log | tail -5; tail -1 gpurun_out/r24_bench_scr.log | cut -c1-330
