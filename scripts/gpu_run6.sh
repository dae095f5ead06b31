mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
L=gpurun_out/r6_info.log
rocminfo 2>/dev/null | grep -m2 -E "gfx|Marketing" > $L; nproc >> $L; lscpu | grep -m1 "Model name" >> $L
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/r6_kernels.log 2>&1; echo "kernels rc=$?" >> $L
timeout 600 python -m pytest tests/test_gpu_net.py -m gpu -q --tb=short -p no:cacheprovider -s > gpurun_out/r6_net.log 2>&1; echo "net rc=$?" >> $L
timeout 900 python -m pytest tests/test_gpu_steps.py -m gpu -q --tb=short -p no:cacheprovider -s > gpurun_out/r6_steps.log 2>&1; echo "steps rc=$?" >> $L
timeout 300 python __graft_entry__.py smoke > gpurun_out/r6_smoke.log 2>&1; echo "smoke rc=$?" >> $L
timeout 600 python bench.py --steps 100 --warmup 10 --cpu-steps 20 > gpurun_out/r6_bench.log 2>&1; echo "bench rc=$?" >> $L
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r6_prof -o scr -- python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-roofline > gpurun_out/r6_prof.log 2>&1; echo "prof rc=$?" >> $L
find gpurun_out/r6_prof -name "*trace*" -size +2M -delete
cat $L; tail -5 gpurun_out/r6_kernels.log; grep -E "^FAILED|passed|failed" gpurun_out/r6_net.log | tail -8; grep -E "^FAILED|passed|failed" gpurun_out/r6_steps.log | tail -20; tail -2 gpurun_out/r6_smoke.log; tail -3 gpurun_out/r6_bench.log
