mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
rocminfo 2>/dev/null | grep -m2 -E "gfx|Marketing" > gpurun_out/r1_info.log; nproc >> gpurun_out/r1_info.log; lscpu | grep -m1 "Model name" >> gpurun_out/r1_info.log
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/r1_kernels.log 2>&1; echo "kernels rc=$?" >> gpurun_out/r1_info.log
timeout 900 python -m pytest tests/test_gpu_net.py -m gpu -q --tb=short -p no:cacheprovider -s > gpurun_out/r1_net.log 2>&1; echo "net rc=$?" >> gpurun_out/r1_info.log
timeout 900 python -m pytest tests/test_gpu_steps.py -m gpu -q --tb=short -p no:cacheprovider -s > gpurun_out/r1_steps.log 2>&1; echo "steps rc=$?" >> gpurun_out/r1_info.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/r1_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/r1_info.log
timeout 600 python bench.py --steps 50 --warmup 10 --cpu-steps 20 > gpurun_out/r1_bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/r1_info.log
cat gpurun_out/r1_info.log; tail -5 gpurun_out/r1_kernels.log; tail -3 gpurun_out/r1_net.log; tail -3 gpurun_out/r1_steps.log; tail -3 gpurun_out/r1_bench.log
