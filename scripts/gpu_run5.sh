mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/r5_kernels.log 2>&1; echo "kernels rc=$?"
timeout 900 python -m pytest tests/test_gpu_net.py -m gpu -q --tb=short -p no:cacheprovider -s > gpurun_out/r5_net.log 2>&1; echo "net rc=$?"
timeout 1200 python -m pytest tests/test_gpu_steps.py -m gpu -q --tb=short -p no:cacheprovider -s > gpurun_out/r5_steps.log 2>&1; echo "steps rc=$?"
timeout 300 python __graft_entry__.py smoke > gpurun_out/r5_smoke.log 2>&1; echo "smoke rc=$?"
tail -4 gpurun_out/r5_kernels.log; grep -E "mismatches|^FAILED|passed|failed" gpurun_out/r5_net.log | tail -15; grep -E "update err|^FAILED|passed|failed|ASER steps|divergence" gpurun_out/r5_steps.log | tail -40; tail -2 gpurun_out/r5_smoke.log
