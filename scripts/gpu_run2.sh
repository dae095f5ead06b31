mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 600 python scripts/diag_backward.py > gpurun_out/r2_diag.log 2>&1; echo "diag rc=$?"
tail -100 gpurun_out/r2_diag.log
