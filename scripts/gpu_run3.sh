mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 600 python scripts/diag_bn.py > gpurun_out/r3_diag.log 2>&1; echo "diag rc=$?"
tail -60 gpurun_out/r3_diag.log
