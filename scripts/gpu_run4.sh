mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
cd scripts && timeout 600 python diag_bn_inputs.py > ../gpurun_out/r4_diag.log 2>&1; echo "diag rc=$?"; cd ..
tail -60 gpurun_out/r4_diag.log
