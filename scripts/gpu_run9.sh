mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
export TMPDIR=/tmp
K=online-continual-learning_amd/csrc/kbench
timeout 120 $K 220 2 32 peak 0 > gpurun_out/r9_peak.log 2>&1; echo "peak rc=$?"
rocprofv3 -L > gpurun_out/r9_counters.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/r9_pmc1 -o k -- $K 220 2 32 conv 0 > gpurun_out/r9_pmc1.log 2>&1; echo "pmc1 rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d gpurun_out/r9_pmc2 -o k -- $K 220 2 32 conv 0 > gpurun_out/r9_pmc2.log 2>&1; echo "pmc2 rc=$?"
cat gpurun_out/r9_peak.log
ls gpurun_out/r9_pmc1 gpurun_out/r9_pmc2
