mkdir -p gpurun_out
export TMPDIR=/tmp
K=online-continual-learning_amd/csrc/kbench
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/r32_pmc1 -o k -- $K 220 2 32 wgrad 0 > gpurun_out/r32_pmc1.log 2>&1; echo "pmc1 rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d gpurun_out/r32_pmc2 -o k -- $K 220 2 32 wgrad 0 > gpurun_out/r32_pmc2.log 2>&1; echo "pmc2 rc=$?"
