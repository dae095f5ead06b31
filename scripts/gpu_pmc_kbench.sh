# PMC passes over `kbench N 2 32 conv 0` (every forward / data-gradient launch once per plan): one pass per counter group, summarised per kernel and grid.
#   gpurun -- 'bash scripts/gpu_pmc_kbench.sh r3t 220'
mkdir -p gpurun_out
export TMPDIR=/tmp
T=$1; N=${2:-220}
K=online-continual-learning_amd/csrc/kbench
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum" \
           "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU"; do
  timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/${T}_pmc$i -o p -- $K $N 2 32 conv 0 > gpurun_out/${T}_pmc$i.log 2>&1; echo "group $i rc=$?"
  f=$(find gpurun_out/${T}_pmc$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python scripts/pmc_mfma.py $(dirname $f) gpurun_out/${T}_pmc$i.txt
  rm -rf gpurun_out/${T}_pmc$i
  i=$((i+1))
done
