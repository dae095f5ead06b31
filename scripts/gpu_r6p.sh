# Round 6, call 22: (a) the GPU A/B test of conv_w_kernel against conv_t_kernel on the whole network; (b) the ASER stall probe (engine log of plan sets / arena chunks
# per step) (c) the bench's ASER leg as the driver runs it
T=${1:-r6p}
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_gpu_ring.py -x -q -k "conv_w" -s > gpurun_out/${T}_convw_ab.txt 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/${T}_convw_ab.txt
OCL_LOG_PLANS=1 timeout 600 python scripts/aser_stall_probe.py > gpurun_out/${T}_aser_probe.txt 2> gpurun_out/${T}_aser_probe.err; echo "probe rc=$?"
cat gpurun_out/${T}_aser_probe.txt
grep -n "\[ocl\]" gpurun_out/${T}_aser_probe.err | tail -40
