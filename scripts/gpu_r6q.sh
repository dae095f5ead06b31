# Round 6, call 23: the ASER stall probe with the garbage collector's events logged; the oracle parity tests of the network with conv_w_kernel wherever it fits
T=${1:-r6q}
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
OCL_LOG_PLANS=1 timeout 600 python scripts/aser_stall_probe.py > gpurun_out/${T}_aser_probe.txt 2> gpurun_out/${T}_aser_probe.err; echo "probe rc=$?"
cat gpurun_out/${T}_aser_probe.txt
grep -n "gc generation" gpurun_out/${T}_aser_probe.err | tail -20
OCL_CONV_W=2 timeout 1200 python -m pytest tests/test_gpu_net.py tests/test_gpu_kernels.py -x -q > gpurun_out/${T}_parity_w2.txt 2>&1; echo "parity (OCL_CONV_W=2) rc=$?"; tail -4 gpurun_out/${T}_parity_w2.txt
