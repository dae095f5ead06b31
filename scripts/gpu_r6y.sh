# Round 6: full GPU suite on the tree with the merged weight-gradient launch, per-layer dL/dy buffers, gc.freeze, conv_w.
T=${1:-r6y}
mkdir -p gpurun_out
timeout -k 10 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/${T}_pytest_gpu.txt
