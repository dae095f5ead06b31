# conv_s_kernel variants (channel tiles per workgroup x weight rounds in flight) against conv_t_kernel, per layer-3/4 launch.
#   gpurun -- 'bash scripts/gpu_convs_ab.sh r3o "20 100 220"'   then   python scripts/gpu_convs_ab.sh.py (parse) -- see profiles/r3_conv_s_ab.md
K=online-continual-learning_amd/csrc/kbench
T=${1:-convs}
for n in ${2:-20 100 220}; do
  for nt in ${3:-1 2}; do for d in ${4:-4 8}; do
    OCL_CONV_S_UNITS=100000 OCL_CONV_S_NT=$nt OCL_CONV_S_DEPTH=$d timeout 100 $K $n 2 32 conv 0 > gpurun_out/${T}_n${n}_nt${nt}_d${d}.txt 2>&1
  done; done
done
