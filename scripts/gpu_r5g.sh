# Round 5, call 8: the round-end bundle on the final tree -- few-slab reduction (S <= 4) bit for bit against the library before, the full
# GPU suite, smoke, the driver's bench command (all configs + accuracy + cpu baseline), then the profiles of this tree.
# gpurun --timeout 2700 -- 'bash scripts/gpu_r5g.sh r5g'
T=${1:-r5g}
bash scripts/gpu_r5f.sh ${T}f > /dev/null 2>&1   # (needs csrc/base/libocl_hip.so of the commit before the few-slab reduction: skipped when absent)
bash scripts/gpu_final.sh ${T}
bash scripts/gpu_prof.sh ${T}
