# Round 6: (1) the ASER leg after gc.freeze (stall probe + the bench's own aser workload), (2) the torch-eager same-node comparator,
# (3) the RCCL world-of-one test and the conv_w tests.
# gpurun --timeout 1500 -- 'bash scripts/gpu_r6r.sh r6r'
T=${1:-r6r}
mkdir -p gpurun_out
OCL_LOG_PLANS=1 timeout 300 python scripts/aser_stall_probe.py > gpurun_out/${T}_aser_probe.txt 2> gpurun_out/${T}_aser_probe.err; echo "probe rc=$?"
grep -E "repeat|gc generation" gpurun_out/${T}_aser_probe.txt gpurun_out/${T}_aser_probe.err | tail -20
timeout 400 python bench.py --workload aser --steps 100 --warmup 5 --no-roofline --no-accuracy --no-cpu-baseline > gpurun_out/${T}_bench_aser.json 2> gpurun_out/${T}_bench_aser.err; echo "bench aser rc=$?"
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r6r_bench_aser.json") if l.startswith("{")][-1])
print("aser", d["ms_per_step"], d.get("ms_per_step_max"), d["ms_per_step_repeats"])
PY
timeout 600 python scripts/torch_eager_on_mi355x.py > gpurun_out/${T}_torch_eager.txt 2> gpurun_out/${T}_torch_eager.err; echo "eager rc=$?"
cat gpurun_out/${T}_torch_eager.txt
timeout 900 python -m pytest tests/test_gpu_parity2.py -x -q -m gpu -k "rccl or sharded or bench_gpus" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_ring.py -x -q -m gpu -k "conv_w" 2>&1 | tail -5
