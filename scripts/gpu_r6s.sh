# Round 6: conv_w planner gate A/B on the three bench legs (OCL_CONV_W = 0 never / 1 hot shapes only / 2 wherever a plan fits), + the eager comparator.
# gpurun --timeout 1500 -- 'bash scripts/gpu_r6s.sh r6s'
T=${1:-r6s}
mkdir -p gpurun_out
for wl in aser er scr; do
  for cw in 0 1 2; do
    OCL_CONV_W=$cw timeout 300 python bench.py --workload $wl --steps 100 --warmup 5 --no-roofline --no-accuracy --no-cpu-baseline --no-also > gpurun_out/${T}_${wl}_cw${cw}.json 2> gpurun_out/${T}_${wl}_cw${cw}.err
    python - $wl $cw gpurun_out/${T}_${wl}_cw${cw}.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[3]) if l.startswith("{")][-1])
print(sys.argv[1], "OCL_CONV_W=%s" % sys.argv[2], "ms_per_step %.4f max %.4f" % (d["ms_per_step"], d.get("ms_per_step_max",0)), d["ms_per_step_repeats"])
PY
  done
done 2>&1 | tee gpurun_out/${T}_convw_gate_ab.txt
timeout 600 python scripts/torch_eager_on_mi355x.py --workloads scr,er > gpurun_out/${T}_torch_eager.txt 2> gpurun_out/${T}_torch_eager.err; echo "eager rc=$?"
cat gpurun_out/${T}_torch_eager.txt
