# Round 6 (experiment): does the ASER step drift with the step count?  15 back-to-back repeats of 100 steps (same batches every repeat).
T=${1:-r6an}
mkdir -p gpurun_out
timeout -k 10 600 python bench.py --workload aser --steps 100 --warmup 5 --repeats 15 --no-roofline --no-accuracy --no-cpu-baseline --no-also > gpurun_out/${T}_aser15.json 2> gpurun_out/${T}_aser15.err
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r6an_aser15.json") if l.startswith("{")][-1])
print("aser 15 repeats (sorted as reported):", d["ms_per_step_repeats"])
PY
OCL_LOG_PLANS=1 timeout -k 10 600 python scripts/aser_stall_probe.py --repeats 12 > gpurun_out/${T}_probe.txt 2> gpurun_out/${T}_probe.err
grep -E "^repeat" gpurun_out/${T}_probe.txt | cut -c1-140
