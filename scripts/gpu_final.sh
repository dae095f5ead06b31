# Round-end measurement bundle: parity tests, smoke, the driver's bench command (SCR + also.aser / er / mir, accuracy, cpu_baseline).
#   gpurun --timeout 2400 -- 'bash scripts/gpu_final.sh r3f; bash scripts/gpu_prof.sh r3f'      then: python scripts/collect_profiles.py r3f v2 r3
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
export TMPDIR=/tmp
T=${1:-final}
L=gpurun_out/${T}_info.log; : > $L
timeout 1200 python -m pytest tests/ -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/${T}_tests.log 2>&1; echo "tests rc=$?" >> $L
timeout 300 python __graft_entry__.py smoke > gpurun_out/${T}_smoke.log 2>&1; echo "smoke rc=$?" >> $L
# the driver's exact command (its record is BENCH_rNN.json): SCR + also.{aser,er,mir} + accuracy + cpu_baseline in one line
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${T}_bench_scr.log 2>gpurun_out/${T}_bench_scr.err; echo "bench rc=$?" >> $L
cat $L; grep -E "^FAILED|passed|failed|Error" gpurun_out/${T}_tests.log | tail -5; tail -1 gpurun_out/${T}_smoke.log; tail -1 gpurun_out/${T}_bench_scr.log | cut -c1-600
