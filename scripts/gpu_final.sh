# Round-end measurement bundle: parity tests, smoke, default bench (with cpu_baseline, also.aser, accuracy), the other three configs.
#   gpurun --timeout 2400 -- 'bash scripts/gpu_final.sh r3f; bash scripts/gpu_prof.sh r3f'      then: python scripts/collect_profiles.py r3f v2 r3
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
export TMPDIR=/tmp
T=${1:-final}
L=gpurun_out/${T}_info.log; : > $L
timeout 1200 python -m pytest tests/ -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/${T}_tests.log 2>&1; echo "tests rc=$?" >> $L
timeout 300 python __graft_entry__.py smoke > gpurun_out/${T}_smoke.log 2>&1; echo "smoke rc=$?" >> $L
timeout 1200 python bench.py > gpurun_out/${T}_bench_scr.log 2>gpurun_out/${T}_bench_scr.err; echo "bench rc=$?" >> $L
Q="--no-cpu-baseline --no-also --no-accuracy"
for w in aser er mir; do timeout 600 python bench.py --workload $w --steps 100 --warmup 10 $Q > gpurun_out/${T}_bench_$w.log 2>&1; echo "bench $w rc=$?" >> $L; done
cat $L; grep -E "^FAILED|passed|failed|Error" gpurun_out/${T}_tests.log | tail -5; tail -1 gpurun_out/${T}_smoke.log; for f in gpurun_out/${T}_bench_*.log; do tail -1 $f | cut -c1-330; done
