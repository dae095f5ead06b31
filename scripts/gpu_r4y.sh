# round 4: the bench line (driver's command) + profile bundle once more, on another lease.   gpurun --timeout 1500 -- 'bash scripts/gpu_r4y.sh r4y'
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
export TMPDIR=/tmp
T=${1:-r4y}
rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|mclk|fclk|Power" | head -6 > gpurun_out/${T}_smi.txt
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${T}_bench_scr.log 2>gpurun_out/${T}_bench_scr.err; echo "bench rc=$?"
tail -1 gpurun_out/${T}_bench_scr.log | cut -c1-700
bash scripts/gpu_prof.sh $T
