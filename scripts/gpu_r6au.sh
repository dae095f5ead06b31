# Round 6: both bench commands on the final tree (accuracy.aser on the stream that fills its memory early, five seeds)
T=${1:-r6au}
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
t0=$(date +%s); timeout -k 10 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${T}_bench_scr.log 2> gpurun_out/${T}_bench_scr.err; echo "bench (driver's command) rc=$? wall $(( $(date +%s) - t0 )) s"
t0=$(date +%s); timeout -k 10 1500 python bench.py > gpurun_out/${T}_bench_default.log 2> gpurun_out/${T}_bench_default.err; echo "bench (defaults) rc=$? wall $(( $(date +%s) - t0 )) s"
