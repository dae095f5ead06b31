mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
T=g1
timeout 600 python -m pytest tests/test_gpu_net.py -m gpu -q --tb=short -p no:cacheprovider -x -k "graph_replay" > gpurun_out/${T}_tests_graph.log 2>&1; echo "graph tests rc=$?"
tail -15 gpurun_out/${T}_tests_graph.log
for w in scr er; do timeout 120 python scripts/host_enqueue.py $w 200 2>&1 | tail -2; done
for w in scr er aser mir; do timeout 300 python bench.py --workload $w --steps 100 --warmup 20 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c1-260; done
OCL_NO_GRAPH=1 timeout 300 python bench.py --workload scr --steps 100 --warmup 20 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c1-260
