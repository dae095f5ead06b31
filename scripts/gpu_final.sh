# Round-end measurement bundle: parity tests, smoke, default bench (with cpu_baseline), rocprofv3 kernel stats, PMC traffic passes.
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
export TMPDIR=/tmp
T=${1:-final}
L=gpurun_out/${T}_info.log; : > $L
timeout 900 python -m pytest tests/ -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/${T}_tests.log 2>&1; echo "tests rc=$?" >> $L
timeout 300 python __graft_entry__.py smoke > gpurun_out/${T}_smoke.log 2>&1; echo "smoke rc=$?" >> $L
timeout 900 python bench.py > gpurun_out/${T}_bench_scr.log 2>&1; echo "bench rc=$?" >> $L
Q="--no-cpu-baseline --no-also --no-accuracy"
for w in aser er mir; do timeout 600 python bench.py --workload $w --steps 100 --warmup 10 $Q > gpurun_out/${T}_bench_$w.log 2>&1; echo "bench $w rc=$?" >> $L; done
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/${T}_prof -o scr -- python bench.py --steps 50 --warmup 10 $Q --no-roofline > gpurun_out/${T}_prof.log 2>&1; echo "prof rc=$?" >> $L
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/${T}_prof1 -o scr -- python bench.py --steps 50 --warmup 10 $Q --no-roofline --single-stream > gpurun_out/${T}_prof1.log 2>&1; echo "prof single-stream rc=$?" >> $L
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/${T}_prof2 -o aser -- python bench.py --workload aser --steps 50 --warmup 10 $Q --no-roofline --single-stream > gpurun_out/${T}_prof2.log 2>&1; echo "prof aser single-stream rc=$?" >> $L
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/${T}_pmc_$c -o p -- python bench.py --steps 10 --warmup 3 $Q --no-roofline --single-stream > gpurun_out/${T}_pmc_$c.log 2>&1; echo "pmc $c rc=$?" >> $L
done
python - "$T" <<'PY'
import csv, collections, json, glob, sys
T = sys.argv[1]
out = {}
for tag in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("gpurun_out/%s_pmc_%s/*counter_collection.csv" % (T, tag))
    if not f: continue
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].split("(")[0]
        agg[(k, r["Counter_Name"])][0] += float(r["Counter_Value"]); agg[(k, r["Counter_Name"])][1] += 1
    out[tag] = {"%s|%s" % k: dict(sum=v[0], n=v[1]) for k, v in agg.items()}
json.dump(out, open("gpurun_out/%s_pmc_summary.json" % T, "w"), indent=1)
PY
rm -rf gpurun_out/${T}_pmc_FETCH_SIZE gpurun_out/${T}_pmc_WRITE_SIZE
cat $L; grep -E "^FAILED|passed|failed|Error" gpurun_out/${T}_tests.log | tail -5; tail -1 gpurun_out/${T}_smoke.log; for f in gpurun_out/${T}_bench_*.log; do tail -1 $f | cut -c1-330; done
