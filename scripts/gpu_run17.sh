mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
export TMPDIR=/tmp
L=gpurun_out/r17_info.log; : > $L
timeout 900 python -m pytest tests/test_gpu_steps.py tests/test_gpu_net.py -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/r17_tests.log 2>&1; echo "tests rc=$?" >> $L
for w in scr er aser mir; do timeout 600 python bench.py --workload $w --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/r17_bench_$w.log 2>&1; echo "bench $w rc=$?" >> $L; done
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/r17_pmc_fetch -o p -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/r17_pmc_fetch.log 2>&1; echo "pmc fetch rc=$?" >> $L
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/r17_pmc_write -o p -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/r17_pmc_write.log 2>&1; echo "pmc write rc=$?" >> $L
# keep the merge small: aggregate per kernel name on the box
python - <<'PY'
import csv, collections, json, glob
out = {}
for tag in ("fetch", "write"):
    f = glob.glob("gpurun_out/r17_pmc_%s/*counter_collection.csv" % tag)
    if not f: continue
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].split("(")[0]
        agg[(k, r["Counter_Name"])][0] += float(r["Counter_Value"]); agg[(k, r["Counter_Name"])][1] += 1
    out[tag] = {"%s|%s" % k: dict(sum=v[0], n=v[1]) for k, v in agg.items()}
json.dump(out, open("gpurun_out/r17_pmc_summary.json", "w"), indent=1)
PY
rm -rf gpurun_out/r17_pmc_fetch gpurun_out/r17_pmc_write
cat $L; grep -E "^FAILED|passed|failed|Error" gpurun_out/r17_tests.log | tail -5; for f in gpurun_out/r17_bench_*.log; do tail -1 $f | cut -c1-330; done
