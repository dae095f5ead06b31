# rocprofv3 kernel stats (product schedule, single stream; SCR and ASER) + the two PMC traffic passes.  gpurun --timeout 900 -- 'bash scripts/gpu_prof.sh r3p'
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
export TMPDIR=/tmp
T=${1:-prof}
Q="--no-cpu-baseline --no-also --no-accuracy --no-roofline --preroll 0 --repeats 1"   # (60 steps per run: 10 warm-up + 50 timed)
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/${T}_prof -o scr -- python bench.py --steps 50 --warmup 10 $Q > gpurun_out/${T}_prof.log 2>&1; echo "prof rc=$?"
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/${T}_prof1 -o scr -- python bench.py --steps 50 --warmup 10 $Q --single-stream > gpurun_out/${T}_prof1.log 2>&1; echo "prof single-stream rc=$?"
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/${T}_prof2 -o aser -- python bench.py --workload aser --steps 50 --warmup 10 $Q --single-stream > gpurun_out/${T}_prof2.log 2>&1; echo "prof aser rc=$?"
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/${T}_prof3 -o er -- python bench.py --workload er --steps 50 --warmup 10 $Q --single-stream > gpurun_out/${T}_prof3.log 2>&1; echo "prof er rc=$?"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/${T}_pmc_$c -o p -- python bench.py --steps 10 --warmup 3 $Q --single-stream > gpurun_out/${T}_pmc_$c.log 2>&1; echo "pmc $c rc=$?"
done
python - "$T" <<'PY'
import csv, collections, json, glob, sys
T = sys.argv[1]
out = {}
for tag in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("gpurun_out/%s_pmc_%s/**/*counter_collection.csv" % (T, tag), recursive=True)
    if not f: continue
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].split("(")[0]
        agg[(k, r["Counter_Name"])][0] += float(r["Counter_Value"]); agg[(k, r["Counter_Name"])][1] += 1
    out[tag] = {"%s|%s" % k: dict(sum=v[0], n=v[1]) for k, v in agg.items()}
json.dump(out, open("gpurun_out/%s_pmc_summary.json" % T, "w"), indent=1)
print({k: len(v) for k, v in out.items()})
PY
rm -rf gpurun_out/${T}_pmc_FETCH_SIZE gpurun_out/${T}_pmc_WRITE_SIZE
