# Round 6 measurement bundle (one call): parity tests, smoke, the driver's bench command, rocprofv3 kernel statistics (product schedule and
# single stream; SCR, ASER, ER), the PMC passes (MFMA-pipe busy cycles; FETCH_SIZE / WRITE_SIZE, one pass each, counters alone with --kernel-trace).
#   gpurun --timeout 3000 -- 'bash scripts/gpu_r6_final.sh r6f'        then copy the summaries into profiles/ (scripts/README.md)
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
export TMPDIR=/tmp
T=${1:-r6f}
L=gpurun_out/${T}_info.log; : > $L
timeout -k 10 1500 python -m pytest tests/ -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/${T}_tests.log 2>&1; echo "tests rc=$?" >> $L
timeout -k 10 300 python __graft_entry__.py smoke > gpurun_out/${T}_smoke.log 2>&1; echo "smoke rc=$?" >> $L
timeout -k 10 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${T}_bench_scr.log 2>gpurun_out/${T}_bench_scr.err; echo "bench (driver's command) rc=$?" >> $L
timeout -k 10 1500 python bench.py > gpurun_out/${T}_bench_default.log 2>gpurun_out/${T}_bench_default.err; echo "bench (defaults) rc=$?" >> $L
Q="--no-cpu-baseline --no-also --no-accuracy --no-roofline --preroll 0 --repeats 1"
prof() {  # tag, bench args...
  tag=$1; shift
  rm -rf gpurun_out/${T}_p_$tag
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/${T}_p_$tag -o p -- python bench.py --steps 50 --warmup 10 $Q "$@" > gpurun_out/${T}_p_$tag.log 2>&1; echo "prof $tag rc=$?" >> $L
  DB=$(find gpurun_out/${T}_p_$tag -name "*_results.db" | head -1)
  if [ -n "$DB" ]; then python scripts/rocpd_stats.py "$DB" gpurun_out/${T}_${tag}_kernel_stats.csv; fi
  rm -rf gpurun_out/${T}_p_$tag
}
prof scr
prof scr_single_stream --single-stream
prof aser_single_stream --workload aser --single-stream
prof er_single_stream --workload er --single-stream
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d gpurun_out/${T}_pmc -o p -- python bench.py --steps 10 --warmup 3 $Q --single-stream > gpurun_out/${T}_pmc.log 2>&1; echo "pmc mfma rc=$?" >> $L
f=$(find gpurun_out/${T}_pmc -name "*counter_collection.csv" | head -1)
if [ -n "$f" ]; then python scripts/pmc_mfma.py $(dirname $f) gpurun_out/${T}_pmc_mfma.txt; fi
rm -rf gpurun_out/${T}_pmc
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/${T}_pmc_$c -o p -- python bench.py --steps 10 --warmup 3 $Q --single-stream > gpurun_out/${T}_pmc_$c.log 2>&1; echo "pmc $c rc=$?" >> $L
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
cat $L; grep -E "^FAILED|passed|failed|Error" gpurun_out/${T}_tests.log | tail -5; tail -1 gpurun_out/${T}_smoke.log; tail -1 gpurun_out/${T}_bench_scr.log | cut -c1-400
