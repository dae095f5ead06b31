"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel / grid: MFMA-pipe busy cycles against GPU-active cycles.
usage: python scripts/pmc_mfma.py <dir with *counter_collection.csv> <out.txt>"""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/*counter_collection.csv")
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].split("(")[0][-40:] + " grid=" + r.get("Grid_Size", "?") + " lds=" + r.get("LDS_Block_Size", "?")
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[(k, r["Counter_Name"])] += 1
with open(sys.argv[2], "w") as out:
    for k, v in sorted(agg.items()):
        n = max(cnt[(k, c)] for c in v)
        line = "%-90s n=%3d " % (k, n) + " ".join("%s=%.3g" % (c, x / cnt[(k, c)]) for c, x in sorted(v.items()))
        if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "GRBM_GUI_ACTIVE" in v and v["GRBM_GUI_ACTIVE"] > 0:
            # SQ_* counters are summed over the SIMDs of the chip (4 x 256), GRBM_GUI_ACTIVE counts once
            line += "  mfma_busy/(gui_active*1024)=%.3f" % (v["SQ_VALU_MFMA_BUSY_CYCLES"] / (v["GRBM_GUI_ACTIVE"] * 1024.0))
        out.write(line + "\n")
