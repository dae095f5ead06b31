"""Per-kernel register / LDS / occupancy table of a HIP source (hipcc -Rpass-analysis=kernel-resource-usage), one line per kernel.
   python scripts/kernel_resources.py online-continual-learning_amd/csrc/conv.hip [filter]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
extra = sys.argv[3:] 
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", "/dev/null",
                    "-Rpass-analysis=kernel-resource-usage"] + extra, capture_output=True, text=True)
cur = None
rows = []
for line in r.stderr.splitlines():
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        name = t.split(":", 1)[1].strip()
        d = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        cur = {"name": re.sub(r"\(.*", "", d).replace("void ocl::", "")}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
for c in rows:
    if flt in c["name"]:
        print("%-60s sgpr %-4s vgpr %-4s agpr %-4s spill s/v %s/%s occ %-2s lds %s" % (
            c["name"], c.get("TotalSGPRs"), c.get("VGPRs"), c.get("AGPRs"), c.get("SGPRs Spill"), c.get("VGPRs Spill"),
            c.get("Occupancy [waves/SIMD]"), c.get("LDS Size [bytes/block]")))
