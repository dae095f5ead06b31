# wgrad kernels: this tree's kbench against kbench_base, same box, same run.  gpurun --timeout 600 -- 'bash scripts/gpu_wgrad_ab.sh r3d'
mkdir -p gpurun_out
T=${1:-wab}
K=online-continual-learning_amd/csrc/kbench
for n in 220 20; do
  timeout 200 ${K}_base $n 2 32 wgrad 0 > gpurun_out/${T}_wgrad_${n}_base.txt 2>&1; echo "base $n rc=$?"
  timeout 200 $K $n 2 32 wgrad 0 > gpurun_out/${T}_wgrad_${n}_new.txt 2>&1; echo "new $n rc=$?"
done
grep -c MISMATCH gpurun_out/${T}_wgrad_*_new.txt
python - "$T" <<'PY'
import re, sys
T = sys.argv[1]
def rows(f):
    out = []
    for l in open(f):
        m = re.match(r"^(\S+)\s+wgrad\s+M=.*?MTW=(\d) NTW=(\d).*?S=\s*(\d+).*?\s([\d.]+) us \+ reduce\s+([\d.]+) us", l)
        if m: out.append((m.group(1), float(m.group(5)), float(m.group(6))))
    return out
for n in (220, 20):
    b, w = rows("gpurun_out/%s_wgrad_%d_base.txt" % (T, n)), rows("gpurun_out/%s_wgrad_%d_new.txt" % (T, n))
    tb = tw = 0.0
    print("== N=%d: layer, base us (+reduce), new us (+reduce)" % n)
    for (nb, vb, rb), (nw, vw, rw) in zip(b, w):
        print("%-22s %7.1f +%4.1f  %7.1f +%4.1f  %+5.1f%%" % (nb, vb, rb, vw, rw, (vw / vb - 1) * 100))
        tb += vb; tw += vw
    print("sum %.1f -> %.1f" % (tb, tw))
PY
