# Per-layer A/B of the convolution kernels: csrc/kbench (this tree) against csrc/kbench_base (a binary kept from an earlier commit),
# same box, same run.   gpurun --timeout 900 -- 'bash scripts/gpu_conv_ab.sh r3b'
mkdir -p gpurun_out
export TMPDIR=/tmp
T=${1:-ab}
K=online-continual-learning_amd/csrc/kbench
for n in 220 20; do
  timeout 150 ${K}_base $n 2 32 conv 0 > gpurun_out/${T}_conv_${n}_base.txt 2>&1; echo "base $n rc=$?"
  timeout 150 $K $n 2 32 conv 0 > gpurun_out/${T}_conv_${n}_new.txt 2>&1; echo "new $n rc=$?"
done
timeout 150 $K 410 1 32 conv 0 > gpurun_out/${T}_conv_410_new.txt 2>&1; echo "new 410 rc=$?"
timeout 150 $K 15 1 84 conv 0 > gpurun_out/${T}_conv_84_new.txt 2>&1; echo "new 15x84 rc=$?"
KBENCH_TRACE=1 timeout 150 $K 220 2 32 conv 0 > gpurun_out/${T}_trace_220_new.txt 2>&1; echo "trace rc=$?"
timeout 100 $K 220 2 32 peak4 > gpurun_out/${T}_peak4.txt 2>&1; echo "peak4 rc=$?"; cat gpurun_out/${T}_peak4.txt
grep -c MISMATCH gpurun_out/${T}_conv_*_new.txt
python - "$T" <<'PY'
import re, sys
T = sys.argv[1]
def rows(f):
    out, name = [], None
    for l in open(f):
        m = re.match(r"^(\S+)\s+(fwd|dgrad\S*)\s+M=", l)
        if m: name = m.group(1) + " " + m.group(2); continue
        m = re.search(r"conv_([tqs]) \((auto|no-q)\)( ring)?\s+MT=(\d) NT=(\d).*?res=(\d)\s+([\d.]+) us", l)
        if m and name and m.group(2) == "auto": out.append((name + (" ring" if m.group(3) else ""), float(m.group(7))))
        m = re.search(r"^(\S+)\s+dgradM\*.*merged( \(ring\))? MT.*?res=\d\s+([\d.]+) us", l)
        if m: out.append((m.group(1) + " dgradM" + (" ring" if m.group(2) else ""), float(m.group(3))))
    return out
for n in (220, 20):
    b, w = rows("gpurun_out/%s_conv_%d_base.txt" % (T, n)), rows("gpurun_out/%s_conv_%d_new.txt" % (T, n))
    tb = tw = 0.0
    print("== N=%d: layer, base us, new us" % n)
    for (nb, vb), (nw, vw) in zip(b, w):
        assert nb == nw, (nb, nw)
        print("%-34s %7.1f %7.1f  %+5.1f%%" % (nb, vb, vw, (vw / vb - 1) * 100))
        tb += vb; tw += vw
    print("sum %.1f -> %.1f" % (tb, tw))
PY
