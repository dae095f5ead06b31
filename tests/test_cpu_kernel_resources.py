"""CPU: the hot kernels must not use scratch (a private segment is paid at wave launch: profiles/r3_conv_s_ab.md -- a conv_s_kernel build
with 10 spilled VGPRs was 1 - 4 us per launch slower than the build before it, with a faster loop).  Compiles conv.hip, wgrad.hip and convw.hip for
gfx950 with the compiler's resource remarks (no GPU needed; the two files side by side) and reads ScratchSize per kernel."""
import re
import shutil
import subprocess

import pytest

from conftest import ROOT

HIPCC = "/opt/rocm/bin/hipcc"
SRCS = [ROOT + "/online-continual-learning_amd/csrc/" + f for f in ("conv.hip", "wgrad.hip", "convw.hip")]

# the kernels a training / eval step launches (templates: the instantiations the planner picks at the BASELINE sizes)
HOT = [
    r"conv_s_kernel<1, false, (true|false), false>", r"conv_s_kernel<2, false, (true|false), false>",   # (the deterministic-mode instantiations <..., true> spill 4 VGPRs: that mode costs 12 % anyway)
    r"conv_q_kernel<2, 4, [012], 0>", r"conv_q_kernel<2, 12, [012], 0>", r"conv_q_kernel<1, 12, [012], 0>",
    r"conv_t_kernel<3, 1, 8, true, false, false, (true|false)>", r"conv_t_kernel<5, 1, 8, false, false, true, (true|false)>",
    r"conv_t_kernel<3, 1, 8, false, false, true, (true|false)>",
    r"conv_t_kernel<1, 1, 8, (true|false), (true|false), (true|false), false>", r"conv_t_kernel<2, 1, 4, true, (true|false), false, false>",
    r"conv_t_kernel<1, 1, 8, true, false, false, true>", r"conv_t_kernel<2, 1, 4, true, false, false, true>",   # the EPI_BNB data gradients of a replay-sized pass
    r"conv_wgrad_kernel<2, 3, 8, 0, 0>", r"conv_wgrad_kernel<3, 2, 8, 0, 0>", r"conv_wgrad_kernel<1, 3, 8, 0, 0>",
    r"conv_wgrad_kernel<1, 2, 8, 0, 0>", r"conv_wgrad_kernel<1, 2, 4, 0, 0>",
    r"conv_wgrad_kernel<1, 5, [48], 0, 0>",       # (80-channel blocks: layers 3 - 4 of passes of 48 images and more)
    r"conv_wgrad_kernel<1, 1, [48], [123], 0>",   # (the 4x4x1 form: layer 1 of the large passes)
    r"conv_wgrad_multi_kernel<[01]>",             # (every layer of a replay-sized pass in one launch)
    # the one-wave-per-SIMD forms the planner takes by default (layers 2 - 3, 3x3 stride 1, forward without input transform / data gradient)
    r"conv_wx_kernel<3, 1, 23, 10, (true|false), false, false>", r"conv_wx_kernel<1, 1, 45, 14, (true|false), false, false>",
    r"conv_wx_kernel<1, 1, 23, 7, (true|false), false, false>",
    r"bn_fwd_kernel", r"bn_bwd_fused_kernel<\d+, \d>", r"bn_bwd_chan_kernel<1, [12]>", r"bn_bwd_apply_e_kernel", r"wgrad_reduce_kernel", r"wgrad_reduce_multi_kernel",
]


@pytest.mark.skipif(not shutil.which(HIPCC), reason="hipcc not installed")
def test_hot_kernels_use_no_scratch():
    procs = [subprocess.Popen([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", "/dev/null",
                               "-Rpass-analysis=kernel-resource-usage"], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True) for src in SRCS]
    remarks = ""
    for p in procs:
        _, err = p.communicate(timeout=900)
        assert p.returncode == 0, err[-2000:]
        remarks += err
    scratch, cur = {}, None
    for line in remarks.splitlines():
        m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            name = t.split(":", 1)[1].strip()
            cur = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            cur = re.sub(r"\(.*", "", cur).replace("void ", "").replace("ocl::", "")
        elif cur and t.startswith("ScratchSize"):
            scratch[cur] = int(re.search(r":\s*(\d+)", t).group(1))
    assert len(scratch) > 100, "no resource remarks parsed"
    for pat in HOT:
        hits = {k: v for k, v in scratch.items() if re.fullmatch(pat, k)}
        assert hits, "no kernel matches %s" % pat
        bad = {k: v for k, v in hits.items() if v}
        assert not bad, "scratch in a hot kernel: %s" % bad
