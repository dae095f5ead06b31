"""GPU: run-time forms of the weight gradient through the WHOLE training pass, via the C-ABI alone.

`csrc/netcheck` is a consumer of include/ocl_hip.h + libocl_hip.so and nothing else (no torch in the process): one training forward +
backward of the engine on seeded inputs; the flat gradient, the outputs and the BatchNorm running statistics go to a file or are compared
with one, tensor by tensor.  The planner reads its knobs once per process, so the two forms of an A/B are two processes.

  * the 4x4x1 form of layer 1 (the default where the planner's gate takes it; OCL_WGRAD_Q=0 = the 16x16x4 form everywhere) sums in another
    order: exactly the four 3x3 weights of layer 1 may differ, by rounding; on the passes the gate leaves alone nothing may differ.
"""
import os
import re
import subprocess

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

CSRC = os.path.join(ROOT, "online-continual-learning_amd", "csrc")
NETCHECK = os.path.join(CSRC, "netcheck")


def _run(cfg, mode, path, env):
    if not os.path.exists(NETCHECK):
        subprocess.run(["make", "-C", CSRC, "netcheck"], check=True, stdout=subprocess.DEVNULL)
    r = subprocess.run([NETCHECK] + [str(v) for v in cfg] + [mode, path], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, OCL_DETERMINISTIC="1", **env))
    return r.returncode, r.stdout + r.stderr


# (images, BatchNorm groups, input size, head): SCR's 220 views, a replay-sized ER pass, an odd batch, mini-ImageNet's 84 x 84
CASES = [(220, 2, 32, 1), (20, 1, 32, 0), (13, 1, 32, 0), (6, 2, 84, 1)]


@pytest.mark.parametrize("cfg", CASES[1:], ids=lambda c: "n%d_g%d_hw%d_head%d" % c)
def test_4x4x1_weight_gradient_form_is_not_planned_on_small_passes(cfg, tmp_path):
    ref = str(tmp_path / "ref.bin")
    rc, out = _run(cfg, "write", ref, {"OCL_WGRAD_Q": "0"})
    assert rc == 0, out
    rc, out = _run(cfg, "compare", ref, {})
    assert rc == 0, out
    m = re.search(r"(\d+) of (\d+) tensors differ in some bit", out)
    assert m and int(m.group(1)) == 0 and int(m.group(2)) >= 60, out


def test_4x4x1_weight_gradient_form_changes_only_layer_1_and_only_by_rounding(tmp_path):
    cfg = (220, 2, 32, 1)
    ref = str(tmp_path / "ref.bin")
    rc, out = _run(cfg, "write", ref, {"OCL_WGRAD_Q": "0"})
    assert rc == 0, out
    rc, out = _run(cfg, "compare", ref, {})
    assert rc == 0, out                                   # (exit 1: some tensor off by more than 1e-4 of its largest entry, or NaN)
    differing = re.findall(r"^\s+(\S+)\s+\d+ floats\s+reldiff (\S+)", out, re.M)
    assert sorted(n for n, _ in differing) == ["encoder.layer1.%d.conv%d.weight" % (b, c) for b in (0, 1) for c in (1, 2)], out
    assert all(float(v) < 5e-6 for _, v in differing), out
