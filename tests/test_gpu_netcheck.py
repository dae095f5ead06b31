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


@pytest.mark.parametrize("cfg", [(20, 2, 32, 0), (10, 1, 32, 0), (13, 1, 32, 1)], ids=lambda c: "n%d_g%d_hw%d_head%d" % c)
def test_merged_weight_gradient_launch_is_bit_identical_to_per_layer_launches(cfg, tmp_path):
    """conv_wgrad_multi_kernel (every layer of a replay-sized pass in one launch at the end of the backward) runs the per-layer kernel's
    body on the same slabs: with the same pixel split (OCL_WGRAD_MULTI_TARGET=0) not one bit of any gradient, output or running statistic
    differs from the per-layer launches (OCL_WGRAD_MULTI=0)."""
    ref = str(tmp_path / "ref.bin")
    rc, out = _run(cfg, "write", ref, {"OCL_WGRAD_MULTI": "0", "OCL_WGRAD_MULTI_TARGET": "0"})
    assert rc == 0, out
    rc, out = _run(cfg, "compare", ref, {"OCL_WGRAD_MULTI_TARGET": "0"})
    assert rc == 0, out
    m = re.search(r"(\d+) of (\d+) tensors differ in some bit", out)
    assert m and int(m.group(1)) == 0 and int(m.group(2)) >= 60, out


@pytest.mark.parametrize("cfg", [(20, 2, 32, 0), (47, 1, 32, 1)], ids=lambda c: "n%d_g%d_hw%d_head%d" % c)
def test_merged_weight_gradient_launch_at_its_own_pixel_split_differs_by_rounding_only(cfg, tmp_path):
    """The product's split inside the merged launch (96 workgroups per layer instead of 512) sums the pixel tiles in another order: the
    convolution weights' gradients may differ by rounding, nothing else may differ at all."""
    ref = str(tmp_path / "ref.bin")
    rc, out = _run(cfg, "write", ref, {"OCL_WGRAD_MULTI": "0", "OCL_WGRAD_MULTI_TARGET": "0"})
    assert rc == 0, out
    rc, out = _run(cfg, "compare", ref, {})
    assert rc == 0, out
    differing = re.findall(r"^\s+(\S+)\s+\d+ floats\s+reldiff (\S+)", out, re.M)
    assert all(("conv" in n or "shortcut.0" in n) and n.endswith(".weight") for n, _ in differing), out
    assert all(float(v) < 5e-6 for _, v in differing), out


def test_two_stream_backward_with_one_gradient_buffer_per_layer_is_bit_identical_to_the_ring(tmp_path):
    """SCR's 220-view pass: per-layer dL/dy buffers and a hand-over to the weight-gradient stream every third layer change WHEN kernels run,
    not what they compute."""
    cfg = (220, 2, 32, 1)
    ref = str(tmp_path / "ref.bin")
    rc, out = _run(cfg, "write", ref, {"OCL_DY_KEEP": "0"})
    assert rc == 0, out
    rc, out = _run(cfg, "compare", ref, {})
    assert rc == 0, out
    m = re.search(r"(\d+) of (\d+) tensors differ in some bit", out)
    assert m and int(m.group(1)) == 0, out


@pytest.mark.parametrize("cfg", [(20, 2, 32, 0), (20, 1, 32, 1), (13, 1, 32, 0), (32, 2, 32, 0)], ids=lambda c: "n%d_g%d_hw%d_head%d" % c)
def test_channel_partitioned_batchnorm_backward_differs_by_rounding_only(cfg, tmp_path):
    """bn_bwd_chan_kernel (layer 4 of a replay-sized pass: one workgroup per channel quad, no cross-workgroup reduction) sums the pixels in
    another order than bn_bwd_fused_kernel (OCL_BN_CHAN=0): every tensor stays within 1e-4 of its largest entry (netcheck's exit code),
    the observed differences are ~1e-7."""
    ref = str(tmp_path / "ref.bin")
    rc, out = _run(cfg, "write", ref, {"OCL_BN_CHAN": "0"})
    assert rc == 0, out
    rc, out = _run(cfg, "compare", ref, {})
    assert rc == 0, out
    m = re.search(r"(\d+) of (\d+) tensors differ in some bit, (\d+) beyond", out)
    assert m and int(m.group(3)) == 0 and int(m.group(2)) >= 60, out
    differing = re.findall(r"^\s+(\S+)\s+\d+ floats\s+reldiff (\S+)", out, re.M)
    assert all(float(v) < 1e-5 for _, v in differing), out
    rc, out2 = _run(cfg, "compare", ref, {"OCL_BN_CHAN": "0"})     # the switch is read: the same mode again is bit-identical
    m = re.search(r"(\d+) of (\d+) tensors differ in some bit", out2)
    assert rc == 0 and m and int(m.group(1)) == 0, out2
