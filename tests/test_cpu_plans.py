"""CPU: the convolution / weight-gradient planner (host code of csrc/conv.hip and csrc/wgrad.hip) through `kbench ... plan`, which needs no GPU.

Every layer of Reduced-ResNet18 at the batch sizes the path uses (replay-sized, the SCR step's 220 views, the 410-image eval-mode
pass, mini-ImageNet's 84x84) must get a tiling that fits the LDS; the experimental three-buffer weight ring (OCL_CONV_PIPE=1,
DESIGN §4.1 (c)) must only ever replace a staged-weight plan and keep the stage geometry its kernel is compiled for."""
import os
import re
import subprocess

import pytest

from conftest import ROOT

CSRC = os.path.join(ROOT, "online-continual-learning_amd", "csrc")
KBENCH = os.path.join(CSRC, "kbench")
LDS_LIMIT = 160 * 1024
PIPE_QS = {1: 64, 2: 32, 3: 16, 4: 16, 5: 16}      # conv.hip: pipe_qs(MT)


def _plan_lines(n, groups, hw, env=None):
    if not os.path.exists(KBENCH):
        subprocess.run(["make", "-C", CSRC, "kbench"], check=True, stdout=subprocess.DEVNULL)
    r = subprocess.run([KBENCH, str(n), str(groups), str(hw), "plan"], capture_output=True, text=True, timeout=120,
                       env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return r.stdout.splitlines()


def _fields(line):
    return {k: int(v) for k, v in re.findall(r"(\w+)= *(-?\d+)", line)}


@pytest.mark.parametrize("n,groups,hw", [(10, 1, 32), (20, 2, 32), (100, 2, 32), (220, 2, 32), (410, 1, 32), (220, 2, 84), (15, 1, 84)])
def test_every_layer_gets_a_plan_that_fits(n, groups, hw):
    lines = _plan_lines(n, groups, hw)
    conv = [l for l in lines if re.search(r"\s(fwd|dgrad)\s", l)]
    wgrad = [l for l in lines if " wgrad " in l]
    ring = [l for l in lines if "ring:" in l]
    assert len(wgrad) == 20 and len(conv) >= 39          # 20 convolutions; forward + at least one data-gradient plan each (not the stem)
    for l in conv:
        f = _fields(l)
        assert 1 <= f["MT"] <= 5 and 1 <= f["NT"] <= 2
        assert f["lds"] <= LDS_LIMIT - 2048
        assert f["Qpad"] % 4 == 0 and f["KC"] % 4 == 0
        if f["res"]:
            assert f["QS"] == f["Qpad"]                   # resident weights: one "stage" = everything
        else:
            assert f["QS"] % 4 == 0 and 0 < f["QS"] <= f["Qpad"]
        if f["classes"] > 1:
            assert f["NT"] == 1 and f["classes"] == 4    # merged parity classes: one pixel tile per wave
    for l in wgrad:
        f = _fields(l)
        assert f["lds"] <= LDS_LIMIT - 1024
        assert f["S"] >= 1 and f["partial"] >= 0
    # a ring plan follows the staged plan it would replace, keeps its tiling and uses the compiled stage geometry
    for i, l in enumerate(lines):
        if "ring:" not in l:
            continue
        base, f = _fields(lines[i - 1]), _fields(l)
        assert base["res"] == 0, "a ring plan for resident weights"
        assert f["MT"] == base["MT"] and f["NT"] == 1 and f["KC"] == base["KC"] and f["Qpad"] == base["Qpad"]
        assert f["QS"] == PIPE_QS[f["MT"]]
        assert f["lds"] <= LDS_LIMIT - 2048
        # three stage buffers of QS groups x 16*MT channels x 16 bytes instead of two of the two-buffer plan's
        assert f["lds"] - base["lds"] == 3 * f["QS"] * 16 * f["MT"] * 16 - 2 * base["QS"] * 16 * base["MT"] * 16
    if hw == 32 and n >= 220:
        assert ring, "layers 3 - 4 stream their weights at these sizes"
    # conv_s_kernel (K split over the four waves): layer 4 at every batch size, layer 3 below ~200 images (two pixel tiles per workgroup
    # once the one-tile grid has >= 800 workgroups); one wave's slice in <= 64 KB
    cs = [l for l in conv if l.rstrip().endswith("conv_s")]
    if hw == 32:
        assert any(l.startswith("layer4.1.conv2") for l in cs)
        assert any(l.startswith("layer3.1.conv2") for l in cs) == (n <= 100)
    for l in cs:
        f = _fields(l)
        assert f["classes"] == 1 and f["MT"] == 1 and f["res"] == 0 and f["lds"] <= 64 * 1024 and f["Qpad"] % 16 == 0
        if hw == 32:
            assert f["NT"] == (2 if (l.startswith("layer3") and n == 100) else 1)
        else:
            assert f["NT"] in (1, 2) and f["ppi"] == 16 * f["NT"]     # 11x11 / 21x21 lattices: runs of 16 / 32 pixels of one image
        assert not l.startswith(("conv1", "layer1", "layer2"))


@pytest.mark.parametrize("n,groups,hw", [(1, 1, 32), (7, 1, 32), (10, 1, 32), (20, 2, 32), (50, 2, 32), (100, 1, 32), (150, 2, 32), (220, 2, 32), (272, 1, 32),
                                         (410, 1, 32), (1, 1, 84), (6, 1, 84), (15, 1, 84), (20, 2, 84), (50, 1, 84), (60, 1, 84), (220, 2, 84)])
def test_every_plan_covers_its_output_exactly_once(n, groups, hw):
    """`kbench ... cover` replays every plan's tables on the host (the lane -> pixel maps of conv_t_kernel / conv_q_kernel /
    conv_s_kernel, aligned and per-tile): every lattice pixel written exactly once, operand reads and patch units inside the patch,
    patch loads inside the input tensor -- at batch sizes the GPU suite never runs as well."""
    if not os.path.exists(KBENCH):
        subprocess.run(["make", "-C", CSRC, "kbench"], check=True, stdout=subprocess.DEVNULL)
    r = subprocess.run([KBENCH, str(n), str(groups), str(hw), "cover"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and " 0 with errors" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("n,groups,hw", [(7, 1, 32), (20, 2, 32), (128, 2, 32), (220, 2, 32), (410, 1, 32), (15, 1, 84)])
def test_conv_w_plans_cover_their_output_exactly_once(n, groups, hw):
    """The same replay for conv_w_kernel / conv_wx_kernel (csrc/convw.hip) wherever their planner accepts a geometry (OCL_CONV_W=2; the
    product takes them only for the hot 3x3 layers of large passes): the wave tiles' arithmetic geometry, the lane -> pixel table, every
    tap's patch slot against the staging table, the staged bytes against the input tensor, chunked and unchunked."""
    r = subprocess.run([KBENCH, str(n), str(groups), str(hw), "cover"], capture_output=True, text=True, timeout=300, env=dict(os.environ, OCL_CONV_W="2"))
    assert r.returncode == 0 and " 0 with errors" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_the_coverage_check_detects_a_wrong_pixel_map():
    """Self-test of the checker: unaligned plans read as aligned must fail it."""
    env = dict(os.environ, KBENCH_COVER_SELFTEST="1")
    r = subprocess.run([KBENCH, "6", "1", "84", "cover"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 1 and "COVER" in r.stdout, r.stdout[-2000:]


def test_ring_schedule_index_model():
    """The index arithmetic of the ring (prefetch cursor, buffer rotation, operand fetches across stage boundaries) replayed for random
    plans: scripts/ring_schedule_model.py mirrors the control flow of conv_t_kernel's `seq`."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ring_schedule_model", os.path.join(ROOT, "scripts", "ring_schedule_model.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    assert m.main(400, seed=7) == 400


def test_weight_gradient_forms_chosen_by_the_planner():
    """The 4x4x1 form (the default since round 5; OCL_WGRAD_Q=0 keeps the 16x16x4 form everywhere) is only planned where it was
    measured faster: layer 1's 3x3 convolutions (<= 20 output channels, one channel chunk, not the stem) of a pass with at least ~6
    tiles of 128 pixels per workgroup -- SCR's 220 views, not a replay-sized batch, not 20 images of 84x84; its slabs are 20 columns
    wide and its pixel tiles whole 16-pixel steps."""
    assert not any(_fields(l)["q4"] for l in _plan_lines(220, 2, 32, {"OCL_WGRAD_Q": "0"}) if " wgrad " in l)
    q = {l.split()[0]: _fields(l) for l in _plan_lines(220, 2, 32) if " wgrad " in l}
    on = sorted(k for k, f in q.items() if f["q4"])
    assert on == ["layer1.0.conv1", "layer1.0.conv2", "layer1.1.conv1", "layer1.1.conv2"]
    for k in on:
        f = q[k]
        assert f["q4"] == 3 and f["KP"] % 16 == 0 and f["CP"] % 4 == 0 and f["KC"] == 20 and f["lds"] <= 72 * 1024
        assert 128 <= f["grid"] <= 256                    # one workgroup per CU: a single wave of workgroups
    for n, groups, hw in [(20, 1, 32), (20, 1, 84)]:
        assert not any(_fields(l)["q4"] for l in _plan_lines(n, groups, hw) if " wgrad " in l)
    assert any(_fields(l)["q4"] for l in _plan_lines(20, 1, 84, {"OCL_WGRAD_Q": "2"}) if " wgrad " in l)   # (2 lifts the size gate)


def test_xcd_aware_workgroup_order_is_planned_where_output_blocks_share_pixel_tiles():
    """The XCD-aware order of conv_wgrad_kernel's workgroups (round 5: measured bit-identical and faster, the default) is planned for
    launches with more than one output block and at least 8 pixel splits, and carries the plan's own block count."""
    some = 0
    for n, groups, hw in [(220, 2, 32), (20, 1, 32), (20, 1, 84)]:
        for l in [l for l in _plan_lines(n, groups, hw) if " wgrad " in l]:
            f = _fields(l)
            gx, gy = (int(v) for v in re.search(r"grid= *(\d+)x *(\d+)", l).groups())
            assert f["xcd"] == (gy if gy > 1 and gx >= 8 else 0) and gx == f["S"]
            some += f["xcd"] > 0
    assert some >= 10


def test_eighty_channel_weight_gradient_blocks_are_planned_by_pass_size():
    """conv_wgrad_kernel's 80-channel output blocks (NTW = 5: layers 3 - 4 exactly, no padded columns) are planned on passes of 48 images
    and more -- faster alone at every size, but beside the dependent chain of a 20-image pass the whole pass got slower
    (profiles/r5_wgrad_ntw5.txt): layers 1 - 2 (<= 40 channels) never, layers 3 - 4 from 48 images on, with slabs of at most 12 MB."""
    for n, groups, hw, wide in [(220, 2, 32, True), (64, 2, 32, True), (48, 2, 32, True), (50, 1, 84, True), (20, 1, 32, False), (20, 1, 84, False),
                                (46, 2, 32, False)]:
        for l in [l for l in _plan_lines(n, groups, hw) if " wgrad " in l]:
            f, name = _fields(l), l.split()[0]
            if name.startswith(("layer3", "layer4")):
                assert f["NTW"] == (5 if wide else 3), l
            else:
                assert f["NTW"] <= 3, l
            assert float(re.search(r"partial= *([\d.]+) MB", l).group(1)) <= 12.6, l


def test_merged_weight_gradient_launch_gets_a_form_for_every_layer_of_a_replay_sized_pass():
    """conv_wgrad_multi_kernel<0> (every layer of a replay-sized pass in one launch) holds the 64-row forms: split as net.hip asks for the merged
    launch (96 workgroups per layer) every layer of a 10 / 13 / 20 / 47-image pass must plan one of them (`multi` = its index, 0 .. 3), with at
    most ~96 workgroups where the output blocks alone do not exceed that, and slabs that fit side by side in the 64 MB workspace."""
    for n, groups in [(10, 1), (13, 1), (20, 2), (47, 1)]:
        lines = [l for l in _plan_lines(n, groups, 32, {"KBENCH_WG_TARGET": "96"}) if " wgrad " in l]
        assert len(lines) == 20
        total_mb = 0.0
        for l in lines:
            f = _fields(l)
            gx, gy = (int(v) for v in re.search(r"grid= *(\d+)x *(\d+)", l).groups())
            assert f["MTW"] == 1 and 0 <= f["multi"] <= 3, l
            assert gx * gy <= max(gy, 96 + gy), l
            total_mb += float(re.search(r"partial= *([\d.]+) MB", l).group(1))
        assert total_mb < 64.0
    # as a launch of its own the 220-view pass keeps forms the merged kernels do not all have: it is never merged (two streams)
    assert any(_fields(l)["multi"] < 0 for l in _plan_lines(220, 2, 32) if " wgrad " in l)
