"""GPU: the Reduced-ResNet18 / SupConResNet engine (conv implicit-GEMM fwd / dgrad / wgrad, BatchNorm, pool, heads)
against the torch-fp32 oracle restatement (oracle.OracleNet, itself pinned against the reference modules).

Tolerances (fp32 everywhere, different summation order than ATen): layer outputs 1e-5 relative-to-max per layer,
network outputs / losses 1e-4 abs, gradients 1e-3 relative to the tensor's max |g| (BatchNorm's 1/std amplifies
round-off over 20 layers)."""
import ctypes as C

import numpy as np
import pytest
import torch

from conftest import gold
from oracle import ocl_oracle as O

pytestmark = pytest.mark.gpu


def build(agent, data, head="mlp", seed=11, cuda=None, max_batch=64):
    from types import SimpleNamespace
    from ocl_amd.setup_elements import setup_architecture
    torch.manual_seed(seed)
    m = setup_architecture(SimpleNamespace(agent=agent, data=data, head=head))
    m.max_batch = max_batch
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m = m.cuda()
    return m, sd


def relmax(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / (1e-12 + np.abs(b).max())


def layer_report(m, net_rec, n):
    """Per-conv raw output error (NHWC engine tape vs oracle NCHW record); returns list of (name, relerr)."""
    from ocl_amd import ffi
    L = ffi.lib()
    rows = []
    for i, (name, ref) in enumerate(net_rec.items()):
        ref = ref.numpy()
        cnt = ref.size
        dst = torch.empty(cnt, dtype=torch.float32, device="cuda")
        nw = ffi.i64(0)
        slot = (m._slot_rr - 1) % m._n_tapes
        ffi.check(L.ocl_net_debug_copy(m._net, slot, 0, i, ffi.ptr(dst), cnt, C.byref(nw), ffi.stream()))
        got = dst.cpu().numpy().reshape(ref.shape[0], ref.shape[2], ref.shape[3], ref.shape[1]).transpose(0, 3, 1, 2)
        rows.append((name, relmax(got, ref)))
    return rows


CASES = [
    # agent, data, head, n, BatchNorm groups (views of one pass)
    ("ER", "cifar100", None, 10, 1),
    ("ER", "cifar10", None, 20, 1),
    ("ER", "cifar100", None, 3, 1),
    ("SCR", "cifar100", "mlp", 14, 2),
    ("SCR", "cifar100", "mlp", 220, 2),
    ("SCR", "cifar100", "mlp", 36, 2),
    ("SCR", "cifar100", "mlp", 50, 2),      # layer 1 on conv_q_kernel's 256-pixel tiles, layer 3 on conv_s_kernel with two pixel tiles
    ("ER", "cifar100", None, 100, 1),       # the same, one group; layer 4's 1000-workgroup conv_s launches
    ("ER", "mini_imagenet", None, 6, 1),
    ("ER", "mini_imagenet", None, 20, 2),
    ("SCR", "cifar100", "linear", 8, 2),
]

BLOCKS = ["layer%d.%d" % (l, b) for l in range(1, 5) for b in range(2)]


def fetch_act(m, what, index, shape_nchw):
    from ocl_amd import ffi
    n, c, h, w = shape_nchw
    cnt = n * c * h * w
    dst = torch.empty(cnt, dtype=torch.float32, device="cuda")
    nw = ffi.i64(0)
    slot = (m._slot_rr - 1) % m._n_tapes
    ffi.check(ffi.lib().ocl_net_debug_copy(m._net, slot, what, index, ffi.ptr(dst), cnt, C.byref(nw), ffi.stream()))
    return dst.cpu().view(n, h, w, c).permute(0, 3, 1, 2).contiguous()


def engine_masks(m, shapes, pre):
    """ReLU activation patterns of the engine's last train-mode forward, keyed like OracleNet._relu."""
    masks = {"z:stem": (fetch_act(m, 5, 0, shapes["z:stem"]) > 0).float()}
    for bi, p in enumerate(BLOCKS):
        masks["a1:" + pre + p] = (fetch_act(m, 4, bi, shapes["a1:" + pre + p]) > 0).float()
        masks["z:" + pre + p] = (fetch_act(m, 1, bi, shapes["z:" + pre + p]) > 0).float()
    return masks


@pytest.mark.parametrize("agent,data,head,n,groups", CASES)
def test_train_forward_backward_vs_oracle(cuda, agent, data, head, n, groups):
    """Forward: raw conv outputs, network output, loss, running statistics vs the oracle.  Backward: every parameter
    gradient vs autograd on the oracle with the ReLU activation pattern teacher-forced to the engine's (a ReLU input
    within fp32 round-off of zero may land on either side — about one element per 10^5 — and the gradient is
    discontinuous there); every element whose pattern differs from ATen's must be such an ambiguous one."""
    hw = 84 if data == "mini_imagenet" else 32
    m, sd = build(agent, data, head or "mlp", cuda=cuda, max_batch=max(64, n))
    m._ensure_bound()
    pre = "encoder." if head is not None else ""
    rng = np.random.default_rng(n)
    x = rng.random((n, 3, hw, hw)).astype(np.float32)
    y = rng.integers(0, 10, n).astype(np.int64)
    xt = torch.from_numpy(x)
    per = n // groups
    # ---- engine forward
    m.train()
    xd = torch.from_numpy(x).to(cuda)
    out = m.forward(xd) if groups == 1 else m.forward_views([xd[g * per:(g + 1) * per] for g in range(groups)])
    torch.cuda.synchronize()
    # ---- oracle forward (one call per group = per view), masks teacher-forced
    st = O.clone_state(sd)
    net = O.OracleNet(st, head=head, training=True)
    net.rec = {}
    probe = O.OracleNet(O.clone_state(sd, requires_grad=False), head=head, training=True)
    probe.pre_act = {}
    with torch.no_grad():
        probe.forward(xt[:per])
    shapes = {k: (n,) + tuple(v.shape[1:]) for k, v in probe.pre_act.items()}
    masks = engine_masks(m, shapes, pre)
    outs, n_mis, n_tot, worst_amb = [], 0, 0, 0.0
    for g in range(groups):
        net.mask_override = {k: v[g * per:(g + 1) * per] for k, v in masks.items()}
        net.pre_act = {}
        outs.append(net.forward(xt[g * per:(g + 1) * per]))
        for k, pa in net.pre_act.items():
            mis = (pa > 0).float() != net.mask_override[k]
            n_mis += int(mis.sum())
            n_tot += pa.numel()
            if mis.any():
                worst_amb = max(worst_amb, float(pa[mis].abs().max() / pa.abs().max()))
    o_ref = torch.cat(outs, 0)
    print("relu pattern mismatches: %d of %d, worst |pre-activation| / max = %.2e" % (n_mis, n_tot, worst_amb))
    assert n_mis <= 5 + 2e-5 * n_tot and worst_amb < 1e-5, "activation patterns differ beyond fp32 round-off"
    if groups == 1:
        rows = layer_report(m, net.rec, n)
        print("\n".join("%-40s %.3e" % r for r in rows))
        assert max(r[1] for r in rows) < 1e-4, "raw conv outputs diverge: %s" % (max(rows, key=lambda r: r[1]),)
    err_out = np.abs(out.detach().cpu().numpy() - o_ref.detach().numpy()).max()
    print("out err", err_out)
    assert err_out < 1e-4
    w = torch.linspace(-1, 1, o_ref.numel()).view_as(o_ref)
    if head is None:
        from ocl_amd.loss import cross_entropy_mean
        loss_ref = torch.nn.functional.cross_entropy(o_ref, torch.from_numpy(y))
        loss = cross_entropy_mean(out, torch.from_numpy(y).to(cuda))
    else:
        loss_ref = (o_ref * w).sum()
        loss = (out * w.to(cuda)).sum()
    assert abs(float(loss) - float(loss_ref.detach())) < 1e-4 * max(1.0, abs(float(loss_ref.detach())))
    loss_ref.backward()
    loss.backward()
    torch.cuda.synchronize()
    worst, rows = 0.0, []
    for k, p in m.named_parameters():
        gref = st[k].grad
        if gref is None:
            assert float(p.grad.abs().max()) == 0.0, "%s should have a zero gradient" % k
            continue
        e = np.abs(p.grad.cpu().numpy() - gref.numpy()).max() / (1e-12 + float(gref.abs().max()))
        rows.append((k, e, float(gref.abs().max())))
        worst = max(worst, e)
    print("\n".join("%-44s rel %.3e  max|g| %.3e" % r for r in rows))
    assert worst < 2e-4, max(rows, key=lambda r: r[1])
    # BatchNorm running statistics (momentum 0.1, unbiased variance, one update per forward call / group)
    sd_new = m.state_dict()
    for k in sd_new:
        if k.endswith("running_mean") or k.endswith("running_var"):
            assert relmax(sd_new[k].cpu().numpy(), st[k].numpy()) < 1e-4, k
        if k.endswith("num_batches_tracked"):
            assert int(sd_new[k]) == int(st[k]) == groups


def test_gradient_accumulation_and_zero_grad_semantics(cuda):
    """loss.backward() twice accumulates (agents/exp_replay.py:55,77); opt.zero_grad() makes the next one overwrite;
    FusedSGD.step equals torch.optim.SGD.step."""
    from ocl_amd.setup_elements import setup_opt
    from ocl_amd.loss import cross_entropy_mean
    m, sd = build("ER", "cifar10", cuda=cuda)
    opt = setup_opt("SGD", m, 0.1, 0)
    rng = np.random.default_rng(1)
    xa, xb = rng.random((10, 3, 32, 32)).astype(np.float32), rng.random((10, 3, 32, 32)).astype(np.float32)
    ya, yb = rng.integers(0, 10, 10), rng.integers(0, 10, 10)
    st = O.clone_state(sd)
    names = [k for k in st if st[k].requires_grad]
    net = O.OracleNet(st, training=True)
    O.ce_mean(net.forward(torch.from_numpy(xa)), torch.from_numpy(ya)).backward()
    O.ce_mean(net.forward(torch.from_numpy(xb)), torch.from_numpy(yb)).backward()
    gref = O.flat_grad(st, names).numpy()
    m.train()
    opt.zero_grad()
    cross_entropy_mean(m.forward(torch.from_numpy(xa).to(cuda)), torch.from_numpy(ya).to(cuda)).backward()
    cross_entropy_mean(m.forward(torch.from_numpy(xb).to(cuda)), torch.from_numpy(yb).to(cuda)).backward()
    g = m.flat_grads().cpu().numpy()
    assert np.linalg.norm(g - gref) < 5e-2 * np.linalg.norm(gref)   # loose: ReLU sign flips at ~0 (see the teacher-forced test)
    O.sgd_step(st, names, 0.1)
    opt.step()
    pref = torch.cat([st[k].detach().reshape(-1) for k in names]).numpy()
    assert np.abs(m.flat_params().cpu().numpy() - pref).max() < 5e-3 * np.abs(pref).max()
    # zero_grad then ONE backward overwrites.  Both sides start this phase from the SAME weights (the oracle's): the two updates
    # above differ by ReLU sign flips, and a gradient taken at slightly different weights is a different question
    m.load_state_dict({k: v.detach() for k, v in st.items()})
    O.zero_grad(st, names)
    O.ce_mean(O.OracleNet(st, training=True).forward(torch.from_numpy(xa)), torch.from_numpy(ya)).backward()
    opt.zero_grad()
    cross_entropy_mean(m.forward(torch.from_numpy(xa).to(cuda)), torch.from_numpy(ya).to(cuda)).backward()
    g1 = O.flat_grad(st, names).numpy()
    assert np.linalg.norm(m.flat_grads().cpu().numpy() - g1) < 5e-2 * np.linalg.norm(g1)
    # p.grad views alias the flat gradient (get_grad_vector layout)
    off = 0
    for p in m.parameters():
        assert p.grad.data_ptr() == m.flat_grads().data_ptr() + 4 * off
        off += p.numel()


@pytest.mark.parametrize("agent,data,head,n", [("ER", "cifar100", None, 70), ("SCR", "cifar100", "mlp", 33), ("ER", "mini_imagenet", None, 5)])
def test_eval_forward_features_vs_oracle(cuda, agent, data, head, n):
    """eval-mode (running statistics) features / outputs: the ASER scoring and NCM path (utils/utils.py:45-90)."""
    hw = 84 if data == "mini_imagenet" else 32
    m, sd = build(agent, data, head or "mlp", cuda=cuda, max_batch=32)
    # make the running statistics non-trivial
    rng = np.random.default_rng(2)
    for k in sd:
        if k.endswith("running_mean"):
            sd[k] = torch.from_numpy(rng.standard_normal(sd[k].shape).astype(np.float32) * 0.1)
        if k.endswith("running_var"):
            sd[k] = torch.from_numpy((0.5 + rng.random(sd[k].shape)).astype(np.float32))
    m.load_state_dict(sd)
    x = rng.random((n, 3, hw, hw)).astype(np.float32)
    st = O.clone_state(sd, requires_grad=False)
    net = O.OracleNet(st, head=head, training=False)
    with torch.no_grad():
        f_ref = net.features(torch.from_numpy(x)).numpy()
        o_ref = net.forward(torch.from_numpy(x)).numpy()
    m.eval()
    with torch.no_grad():
        f = m.features_batched(torch.from_numpy(x).to(cuda)).cpu().numpy()     # n > max_batch: chunked
        o = torch.cat([m.forward(torch.from_numpy(x[i:i + 32]).to(cuda)) for i in range(0, n, 32)]).cpu().numpy()
    assert relmax(f, f_ref) < 1e-4 and np.abs(o - o_ref).max() < 1e-4
    sd2 = m.state_dict()
    for k in sd:
        if "running" in k or "num_batches" in k:
            assert torch.equal(sd2[k].cpu(), sd[k]), "eval forward must not touch BatchNorm buffers"


def test_eval_forward_one_large_batch_with_rounded_plans(cuda):
    """One eval-mode pass of 150 images (the shape of ASER's scoring passes): its plans are made for the batch rounded up to 160
    (net.hip: eval-mode plans are bucketed to 16 images; the extra images are whatever the buffers hold), the 20-channel layers run on
    conv_q_kernel's folded-BatchNorm epilogue (>= 128 tiles of 512 pixels).  Features of the 150 real images against the oracle, and a
    second pass of 146 images -- same plan set, different garbage in the padding -- must reproduce the first 146 rows bit for bit."""
    m, sd = build("ER", "cifar100", "mlp", cuda=cuda, max_batch=256)
    rng = np.random.default_rng(4)
    for k in sd:
        if k.endswith("running_mean"):
            sd[k] = torch.from_numpy(rng.standard_normal(sd[k].shape).astype(np.float32) * 0.1)
        if k.endswith("running_var"):
            sd[k] = torch.from_numpy((0.5 + rng.random(sd[k].shape)).astype(np.float32))
    m.load_state_dict(sd)
    x = rng.random((150, 3, 32, 32)).astype(np.float32)
    net = O.OracleNet(O.clone_state(sd, requires_grad=False), head=None, training=False)
    with torch.no_grad():
        f_ref = net.features(torch.from_numpy(x)).numpy()
    m.eval()
    xd = torch.from_numpy(x).to(cuda)
    with torch.no_grad():
        f = m.features_batched(xd).cpu().numpy()
        f2 = m.features_batched(xd[:146]).cpu().numpy()
    assert f.shape == f_ref.shape and relmax(f, f_ref) < 1e-4
    assert np.array_equal(f2, f[:146])


def test_eval_padding_images_do_not_leak_into_the_real_rows(cuda):
    """Eval-mode plans are made for the batch rounded up (4 images up to 16, 16 above: net.hip): the padding images are computed from
    whatever the activation buffers hold.  Seed those buffers with NaN (a 16-image pass of NaN inputs), then run 1, 3, 5, 10 and 17
    images: every feature of the real rows must be finite and equal, bit for bit, to the same rows out of an engine that never saw a
    NaN -- no cross-image term may reach from the padding into a real image."""
    rng = np.random.default_rng(8)
    x = rng.random((17, 3, 32, 32)).astype(np.float32)
    clean, _ = build("ER", "cifar100", "mlp", cuda=cuda, max_batch=64)
    dirty, _ = build("ER", "cifar100", "mlp", cuda=cuda, max_batch=64)
    clean.eval(); dirty.eval()
    xd = torch.from_numpy(x).to(cuda)
    nan16 = torch.full((16, 3, 32, 32), float("nan"), device=cuda)
    nan32 = torch.full((32, 3, 32, 32), float("nan"), device=cuda)
    with torch.no_grad():
        for n in (1, 3, 5, 10, 17):
            dirty.features_batched(nan32 if n > 16 else nan16)
            got = dirty.features_batched(xd[:n]).cpu().numpy()
            ref = clean.features_batched(xd[:n]).cpu().numpy()
            assert np.isfinite(got).all(), n
            assert np.array_equal(got, ref), n


def test_virtual_params_forward_does_not_touch_model(cuda):
    """MIR's theta - lr*grad forward (mir_retrieve.py:21,25) through params_override."""
    m, sd = build("ER", "cifar100", cuda=cuda)
    rng = np.random.default_rng(3)
    x = rng.random((12, 3, 32, 32)).astype(np.float32)
    delta = (rng.standard_normal(m.flat_params().numel()) * 0.01).astype(np.float32) if False else None
    m.train()
    m._ensure_bound()
    shadow = m.flat_params().clone()
    shadow += torch.from_numpy((rng.standard_normal(shadow.numel()) * 0.01).astype(np.float32)).to(cuda)
    before = {k: v.clone() for k, v in m.state_dict().items()}
    with torch.no_grad():
        out_v = m.forward_with_params(torch.from_numpy(x).to(cuda), shadow).cpu().numpy()
    after = m.state_dict()
    for k in before:
        assert torch.equal(before[k], after[k]), "virtual forward modified %s" % k
    st = O.clone_state(sd, requires_grad=False)
    o = 0
    for k in [kk for kk, _ in m.named_parameters()]:
        nel = st[k].numel()
        st[k] = shadow[o:o + nel].view_as(st[k]).cpu()
        o += nel
    with torch.no_grad():
        ref = O.OracleNet(st, training=True).forward(torch.from_numpy(x)).numpy()
    assert np.abs(out_v - ref).max() < 1e-4


def test_same_weights_reuses_the_packs_and_every_writer_invalidates_them(cuda):
    """OCL_FWD_SAME_WEIGHTS (include/ocl_hip.h): inside `with model.same_weights()` forwards that follow a forward without a step in
    between reuse the engine's weight packs -- same outputs, bit for bit, in eval and train mode; FusedSGD.step() reports itself, so the
    next forward re-packs (its outputs follow the NEW weights: compared with a model that never carries the flag); a params_override
    forward in between makes the engine re-pack by itself; and the raw flag on changed weights DOES give the stale result (the flag
    is honoured, i.e. the test can see it)."""
    from ocl_amd import ops
    from ocl_amd.optim import FusedSGD
    ops.set_deterministic(True)      # (bit-for-bit comparisons of train-mode passes need the order-independent batch sums)
    try:
        _same_weights_body(cuda, FusedSGD)
    finally:
        ops.set_deterministic(False)


def _same_weights_body(cuda, FusedSGD):
    rng = np.random.default_rng(17)
    x = torch.from_numpy(rng.random((12, 3, 32, 32)).astype(np.float32)).to(cuda)
    y = torch.from_numpy(rng.random((12, 100)).astype(np.float32)).to(cuda)
    outs = {}
    for tag in ("flag", "plain"):
        m, _ = build("ER", "cifar100", cuda=cuda)
        opt = FusedSGD(m, 0.1)
        ctx = m.same_weights if tag == "flag" else __import__("contextlib").nullcontext
        seq = []
        with ctx():
            m.eval()
            with torch.no_grad():
                seq.append(m.forward(x))            # packs
                seq.append(m.forward(x))            # flag: reuses (eval: no launch at all in front of the convolutions)
            m.train()
            out = m.forward(x)                      # flag, train mode: the statistics arenas are cleared by a memset instead
            seq.append(out.detach().clone())
            opt.zero_grad()
            (out * y).sum().backward()
            opt.step()                              # writes the weights: the next forward must re-pack
            with torch.no_grad():
                seq.append(m.forward(x))
                shadow = m.flat_params().clone() * 1.01
                seq.append(m.forward_with_params(x, shadow))   # the arena now holds the shadow's packs
                seq.append(m.forward(x))            # flag set by Python, refused by the engine (pack_src differs): re-packs
            # The ASER loop's order after a step: an eval-mode feature pass FIRST (it packs the forward layout; the data-gradient packs
            # in the arena are the previous step's), then a taped pass + backward.  The first version of the flag trusted the stale
            # data-gradient pack here: one more step, and the gradient itself, must match the model that never carries the flag.
            out = m.forward(x)
            opt.zero_grad()
            (out * y).sum().backward()
            opt.step()
            m.eval()
            with torch.no_grad():
                seq.append(m.features(x))
            m.train()
            m.forward_stats_only(x)                 # (and a pass run for its running statistics alone in between)
            out = m.forward(x)
            opt.zero_grad()
            (out * y).sum().backward()
            seq.append(m.flat_grads().clone())
            seq.append(torch.cat([v.flatten().float() for k, v in m.state_dict().items() if "running" in k]))
        outs[tag] = [t.cpu().numpy() for t in seq]
        if tag == "flag":
            with torch.no_grad():
                m.eval()
                with m.same_weights():
                    before = m.forward(x).cpu().numpy()
                    # a torch write nobody reported: the version counters see it, the next forward re-packs
                    m.flat_params().mul_(1.05)
                    seen = m.forward(x).cpu().numpy()
                    dict(m.named_parameters())["layer2.0.conv1.weight"].mul_(1.05)
                    seen2 = m.forward(x).cpu().numpy()
                    # ... and the RAW flag on changed weights does give the stale convolutions (the flag is honoured: this test can see it)
                    m.flat_params().mul_(1.05)
                    m._packed_version = m._weights_version()
                    stale = m.forward(x).cpu().numpy()
                fresh = m.forward(x).cpu().numpy()
            assert not np.array_equal(seen, before) and not np.array_equal(seen2, seen) and not np.array_equal(stale, fresh)
            m2, _ = build("ER", "cifar100", cuda=cuda)
            m2.load_state_dict(m.state_dict())
            m2.eval()
            with torch.no_grad():
                assert np.array_equal(m2.forward(x).cpu().numpy(), fresh)
    for a, b in zip(outs["flag"], outs["plain"]):
        assert np.array_equal(a, b)
    assert np.array_equal(outs["flag"][0], outs["flag"][1])
    assert not np.array_equal(outs["flag"][3], outs["flag"][2])


def test_resnet_matches_reference_golden(cuda):
    """Directly against vectors recorded from the reference modules (tests/golden/resnet.npz)."""
    from ocl_amd.loss import cross_entropy_mean
    g = gold("resnet")
    for name, agent, data, hw, n, head in [("rr18_c100", "ER", "cifar100", 32, 6, None), ("scr_mlp", "SCR", "cifar100", 32, 6, "mlp"),
                                           ("rr18_mini", "ER", "mini_imagenet", 84, 3, None)]:
        m, sd = build(agent, data, head or "mlp", seed=11, cuda=cuda, max_batch=16)
        rng = np.random.default_rng(5)
        x = torch.from_numpy(rng.random((n, 3, hw, hw)).astype(np.float32)).to(cuda)
        y = torch.from_numpy(rng.integers(0, 100, n).astype(np.int64)).to(cuda)
        m.train()
        o = m.forward(x)
        loss = cross_entropy_mean(o, y) if head is None else (o * torch.linspace(-1, 1, o.numel()).view_as(o).to(cuda)).sum()
        loss.backward()
        assert np.abs(o.detach().cpu().numpy() - g[name + "_out"]).max() < 1e-4
        assert abs(float(loss) - float(g[name + "_loss"])) < 1e-4 * max(1.0, abs(float(g[name + "_loss"])))
        named = dict(m.named_parameters())
        for k in g[name + "_picked"]:
            gg = g[name + "_g_" + str(k)]
            got = named[str(k)].grad.cpu().numpy()
            assert np.linalg.norm(got - gg) <= 5e-2 * np.linalg.norm(gg), (name, k)   # loose: ReLU flips; tight check is teacher-forced above
        m.eval()
        with torch.no_grad():
            fe = m.features(x).cpu().numpy()
        assert relmax(fe, g[name + "_feat_eval"]) < 1e-4


def test_backward_without_tape_fails_loudly(cuda):
    from ocl_amd import ffi
    m, _ = build("ER", "cifar10", cuda=cuda)
    m.n_slots = 2
    m.train()
    x = torch.rand(4, 3, 32, 32, device=cuda)
    o1 = m.forward(x)
    m.forward(x)
    m.forward(x)    # third forward overwrites the first one's tape (2 slots)
    with pytest.raises(RuntimeError):
        o1.sum().backward()
    with pytest.raises(RuntimeError):
        m.forward(torch.rand(4, 3, 16, 16, device=cuda))
    with pytest.raises(RuntimeError):
        m.forward(torch.rand(4, 3, 32, 32))     # CPU tensor: no CPU path


@pytest.mark.parametrize("n", [1, 2, 3, 129, 300])
def test_edge_batch_sizes_train_and_eval_forward(cuda, n):
    """Batch sizes the tilings do not divide: a single image, odd counts, more than one 128-pixel tile's worth of 4x4
    images, and a batch above max_batch (eval features are chunked; a train forward above max_batch must fail loudly)."""
    m, sd = build("ER", "cifar100", cuda=cuda, max_batch=160)
    rng = np.random.default_rng(100 + n)
    x = rng.random((n, 3, 32, 32)).astype(np.float32)
    xd = torch.from_numpy(x).to(cuda)
    m.eval()
    with torch.no_grad():
        f = m.features_batched(xd).cpu().numpy()
    net = O.OracleNet(O.clone_state(sd, requires_grad=False), head=None, training=False)
    with torch.no_grad():
        f_ref = net.features(torch.from_numpy(x)).numpy()
    assert f.shape == f_ref.shape and relmax(f, f_ref) < 1e-4
    m.train()
    if n > 160:
        with pytest.raises(RuntimeError):
            m.forward(xd)
        return
    out = m.forward(xd)
    net_t = O.OracleNet(O.clone_state(sd, requires_grad=False), head=None, training=True)
    with torch.no_grad():
        out_ref = net_t.forward(torch.from_numpy(x)).numpy()
    if n > 1:      # n = 1: BatchNorm of a 1x1 feature map over one sample is degenerate at layer 4 (var of 16 values only)
        assert np.abs(out.detach().cpu().numpy() - out_ref).max() < 2e-3 * (1 + np.abs(out_ref).max())
    out.sum().backward()
    g = m.flat_grads().cpu().numpy()
    assert np.isfinite(g).all() and np.abs(g).max() > 0


