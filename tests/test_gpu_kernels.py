"""GPU: every small kernel of the C-ABI against the reference-generated golden vectors and the oracle."""
import numpy as np
import pytest
import torch

from conftest import gold
from oracle import ocl_oracle as O

pytestmark = pytest.mark.gpu


def dev(a, cuda):
    return torch.from_numpy(np.ascontiguousarray(a)).to(cuda)


def test_gemm_small_mfma_layout_is_not_transposed(cuda):
    """A=I-style check with an ASYMMETRIC B (a symmetric B would hide a row/col swap of the MFMA C/D layout)."""
    from ocl_amd import ops
    rng = np.random.default_rng(0)
    for (m, n, k) in [(16, 16, 4), (16, 16, 16), (37, 53, 29), (220, 160, 160), (10, 100, 640), (128, 5, 3)]:
        a = rng.standard_normal((m, k)).astype(np.float32)
        b = rng.standard_normal((k, n)).astype(np.float32)
        bias = rng.standard_normal(n).astype(np.float32)
        c = ops.gemm_small(dev(a, cuda), dev(b, cuda), bias=dev(bias, cuda)).cpu().numpy()
        ref = a.astype(np.float64) @ b.astype(np.float64) + bias
        assert np.abs(c - ref).max() <= 2e-5 * (1 + np.abs(ref).max()), (m, n, k)
        ct = ops.gemm_small(dev(a, cuda), dev(np.ascontiguousarray(b.T), cuda), relu=True, trans_b=True).cpu().numpy()
        assert np.abs(ct - np.maximum(a.astype(np.float64) @ b.astype(np.float64), 0)).max() <= 2e-5 * (1 + np.abs(ref).max())


def test_knn_sv_bit_exact_vs_reference_golden(cuda):
    from ocl_amd import ops
    g = gold("knn_sv")
    for ci in range(int(g["n_cases"])):
        sv, order = ops.knn_sv(dev(g["c%d_ef" % ci], cuda), dev(g["c%d_ey" % ci], cuda), dev(g["c%d_cf" % ci], cuda),
                               dev(g["c%d_cy" % ci], cuda), int(g["c%d_k" % ci]), want_order=True)
        sv, order = sv.cpu().numpy(), order.cpu().numpy()
        gsv, gorder = g["c%d_sv" % ci], g["c%d_order" % ci]
        if not np.array_equal(order, gorder):
            # a rank flip is only legitimate between candidates whose fp32 distances are within round-off of each other
            d = O.sq_dist_matrix(g["c%d_ef" % ci], g["c%d_cf" % ci])
            for r, c in zip(*np.nonzero(order != gorder)):
                a, b = d[r, order[r, c]], d[r, gorder[r, c]]
                assert abs(a - b) <= 4e-6 * max(a, b), "case %d: rank differs beyond a near-tie" % ci
            gsv, _ = O.knn_sv(g["c%d_ef" % ci], g["c%d_ey" % ci], g["c%d_cf" % ci], g["c%d_cy" % ci], int(g["c%d_k" % ci]), order=order)
        assert np.array_equal(sv, gsv), "case %d: max diff %g" % (ci, np.abs(sv - gsv).max())
    sv = ops.knn_sv(dev(g["tie_ef"], cuda), dev(g["tie_ey"], cuda), dev(g["tie_cf"], cuda), dev(g["tie_cy"], cuda), int(g["tie_k"]))
    assert np.array_equal(sv.cpu().numpy(), g["tie_sv"])


def test_knn_sv_edge_sizes(cuda):
    from ocl_amd import ops
    rng = np.random.default_rng(3)
    for (ne, nc, d, k) in [(1, 1, 8, 3), (3, 2, 8, 5), (2, 1000, 32, 3), (4, 2048, 16, 7), (7, 129, 160, 3)]:
        ef, cf = rng.standard_normal((ne, d)).astype(np.float32), rng.standard_normal((nc, d)).astype(np.float32)
        ey, cy = rng.integers(0, 5, ne).astype(np.int64), rng.integers(0, 5, nc).astype(np.int64)
        sv, order = ops.knn_sv(dev(ef, cuda), dev(ey, cuda), dev(cf, cuda), dev(cy, cuda), k, want_order=True)
        exp, _ = O.knn_sv(ef, ey, cf, cy, k, order=order.cpu().numpy())
        assert np.array_equal(sv.cpu().numpy(), exp)
        dist = O.sq_dist_matrix(ef, cf)
        srt = np.take_along_axis(dist, order.cpu().numpy(), 1)
        assert (np.diff(srt, axis=1) >= -4e-6 * (1 + srt[:, 1:])).all(), "candidate order is not ascending in distance"
        # efficiency property of the Shapley value: the values of one evaluation point sum to v(all) = (#matches among the k nearest)/k
        if nc >= k:
            tot = sv.cpu().numpy().sum(1)
            near = np.take_along_axis(np.broadcast_to(cy, (ne, nc)), order.cpu().numpy()[:, :k], 1) == ey[:, None]
            assert np.abs(tot - near.sum(1) / k).max() < 1e-4
    assert ops.knn_sv(torch.zeros(0, 8, device=cuda), torch.zeros(0, dtype=torch.long, device=cuda), torch.zeros(5, 8, device=cuda),
                      torch.zeros(5, dtype=torch.long, device=cuda), 3).shape == (0, 5)
    with pytest.raises(RuntimeError):
        ops.knn_sv(torch.zeros(1, 8, device=cuda), torch.zeros(1, dtype=torch.long, device=cuda), torch.zeros(4096, 8, device=cuda),
                   torch.zeros(4096, dtype=torch.long, device=cuda), 3)


def test_aser_score_colreduce_argsort(cuda):
    from ocl_amd import ops
    rng = np.random.default_rng(4)
    adv, coop = rng.standard_normal((10, 100)).astype(np.float32), rng.standard_normal((97, 100)).astype(np.float32)
    for t in ("asvm", "asv", "neg_sv"):
        got = ops.aser_score(dev(adv, cuda), dev(coop, cuda), t).cpu().numpy()
        assert np.abs(got - O.aser_score(adv, coop, t)).max() < 1e-6
    for mode, fn in (("sum", np.sum), ("mean", np.mean), ("max", np.max), ("min", np.min)):
        assert np.abs(ops.col_reduce(dev(coop, cuda), mode).cpu().numpy() - fn(coop.astype(np.float64), 0)).max() < 1e-5
    for n in (1, 2, 10, 160, 257, 4096):
        v = rng.integers(0, 6, n).astype(np.float32)          # many exact ties
        idx = ops.argsort_desc(dev(v, cuda)).cpu().numpy()
        assert np.array_equal(idx, O.argsort_desc_stable(v)), n   # descending, ties in ascending index order
    v = rng.standard_normal(1000).astype(np.float32)
    assert np.array_equal(ops.argsort_desc(dev(v, cuda)).cpu().numpy(), np.argsort(-v, kind="stable"))


def test_supcon_matches_reference_golden(cuda):
    """loss / gradient within 1e-5 abs of the reference's SupConLoss + autograd (tolerance: fp32 exp/log round-off)."""
    from ocl_amd import ops
    from ocl_amd.loss import SupConLoss
    g = gold("supcon")
    for ci in range(int(g["n_cases"])):
        f, y, t = g["c%d_f" % ci], g["c%d_y" % ci], float(g["c%d_t" % ci])
        b = f.shape[0]
        vm = np.concatenate([f[:, 0], f[:, 1]], 0)
        loss, df = ops.supcon(dev(vm, cuda), dev(y, cuda), 2, t)
        assert abs(float(loss) - float(g["c%d_loss" % ci])) < 1e-5
        gg = g["c%d_grad" % ci]
        df = df.cpu().numpy()
        assert np.abs(df[:b] - gg[:, 0]).max() < 1e-5 and np.abs(df[b:] - gg[:, 1]).max() < 1e-5
        # the nn.Module mirror with the reference's [bsz, n_views, dim] layout and autograd
        ft = dev(f, cuda).requires_grad_(True)
        l2 = SupConLoss(temperature=t)(ft, dev(y, cuda))
        l2.backward()
        assert abs(float(l2) - float(g["c%d_loss" % ci])) < 1e-5 and np.abs(ft.grad.cpu().numpy() - gg).max() < 1e-5
    with pytest.raises(ValueError):
        SupConLoss()(torch.zeros(4, 8, device=cuda), torch.zeros(4, dtype=torch.long, device=cuda))
    with pytest.raises(ValueError):
        SupConLoss()(torch.zeros(4, 2, 8, device=cuda), torch.zeros(3, dtype=torch.long, device=cuda))


def test_cross_entropy_and_mir_scores(cuda):
    from ocl_amd import ops
    rng = np.random.default_rng(6)
    for (n, c) in [(10, 10), (50, 100), (20, 100), (1, 2), (110, 100), (7, 130)]:
        lg = (rng.standard_normal((n, c)) * 3).astype(np.float32)
        y = rng.integers(0, c, n).astype(np.int64)
        t = torch.from_numpy(lg).requires_grad_(True)
        ref = torch.nn.functional.cross_entropy(t, torch.from_numpy(y))
        ref.backward()
        loss, dl = ops.cross_entropy(dev(lg, cuda), dev(y, cuda), "mean")
        assert abs(float(loss) - float(ref.detach())) < 1e-5 and np.abs(dl.cpu().numpy() - t.grad.numpy()).max() < 1e-6
        per, _ = ops.cross_entropy(dev(lg, cuda), dev(y, cuda), "none", want_grad=False)
        refn = torch.nn.functional.cross_entropy(torch.from_numpy(lg), torch.from_numpy(y), reduction="none").numpy()
        assert np.abs(per.cpu().numpy() - refn).max() < 1e-5
        lg2 = (lg + rng.standard_normal((n, c)).astype(np.float32))
        sc = ops.mir_scores(dev(lg, cuda), dev(lg2, cuda), dev(y, cuda)).cpu().numpy()
        assert np.abs(sc - O.mir_scores(torch.from_numpy(lg), torch.from_numpy(lg2), torch.from_numpy(y)).numpy()).max() < 2e-5


def test_gather_scatter_rows_and_u8_images(cuda):
    from ocl_amd import ops
    rng = np.random.default_rng(8)
    for shape in [(50, 3, 32, 32), (20, 3, 84, 84), (40, 7), (33,)]:
        src = rng.standard_normal(shape).astype(np.float32)
        idx = rng.integers(0, shape[0], 17).astype(np.int64)
        assert np.array_equal(ops.gather_rows(dev(src, cuda), dev(idx, cuda)).cpu().numpy(), src[idx])
        dst = dev(np.zeros(shape, np.float32), cuda)
        uniq = rng.permutation(shape[0])[:9].astype(np.int64)
        rows = rng.standard_normal((9,) + shape[1:]).astype(np.float32)
        ops.scatter_rows(dst, dev(uniq, cuda), dev(rows, cuda))
        exp = np.zeros(shape, np.float32)
        exp[uniq] = rows
        assert np.array_equal(dst.cpu().numpy(), exp)
    lab = rng.integers(0, 100, 64).astype(np.int64)
    idx = rng.integers(0, 64, 10).astype(np.int64)
    assert np.array_equal(ops.gather_rows(dev(lab, cuda), dev(idx, cuda)).cpu().numpy(), lab[idx])
    assert ops.gather_rows(dev(lab, cuda), torch.zeros(0, dtype=torch.long, device=cuda)).numel() == 0   # empty retrieve (first iteration)
    for hw in (32, 84):
        u8 = rng.integers(0, 256, (12, hw, hw, 3), dtype=np.uint8)
        idx = rng.integers(0, 12, 5).astype(np.int64)
        got = ops.gather_u8_images(dev(u8, cuda), dev(idx, cuda)).cpu()
        assert torch.equal(got, O.to_tensor(u8[idx]))     # bit-exact ToTensor: u8 -> float / 255


def test_sgd_step_and_virtual_step(cuda):
    from ocl_amd import ops
    rng = np.random.default_rng(9)
    for n in (1155608, 1003, 4):
        p, g = rng.standard_normal(n).astype(np.float32), rng.standard_normal(n).astype(np.float32)
        pt = torch.from_numpy(p.copy()).requires_grad_(True)
        pt.grad = torch.from_numpy(g.copy())
        torch.optim.SGD([pt], lr=0.1, weight_decay=0).step()
        pd, gd = dev(p, cuda), dev(g, cuda)
        shadow = torch.empty_like(pd)
        ops.sgd_step(pd, gd, 0.1, out=shadow)
        assert torch.equal(pd.cpu(), torch.from_numpy(p)), "virtual step must not touch the parameters"
        assert np.abs(shadow.cpu().numpy() - (p - np.float32(0.1) * g)).max() <= 2.4e-7 * (1 + np.abs(p).max())   # 1 ulp: fma vs mul+sub
        ops.sgd_step(pd, gd, 0.1)
        assert np.abs(pd.cpu().numpy() - pt.detach().numpy()).max() <= 2.4e-7 * (1 + np.abs(p).max())


def test_ncm_means_and_predict(cuda):
    from ocl_amd import ops
    rng = np.random.default_rng(10)
    n, d = 500, 160
    feats = rng.standard_normal((n, d)).astype(np.float32) + 2.0
    labels = rng.integers(0, 12, n).astype(np.int64)
    class_ids = [5, 3, 11, 0, 42, 7]          # 42 has no exemplar
    means, counts = ops.ncm_class_means(dev(feats, cuda), dev(labels, cuda), torch.tensor(class_ids, device=cuda))
    exp = O.ncm_means(torch.from_numpy(feats), torch.from_numpy(labels), class_ids).numpy()
    counts = counts.cpu().numpy()
    assert counts.tolist() == [int((labels == c).sum()) for c in class_ids]
    ok = counts > 0
    assert np.abs(means.cpu().numpy()[ok] - exp[ok]).max() < 1e-6
    m = exp.copy()
    m[~ok] = rng.standard_normal((int((~ok).sum()), d)).astype(np.float32)
    test = rng.standard_normal((128, d)).astype(np.float32) + 2.0
    pred = ops.ncm_predict(dev(test, cuda), dev(m, cuda)).cpu().numpy()
    assert np.array_equal(pred, O.ncm_predict(torch.from_numpy(test), torch.from_numpy(m)).numpy())


def test_scr_augment_kernel_properties(cuda):
    """kornia parity is unpinned; the kernel is checked against its own contract."""
    from ocl_amd import ops
    rng = np.random.default_rng(11)
    x = rng.random((6, 3, 32, 32)).astype(np.float32)
    ident = np.tile(np.array([0, 0, 32, 32, 0, 0, 1, 1, 1, 0, 0, 0], np.float32), (6, 1))
    out = ops.scr_augment(dev(x, cuda), dev(ident, cuda)).cpu().numpy()
    assert np.abs(out - x).max() < 1e-6                                 # full crop, no flip / jitter / gray = identity
    flip = ident.copy(); flip[:, 4] = 1
    assert np.abs(ops.scr_augment(dev(x, cuda), dev(flip, cuda)).cpu().numpy() - x[:, :, :, ::-1]).max() < 1e-6
    gray = ident.copy(); gray[:, 11] = 1
    og = ops.scr_augment(dev(x, cuda), dev(gray, cuda)).cpu().numpy()
    lum = 0.299 * x[:, 0] + 0.587 * x[:, 1] + 0.114 * x[:, 2]
    assert np.abs(og - lum[:, None]).max() < 1e-6
    jit = ident.copy(); jit[:, 5] = 1; jit[:, 6] = 1.3; jit[:, 7] = 0.8; jit[:, 10] = 0   # brightness +0.3 then contrast x0.8
    oj = ops.scr_augment(dev(x, cuda), dev(jit, cuda)).cpu().numpy()
    assert np.abs(oj - np.clip(np.clip(x + 0.3, 0, 1) * 0.8, 0, 1)).max() < 1e-5
    hue = ident.copy(); hue[:, 5] = 1; hue[:, 9] = 1.0; hue[:, 10] = 23                    # a full turn of hue = identity
    assert np.abs(ops.scr_augment(dev(x, cuda), dev(hue, cuda)).cpu().numpy() - x).max() < 1e-4
    crop = ident.copy(); crop[:, 0] = 8; crop[:, 1] = 4; crop[:, 2] = 16; crop[:, 3] = 16   # 2x upsample of a 16x16 window
    oc = ops.scr_augment(dev(x, cuda), dev(crop, cuda)).cpu().numpy()
    ref = torch.nn.functional.interpolate(torch.from_numpy(x[:, :, 8:24, 4:20]), size=(32, 32), mode="bilinear", align_corners=False).numpy()
    assert np.abs(oc[:, :, 2:-2, 2:-2] - ref[:, :, 2:-2, 2:-2]).max() < 1e-5    # interior (edges clamp to the full image, not the crop)
    assert oc.min() >= 0 and oc.max() <= 1


def test_upload_ring_keeps_every_payload(cuda):
    """ocl_upload: small host arrays through the library's ring of pinned staging slots (128 slots of 64 KB), larger ones by a
    plain copy.  400 uploads of changing sizes (the ring wraps three times; the host array is overwritten right after each call,
    which the contract allows) must all arrive intact, on the stream they were issued on."""
    from ocl_amd import ops
    rng = np.random.default_rng(21)
    kept = []
    for i in range(400):
        n = int(rng.choice([1, 7, 10, 100, 1000, 16384, 20000]))        # 20000 int64 = 160 KB: past the slot size
        host = torch.from_numpy(rng.integers(-2**40, 2**40, n))
        want = host.clone()
        dev_t = ops.upload(host, cuda)
        host.zero_()                                                     # the caller may reuse its array immediately
        kept.append((dev_t, want))
    torch.cuda.synchronize()
    for dev_t, want in kept:
        assert dev_t.is_cuda and torch.equal(dev_t.cpu(), want)
    f = ops.upload(np.arange(12, dtype=np.float32).reshape(3, 4), cuda)
    assert f.shape == (3, 4) and f.dtype == torch.float32 and torch.equal(f.cpu(), torch.arange(12.).reshape(3, 4))
    assert ops.upload(torch.zeros(0, dtype=torch.long), cuda).numel() == 0


def test_scr_augment_parameters_on_the_device_match_the_host_statement(cuda):
    """aug_params_kernel (crop attempts, fallback, position, jitter factors from raw uniform draws) against
    ScrAugment.params_from_uniform on the same draws; the fused call equals the two-step call on its own parameters."""
    from ocl_amd import ops
    from ocl_amd.agents.scr import ScrAugment
    torch.manual_seed(5)
    for size, scale, n in (((32, 32), (0.2, 1.0), 4096), ((84, 84), (0.2, 1.0), 2048), ((16, 64), (0.99, 1.0), 512)):
        aug = ScrAugment(size, scale=scale)
        u = aug.draw(n)
        want = aug.params_from_uniform(u).numpy()
        x = torch.rand(n, 3, size[0], size[1])
        out, params = ops.scr_augment_uniform(dev(x, cuda), dev(u, cuda), aug.config(), want_params=True)
        got = params.cpu().numpy()
        # exp / sqrt differ by an ulp between the two maths libraries: a crop edge may round the other way for a draw that lands on
        # a rounding boundary (and then the position, drawn inside the crop's slack, moves with it): rare, and never by more than 1
        differs = np.abs(got - want).max(1) > 1e-5
        assert differs.mean() < 2e-3, differs.mean()
        assert np.abs(got[:, 2:4] - want[:, 2:4]).max() <= 1.0
        assert np.array_equal(got[:, [4, 5, 10, 11]], want[:, [4, 5, 10, 11]])
        assert np.abs(got[:, 6:10] - want[:, 6:10]).max() < 1e-6
        if size == (16, 64):
            assert (got[:, 2] == 16).all() and (got[:, 3] == 21).all()          # fallback crop: ratio clamped to 4/3
        two_step = ops.scr_augment(dev(x, cuda), params)
        assert torch.equal(out, two_step)


def test_ce_tricks_match_reference_golden(cuda):
    """ocl_ce_segmented_fwd_bwd through the agent's criterion (host-built segment table) vs the reference's labels trick /
    separated softmax + autograd: loss and d(loss)/d(logits) within 1e-5 abs (fp32 exp/log round-off); a label outside
    old + new classes raises KeyError as in the reference."""
    from types import SimpleNamespace
    from ocl_amd.agents.base import ContinualLearner
    g = gold("ce_tricks")
    for ci in range(int(g["n_cases"])):
        kind = str(g["c%d_kind" % ci])
        old, new = g["c%d_old" % ci].tolist(), g["c%d_new" % ci].tolist()
        trick = {k: False for k in ('labels_trick', 'kd_trick', 'separated_softmax', 'review_trick', 'ncm_trick', 'kd_trick_star')}
        trick['labels_trick' if kind == "labels" else 'separated_softmax'] = True
        fake = SimpleNamespace(params=SimpleNamespace(trick=trick, agent="ER", temp=0.07), old_labels=old, new_labels=new,
                               lbl_inv_map={l: i for i, l in enumerate(old + new)})
        fake._host_labels = lambda labels: ContinualLearner._host_labels(fake, labels)
        for with_host in (False, True):
            lt = dev(g["c%d_logits" % ci], cuda).requires_grad_(True)
            y = dev(g["c%d_y" % ci], cuda)
            if with_host:
                y.host = g["c%d_y" % ci]
            loss = ContinualLearner.criterion(fake, lt, y)
            loss.backward()
            assert abs(float(loss) - float(g["c%d_loss" % ci])) < 1e-5
            assert np.abs(lt.grad.cpu().numpy() - g["c%d_grad" % ci]).max() < 1e-5
        if kind == "sep":
            bad = dev(np.array([99], dtype=np.int64), cuda)
            with pytest.raises(KeyError):
                ContinualLearner.criterion(fake, dev(g["c%d_logits" % ci][:1], cuda), bad)


def test_kd_loss_matches_reference_golden(cuda):
    """ocl_kd_fwd_bwd + its autograd node vs the reference's loss_fn_kd + autograd (1e-5 abs)."""
    from ocl_amd.loss import loss_fn_kd
    g = gold("kd")
    for ci in range(int(g["n_cases"])):
        st = dev(g["c%d_s" % ci], cuda).requires_grad_(True)
        loss = loss_fn_kd(st, dev(g["c%d_t" % ci], cuda), float(g["c%d_T" % ci]))
        (0.25 * loss).backward()
        assert abs(float(loss) - float(g["c%d_loss" % ci])) < 1e-5
        assert np.abs(st.grad.cpu().numpy() - 0.25 * g["c%d_grad" % ci]).max() < 1e-5
