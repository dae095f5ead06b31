"""Tensor-level wrappers over the C-ABI (one function per reference op sequence).

All inputs must live on the MI355X; outputs are allocated with torch (caller-owned memory is the
C-ABI's convention).  Nothing here computes on the CPU.
"""
import torch

from . import ffi


_data_streams = {}


def data_stream(device):
    """The stream of the data path (loader gather, buffer retrieve / update, concatenation, augmentation): none of it depends on
    the weights, so an agent can issue step i+1's data work next to step i's backward (agents/scr.py)."""
    key = torch.device(device).index
    if key not in _data_streams:
        _data_streams[key] = torch.cuda.Stream(device=device)
    return _data_streams[key]


def upload(t, device):
    """Asynchronous upload of a small CPU tensor (or numpy array) to `device` on the current stream.  A pageable-memory
    `.to(device)` blocks the host until the stream has drained, which serialises the host bookkeeping of step i+1 behind the GPU
    work of step i; `ocl_upload` stages the payload through the library's ring of pinned slots instead (one C call, no torch
    event / pinned-tensor objects: the index vectors of a replay step are 4 such uploads)."""
    if not torch.is_tensor(t):
        t = torch.from_numpy(t)
    if t.is_cuda:
        return t
    t = t.contiguous()
    d = torch.empty(t.shape, dtype=t.dtype, device=device)
    n = t.numel() * t.element_size()
    if n:
        if d.device.index != torch._C._cuda_getDevice():     # the staging ring and the stream belong to the current device
            with torch.cuda.device(d.device):
                return upload(t, device)
        ffi.init()
        ffi.check(ffi.lib().ocl_upload(ffi.vp(t.data_ptr()), n, ffi.vp(d.data_ptr()), ffi.stream()), "upload")
    return d


def _f32(t):
    if t.dtype != torch.float32 or not t.is_cuda:
        raise RuntimeError("expected a float32 tensor on the GPU, got %s on %s" % (t.dtype, t.device))
    return t.contiguous()


def _i64(t):
    if t.dtype != torch.int64 or not t.is_cuda:
        raise RuntimeError("expected an int64 tensor on the GPU, got %s on %s" % (t.dtype, t.device))
    return t.contiguous()


# ---- K9 ----------------------------------------------------------------------------------------------
def gather_rows(src, idx, out=None):
    """src[idx] for a contiguous [R, ...] tensor (utils/buffer/buffer_utils.py:19-21)."""
    ffi.init()
    idx = _i64(idx)
    src = src.contiguous()
    n = idx.numel()
    row_bytes = src[0].numel() * src.element_size() if src.shape[0] > 0 else 0
    if out is None:
        out = torch.empty((n,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    if n == 0:
        return out
    if row_bytes % 4 != 0:
        raise RuntimeError("gather_rows: row size must be a multiple of 4 bytes")
    ffi.check(ffi.lib().ocl_gather_rows(ffi.ptr(src), ffi.ptr(idx), n, row_bytes, ffi.ptr(out), ffi.stream()), "gather_rows")
    return out


def gather_pair(src_a, src_b, idx):
    """(src_a[idx], src_b[idx]) -- replay-buffer images and labels by one index vector -- as ONE call and one launch.  idx: an int64
    tensor on the host (the usual case: indices drawn by the numpy / torch-CPU generators; uploaded inside the call) or on the device."""
    ffi.init()
    n = idx.numel()
    dev = src_a.device
    out_a = torch.empty((n,) + tuple(src_a.shape[1:]), dtype=src_a.dtype, device=dev)
    out_b = torch.empty((n,) + tuple(src_b.shape[1:]), dtype=src_b.dtype, device=dev)
    if n == 0:
        return out_a, out_b
    if not (src_a.is_contiguous() and src_b.is_contiguous()):
        raise RuntimeError("gather_pair: sources must be contiguous")
    rb_a = src_a[0].numel() * src_a.element_size()
    rb_b = src_b[0].numel() * src_b.element_size()
    if idx.is_cuda:
        idx = _i64(idx)
        host_ptr, idx_dev = ffi.vp(0), idx
    else:
        idx = idx.contiguous()
        if idx.dtype != torch.int64:
            raise RuntimeError("gather_pair: int64 indices")
        host_ptr, idx_dev = ffi.vp(idx.data_ptr()), torch.empty(n, dtype=torch.int64, device=dev)
    if dev.index != torch._C._cuda_getDevice():
        with torch.cuda.device(dev):
            return gather_pair(src_a, src_b, idx)
    ffi.check(ffi.lib().ocl_gather_rows_pair(ffi.ptr(src_a), rb_a, ffi.ptr(out_a), ffi.ptr(src_b), rb_b, ffi.ptr(out_b), host_ptr,
                                             ffi.ptr(idx_dev), n, ffi.stream()), "gather_rows_pair")
    return out_a, out_b


def scatter_rows(dst, idx, src):
    """dst[idx] = src (utils/buffer/reservoir_update.py:59-60)."""
    ffi.init()
    idx = _i64(idx)
    n = idx.numel()
    if n == 0:
        return dst
    src = src.contiguous()
    if not dst.is_contiguous():
        raise RuntimeError("scatter_rows: destination must be contiguous")
    row_bytes = dst[0].numel() * dst.element_size()
    if src.numel() * src.element_size() != n * row_bytes:
        raise RuntimeError("scatter_rows: source/destination row size mismatch")
    ffi.check(ffi.lib().ocl_scatter_rows(ffi.ptr(dst), ffi.ptr(idx), n, row_bytes, ffi.ptr(src), ffi.stream()), "scatter_rows")
    return dst


def gather_u8_images(task_u8_nhwc, idx):
    """ToTensor() of task_u8_nhwc[idx]: uint8 [N,H,W,C] -> float32 [n,C,H,W] / 255."""
    ffi.init()
    idx = _i64(idx)
    n = idx.numel()
    _, h, w, c = task_u8_nhwc.shape
    out = torch.empty((n, c, h, w), dtype=torch.float32, device=task_u8_nhwc.device)
    if n:
        ffi.check(ffi.lib().ocl_gather_u8_hwc_to_f32_chw(ffi.ptr(task_u8_nhwc), ffi.ptr(idx), n, h, w, c, ffi.ptr(out),
                                                         ffi.stream()), "gather_u8")
    return out


# ---- K8 ----------------------------------------------------------------------------------------------
def sgd_step(params_flat, grads_flat, lr, weight_decay=0.0, grad_scale=1.0, out=None):
    ffi.init()
    ffi.check(ffi.lib().ocl_sgd_step(ffi.ptr(params_flat), ffi.ptr(grads_flat), params_flat.numel(), float(lr),
                                     float(weight_decay), float(grad_scale), ffi.ptr(out), ffi.stream()), "sgd_step")
    return out if out is not None else params_flat


# ---- K6 ----------------------------------------------------------------------------------------------
def cross_entropy(logits, y, reduction="mean", want_grad=True, dl_out=None):
    """(loss, dlogits): torch.nn.CrossEntropyLoss / F.cross_entropy(reduction='none').  dl_out: a contiguous float32 [n, c] tensor
    (e.g. a row block of a larger gradient buffer) that receives dlogits instead of a fresh one."""
    ffi.init()
    logits = _f32(logits)
    y = _i64(y)
    n, c = logits.shape
    red = {"none": 0, "mean": 1}[reduction]
    loss = torch.empty(n if red == 0 else 1, dtype=torch.float32, device=logits.device)
    if dl_out is not None:
        if dl_out.shape != logits.shape or dl_out.dtype != torch.float32 or not dl_out.is_contiguous() or dl_out.device != logits.device:
            raise RuntimeError("cross_entropy: dl_out must be a contiguous float32 %s tensor on %s" % (tuple(logits.shape), logits.device))
        dl = dl_out
    else:
        dl = torch.empty_like(logits) if want_grad else None
    ffi.check(ffi.lib().ocl_ce_fwd_bwd(ffi.ptr(logits), ffi.ptr(y), n, c, red, ffi.ptr(loss), ffi.ptr(dl), ffi.stream()), "ce")
    return (loss if red == 0 else loss[0]), dl


def cross_entropy_segmented(logits, y, seg, want_grad=True):
    """(mean loss, dlogits) of the softmax over each row's label segment; seg: int32 [c] on the device (agents/base.py:96-108)."""
    ffi.init()
    logits = _f32(logits)
    y = _i64(y)
    n, c = logits.shape
    if seg.dtype != torch.int32 or not seg.is_cuda or seg.numel() != c:
        raise RuntimeError("segment ids must be an int32 [%d] tensor on the GPU" % c)
    loss = torch.empty(1, dtype=torch.float32, device=logits.device)
    dl = torch.empty_like(logits) if want_grad else None
    ffi.check(ffi.lib().ocl_ce_segmented_fwd_bwd(ffi.ptr(logits), ffi.ptr(y), ffi.ptr(seg.contiguous()), n, c, ffi.ptr(loss), ffi.ptr(dl),
                                                 ffi.stream()), "ce_segmented")
    return loss[0], dl


def kd_loss(scores, target_scores, T=2.0, want_grad=True):
    """(loss, dscores) of utils/kd_manager.py:6-11 (loss_fn_kd)."""
    ffi.init()
    scores, target_scores = _f32(scores), _f32(target_scores)
    if scores.shape != target_scores.shape or scores.dim() != 2:
        raise RuntimeError("kd_loss: scores %s vs target %s" % (tuple(scores.shape), tuple(target_scores.shape)))
    n, c = scores.shape
    loss = torch.empty(1, dtype=torch.float32, device=scores.device)
    ds = torch.empty_like(scores) if want_grad else None
    ffi.check(ffi.lib().ocl_kd_fwd_bwd(ffi.ptr(scores), ffi.ptr(target_scores), n, c, float(T), ffi.ptr(loss), ffi.ptr(ds), ffi.stream()), "kd")
    return loss[0], ds


# ---- K7 ----------------------------------------------------------------------------------------------
def supcon(feat_view_major, y, n_views, temperature, want_grad=True):
    """(loss, dfeat) for view-major features [n_views*bsz, dim] (utils/loss.py:19-96)."""
    ffi.init()
    feat = _f32(feat_view_major)
    y = _i64(y)
    a, dim = feat.shape
    bsz = y.numel()
    if a != bsz * n_views:
        raise ValueError("Num of labels does not match num of features")
    ws = torch.empty(ffi.lib().ocl_supcon_workspace_bytes(a), dtype=torch.uint8, device=feat.device)
    loss = torch.empty(1, dtype=torch.float32, device=feat.device)
    df = torch.empty_like(feat) if want_grad else None
    ffi.check(ffi.lib().ocl_supcon_fwd_bwd(ffi.ptr(feat), ffi.ptr(y), bsz, n_views, dim, float(temperature), ffi.ptr(loss),
                                           ffi.ptr(df), ffi.ptr(ws), ffi.stream()), "supcon")
    return loss[0], df


# ---- K10 ---------------------------------------------------------------------------------------------
def knn_sv(eval_f, eval_y, cand_f, cand_y, k, want_order=False):
    """sv_matrix [n_eval, n_cand] (utils/buffer/aser_utils.py:7-61) from deep features."""
    ffi.init()
    eval_f, cand_f = _f32(eval_f), _f32(cand_f)
    eval_y, cand_y = _i64(eval_y), _i64(cand_y)
    ne, d = eval_f.shape
    nc = cand_f.shape[0]
    # (zeros, not empty: with a NaN distance -- a diverged feature -- the sort may move a padding entry into a row and leave a cell unwritten;
    # a Shapley value of 0 there, not whatever the allocator held)
    sv = torch.zeros((ne, nc), dtype=torch.float32, device=eval_f.device)
    order = torch.empty((ne, nc), dtype=torch.int64, device=eval_f.device) if want_order else None
    if ne and nc:
        ffi.check(ffi.lib().ocl_knn_sv(ffi.ptr(eval_f), ffi.ptr(eval_y), ne, ffi.ptr(cand_f), ffi.ptr(cand_y), nc, d, int(k),
                                       ffi.ptr(sv), ffi.ptr(order), ffi.stream()), "knn_sv")
    return (sv, order) if want_order else sv


def col_reduce(m, mode):
    ffi.init()
    m = _f32(m)
    r, c = m.shape
    out = torch.empty(c, dtype=torch.float32, device=m.device)
    ffi.check(ffi.lib().ocl_col_reduce(ffi.ptr(m), r, c, {"sum": 0, "mean": 1, "max": 2, "min": 3}[mode], ffi.ptr(out),
                                       ffi.stream()), "col_reduce")
    return out


def aser_score(sv_adv, sv_coop, aser_type):
    ffi.init()
    sv_adv = _f32(sv_adv)
    t = {"asvm": 0, "asv": 1, "neg_sv": 2}.get(aser_type, 0)  # "asvm or anything else" (aser_retrieve.py:80-82)
    n_adv, n_cand = sv_adv.shape
    out = torch.empty(n_cand, dtype=torch.float32, device=sv_adv.device)
    if t != 2:
        sv_coop = _f32(sv_coop)
        n_coop = sv_coop.shape[0]
    else:
        sv_coop, n_coop = None, 0
    ffi.check(ffi.lib().ocl_aser_score(ffi.ptr(sv_adv), n_adv, ffi.ptr(sv_coop), n_coop, n_cand, t, ffi.ptr(out), ffi.stream()),
              "aser_score")
    return out


def argsort_desc(v):
    ffi.init()
    v = _f32(v)
    out = torch.empty(v.numel(), dtype=torch.int64, device=v.device)
    ffi.check(ffi.lib().ocl_argsort_desc(ffi.ptr(v), v.numel(), ffi.ptr(out), ffi.stream()), "argsort_desc")
    return out


def cosine_max(mem_grads, grad, out=None, eps=1e-8):
    """max_i cosine_similarity(mem_grads[i], grad) (utils/buffer/buffer_utils.py:51-56, gss_greedy_update.py:79,121) -> [1]."""
    ffi.init()
    mem_grads, grad = _f32(mem_grads), _f32(grad)
    k, n = mem_grads.shape
    if grad.numel() != n:
        raise RuntimeError("cosine_max: %d-vector against rows of %d" % (grad.numel(), n))
    if out is None:
        out = torch.empty(1, dtype=torch.float32, device=grad.device)
    ws = torch.empty(ffi.lib().ocl_cosine_max_workspace_bytes(k), dtype=torch.uint8, device=grad.device)
    ffi.check(ffi.lib().ocl_cosine_max(ffi.ptr(mem_grads), k, n, ffi.ptr(grad), float(eps), ffi.ptr(out), ffi.ptr(ws), ffi.stream()),
              "cosine_max")
    return out


# ---- K11 ---------------------------------------------------------------------------------------------
def ncm_class_means(feat, labels, class_ids):
    ffi.init()
    feat, labels, class_ids = _f32(feat), _i64(labels), _i64(class_ids)
    n, d = feat.shape
    nc = class_ids.numel()
    means = torch.zeros((nc, d), dtype=torch.float32, device=feat.device)
    counts = torch.zeros(nc, dtype=torch.int32, device=feat.device)
    ffi.check(ffi.lib().ocl_ncm_class_means(ffi.ptr(feat), ffi.ptr(labels), n, d, ffi.ptr(class_ids), nc, ffi.ptr(means),
                                            ffi.ptr(counts), ffi.stream()), "ncm_class_means")
    return means, counts


def ncm_predict(feat, means):
    ffi.init()
    feat, means = _f32(feat), _f32(means)
    n, d = feat.shape
    pred = torch.empty(n, dtype=torch.int64, device=feat.device)
    ffi.check(ffi.lib().ocl_ncm_predict(ffi.ptr(feat), n, d, ffi.ptr(means), means.shape[0], ffi.ptr(pred), ffi.stream()),
              "ncm_predict")
    return pred


# ---- K12 ---------------------------------------------------------------------------------------------
def mir_scores(logits_pre, logits_post, y):
    ffi.init()
    a, b, y = _f32(logits_pre), _f32(logits_post), _i64(y)
    n, c = a.shape
    out = torch.empty(n, dtype=torch.float32, device=a.device)
    ffi.check(ffi.lib().ocl_mir_scores(ffi.ptr(a), ffi.ptr(b), ffi.ptr(y), n, c, ffi.ptr(out), ffi.stream()), "mir_scores")
    return out


# ---- K13 ---------------------------------------------------------------------------------------------
AUG_NPARAM = 12
AUG_NUNIFORM = 30


def scr_augment(x, params):
    ffi.init()
    x, params = _f32(x), _f32(params)
    n, c, h, w = x.shape
    if c != 3 or params.shape != (n, AUG_NPARAM):
        raise RuntimeError("scr_augment: x must be [n,3,h,w] and params [n,%d]" % AUG_NPARAM)
    out = torch.empty_like(x)
    ffi.check(ffi.lib().ocl_scr_augment(ffi.ptr(x), ffi.ptr(out), n, h, w, ffi.ptr(params), ffi.stream()), "scr_augment")
    return out


def scr_augment_uniform(x, u, cfg12, want_params=False, out=None):
    """Augmented view from raw uniform draws u [n, AUG_NUNIFORM] (device): parameter arithmetic and the image kernel in one call.
    out: a contiguous [n,3,h,w] tensor (or row slice of one) to write into."""
    ffi.init()
    x, u = _f32(x), _f32(u)
    n, c, h, w = x.shape
    if c != 3 or u.shape != (n, AUG_NUNIFORM) or len(cfg12) != 12:
        raise RuntimeError("scr_augment_uniform: x must be [n,3,h,w], u [n,%d], cfg 12 floats" % AUG_NUNIFORM)
    if out is None:
        out = torch.empty_like(x)
    elif out.shape != x.shape or out.dtype != torch.float32 or not out.is_contiguous():
        raise RuntimeError("scr_augment_uniform: out must be a contiguous float32 tensor of x's shape")
    params = torch.empty((n, AUG_NPARAM), dtype=torch.float32, device=x.device)
    cfg = (ffi.C.c_double * 12)(*[float(v) for v in cfg12])
    ffi.check(ffi.lib().ocl_scr_augment_uniform(ffi.ptr(x), ffi.ptr(out), n, h, w, ffi.ptr(u), cfg, ffi.ptr(params), ffi.stream()),
              "scr_augment_uniform")
    return (out, params) if want_params else out


def gemm_small(a, b, bias=None, relu=False, trans_b=False):
    """a[m,k] @ (b[k,n] or b[n,k]^T) — exposed for tests of the MFMA tile mapping."""
    ffi.init()
    a, b = _f32(a), _f32(b)
    m, k = a.shape
    n = b.shape[0] if trans_b else b.shape[1]
    c = torch.empty((m, n), dtype=torch.float32, device=a.device)
    b_rs, b_cs = (1, b.shape[1]) if trans_b else (b.shape[1], 1)
    ffi.check(ffi.lib().ocl_gemm_small(ffi.ptr(a), k, 1, ffi.ptr(b), b_rs, b_cs, ffi.ptr(c), n, m, n, k, ffi.ptr(bias),
                                       int(relu), 0, ffi.stream()), "gemm_small")
    return c


def prof_enable(on):
    ffi.check(ffi.lib().ocl_prof_enable(int(bool(on))), "prof_enable")


def prof_reset():
    ffi.check(ffi.lib().ocl_prof_reset(), "prof_reset")


def prof_query(cls):
    import ctypes as C
    ms = C.c_double(0)
    n = C.c_int64(0)
    ffi.check(ffi.lib().ocl_prof_query(int(cls), C.byref(ms), C.byref(n)), "prof_query")
    return ms.value, n.value


def mfma_calibrate(iters=20000):
    """(TFLOP/s, us) of a register-only v_mfma_f32_16x16x4_f32 stream on the current device, right now (measurement only:
    bench.py's `roofline.calibrated_peak`; blocks the host for ~iters * 60 ns)."""
    import ctypes as C
    ffi.init()
    scratch = torch.empty(256 * 1024, dtype=torch.float32, device=torch.device("cuda", torch.cuda.current_device()))
    tf, us = C.c_double(0), C.c_double(0)
    ffi.check(ffi.lib().ocl_mfma_calibrate(int(iters), ffi.ptr(scratch), C.byref(tf), C.byref(us), ffi.stream()), "mfma_calibrate")
    return tf.value, us.value


def set_deterministic(on):
    """Order-independent (integer) BatchNorm batch sums: bit-reproducible training steps at ~12 % per step (include/ocl_hip.h)."""
    ffi.init()
    ffi.check(ffi.lib().ocl_set_deterministic(int(bool(on))), "set_deterministic")
