/*
 * ocl_hip.h — C-ABI of libocl_hip.so: the MI355X (gfx950) replay-step hot path of
 * RaptorMai/online-continual-learning, hand-written HIP.
 *
 * The reference is pure Python on PyTorch and defines no FFI; every entry point
 * here replaces an ATen op *sequence* inside the reference and cites the file:line it stands in
 * for.  The reference-side binding (a ctypes stub) is shown in INTEGRATION.md.
 *
 * Conventions (all entry points):
 *   - plain C: raw device pointers + sizes, no torch types.  `stream` is a hipStream_t passed as
 *     void* (NULL = the null stream).  Calls are stream-ordered and never synchronise the host.
 *   - caller owns all memory (PyTorch tensors on the Python side); the library allocates nothing on
 *     the device.  Scratch comes from caller-provided workspaces whose sizes are queried first.
 *   - return 0 on success, <0 on error; ocl_last_error() returns a thread-local message.  The Python
 *     wrapper turns non-zero into RuntimeError (the reference's error convention is exceptions:
 *     utils/loss.py:36-50, utils/buffer/reservoir_update.py:46-51).
 *   - labels / indices are int64 (PyTorch LongTensor) everywhere; floats are fp32 (exact-fp32 MFMA,
 *     no reduced precision anywhere).
 */
#ifndef OCL_HIP_H
#define OCL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OCL_OK 0
#define OCL_ERR_ARG (-1)
#define OCL_ERR_HIP (-2)
#define OCL_ERR_STATE (-3)
#define OCL_ERR_UNSUPPORTED (-4)

/* ---- library ---------------------------------------------------------------------------------- */
int ocl_version(void);
const char* ocl_last_error(void);
/* Selects the device and verifies it is gfx950; fails loudly otherwise. */
int ocl_init(int device);

/* ---- K9: replay-buffer row gather / scatter ----------------------------------------------------
 * replaces buffer.buffer_img[indices] / buffer_label[indices] (utils/buffer/buffer_utils.py:19-21,
 * 115-116) and the slot overwrite buffer_img[idx] = x (utils/buffer/reservoir_update.py:59-60,
 * utils/buffer/aser_update.py:111-112).  row_bytes must be a multiple of 4.  Duplicate indices in a
 * scatter are the caller's problem (the reference de-duplicates with a dict first). */
int ocl_gather_rows(const void* src, const int64_t* idx, int64_t n, int64_t row_bytes, void* dst,
                    void* stream);
int ocl_scatter_rows(void* dst, const int64_t* idx, int64_t n, int64_t row_bytes, const void* src,
                     void* stream);
/* The pair buffer.buffer_img[indices], buffer.buffer_label[indices] of every retrieval (utils/buffer/buffer_utils.py:19-21,115-116;
 * utils/buffer/aser_utils.py:152-155) as ONE call and one launch: rows of two arrays by the same index vector.  idx_host != NULL:
 * the indices were drawn on the host (numpy / torch-CPU generators); they are uploaded into idx_dev (n int64, caller-owned) first,
 * asynchronously.  idx_host == NULL: idx_dev already holds them (e.g. the ranking of ocl_argsort_desc). */
int ocl_gather_rows_pair(const void* src_a, int64_t row_bytes_a, void* dst_a, const void* src_b, int64_t row_bytes_b, void* dst_b,
                         const int64_t* idx_host, int64_t* idx_dev, int64_t n, void* stream);
/* Host -> device upload of a small host array (index vectors the reference builds with torch.tensor(...) / torch.from_numpy(...)
 * and moves with maybe_cuda: utils/buffer/buffer_utils.py:17-21, reservoir_update.py:52-60).  Asynchronous on `stream`; `host`
 * may be reused as soon as the call returns (payloads <= 64 KB are staged through pinned memory). */
int ocl_upload(const void* host, int64_t nbytes, void* dev, void* stream);
/* dataset_transform + ToTensor for a whole minibatch (continuum/data_utils.py:38-54,
 * utils/setup_elements.py:29-43): gathers n HWC uint8 images by index from a device-resident task
 * tensor and writes CHW fp32 / 255. */
int ocl_gather_u8_hwc_to_f32_chw(const uint8_t* src, const int64_t* idx, int64_t n, int h, int w,
                                 int c, float* dst, void* stream);

/* ---- K8: SGD -------------------------------------------------------------------------------------
 * torch.optim.SGD.step with momentum 0 (utils/setup_elements.py:73-75): p <- p - lr*(g + wd*p).
 * With out != NULL the result goes to `out` and p is untouched: that is MIR's virtual step on a
 * deepcopy (utils/buffer/mir_retrieve.py:34-47) without the copy. grad_scale multiplies g first
 * (review trick divides grads by 10, agents/base.py:84-87). */
int ocl_sgd_step(float* params, const float* grads, int64_t n, float lr, float weight_decay,
                 float grad_scale, float* out, void* stream);

/* ---- K6: softmax cross-entropy ---------------------------------------------------------------------
 * torch.nn.CrossEntropyLoss(reduction='mean') (agents/base.py:95,113) and
 * F.cross_entropy(reduction='none') (utils/buffer/mir_retrieve.py:26-27).
 * reduction: 0 = none (loss_out[n]), 1 = mean (loss_out[1]).  dlogits (may be NULL) receives
 * d(loss)/d(logits) for reduction=mean, or d(loss_i)/d(logits_i) per row for reduction=none. */
int ocl_ce_fwd_bwd(const float* logits, const int64_t* y, int n, int c, int reduction,
                   float* loss_out, float* dlogits, void* stream);

/* Cross-entropy (mean) over a column segment: the labels trick (agents/base.py:96-101: softmax over the classes present in
 * the batch) and the separated softmax (:102-108: old and new classes normalised separately) as ONE kernel.  seg[c] assigns
 * every logit column to a segment id >= 0, or -1 (column takes no part); row r is a softmax over the columns j with
 * seg[j] == seg[y[r]] (seg[y[r]] must be >= 0: checked by the caller, the labels live on the host).  dlogits (may be NULL)
 * receives d(mean loss)/d(logits): zero outside the row's segment. */
int ocl_ce_segmented_fwd_bwd(const float* logits, const int64_t* y, const int32_t* seg, int n, int c, float* loss_out,
                             float* dlogits, void* stream);

/* Knowledge-distillation loss of the KD tricks (utils/kd_manager.py:6-11, loss_fn_kd):
 * mean_r(-sum_j softmax(target_r/T)_j * log_softmax(scores_r/T)_j) * T^2; dscores (may be NULL) = d(loss)/d(scores). */
int ocl_kd_fwd_bwd(const float* scores, const float* target_scores, int n, int c, float T, float* loss_out, float* dscores,
                   void* stream);

/* ---- K7: supervised contrastive loss ---------------------------------------------------------------
 * SupConLoss.forward with contrast_mode='all' (utils/loss.py:19-96).  feat is VIEW-MAJOR
 * [n_views*bsz, dim] (= torch.cat(torch.unbind(features,1)), loss.py:56).  workspace: at least
 * ocl_supcon_workspace_bytes(bsz*n_views).  dfeat may be NULL (loss only). An anchor without any
 * positive yields NaN exactly as the reference's 0/0 (loss.py:90). */
int64_t ocl_supcon_workspace_bytes(int n_anchor);
int ocl_supcon_fwd_bwd(const float* feat, const int64_t* y, int bsz, int n_views, int dim,
                       float temperature, float* loss_out, float* dfeat, void* workspace,
                       void* stream);

/* ---- K10: kNN Shapley values -----------------------------------------------------------------------
 * sorted_cand_ind + compute_knn_sv on precomputed deep features (utils/buffer/aser_utils.py:7-61,
 * 94-116; distance = sum((u-v)^2), utils/utils.py:93-95).  One workgroup per evaluation row:
 * distances -> LDS bitonic sort (ties broken by ascending candidate index) -> label indicator ->
 * closed-form suffix recursion -> scatter to candidate order.  n_cand <= OCL_KNN_MAX_CAND.
 * sorted_idx (may be NULL) receives the per-row ascending-distance candidate order. */
#define OCL_KNN_MAX_CAND 2048
int ocl_knn_sv(const float* eval_f, const int64_t* eval_y, int n_eval, const float* cand_f,
               const int64_t* cand_y, int n_cand, int dim, int k, float* sv_out,
               int64_t* sorted_idx, void* stream);
/* column reduction over the evaluation rows: mode 0 sum, 1 mean, 2 max, 3 min
 * (aser_retrieve.py:79-86, aser_update.py:80). */
int ocl_col_reduce(const float* m, int rows, int cols, int mode, float* out, void* stream);
/* ASER score (aser_retrieve.py:77-86): type 0 "asvm": coop.mean(0) - adv.mean(0); 1 "asv":
 * coop.max(0) - adv.min(0); 2 "neg_sv": -adv.sum(0) (coop ignored, may be NULL). */
int ocl_aser_score(const float* sv_adv, int n_adv, const float* sv_coop, int n_coop, int n_cand,
                   int type, float* out, void* stream);
/* sv.argsort(descending=True) (aser_retrieve.py:88, aser_update.py:88; scores.sort(descending)
 * mir_retrieve.py:29).  Deterministic: ties keep ascending index. n <= OCL_SORT_MAX. */
#define OCL_SORT_MAX 4096
int ocl_argsort_desc(const float* v, int n, int64_t* idx_out, void* stream);

/* ---- K11: nearest-class-mean classifier ------------------------------------------------------------
 * agents/base.py:121-142 (means) and :159-176 (predict).  feat rows are L2-normalised, averaged per
 * class, the mean re-normalised.  class_ids[n_cls] lists the labels in `old_labels` order; a class
 * with no exemplar gets count 0 and its mean row is left untouched (the caller fills it the way the
 * reference does, base.py:135-137).  predict returns argmin_j ||f/|f| - mean_j||^2 as an index
 * into class_ids (first minimum wins, like torch.min). */
int ocl_ncm_class_means(const float* feat, const int64_t* labels, int n, int d,
                        const int64_t* class_ids, int n_cls, float* means_out, int32_t* counts_out,
                        void* stream);
int ocl_ncm_predict(const float* feat, int n, int d, const float* means, int n_cls,
                    int64_t* pred_out, void* stream);

/* ---- K12: MIR interference score -------------------------------------------------------------------
 * post_loss - pre_loss with per-sample CE (utils/buffer/mir_retrieve.py:26-28). */
int ocl_mir_scores(const float* logits_pre, const float* logits_post, const int64_t* y, int n,
                   int c, float* scores_out, void* stream);

/* ---- GSS-Greedy: gradient-direction similarity ------------------------------------------------------
 * max_i cosine_similarity(mem[i], g) over k stored flat gradient vectors of n floats (utils/buffer/buffer_utils.py:51-56:
 * x1.x2 / max(|x1||x2|, eps); call sites utils/buffer/gss_greedy_update.py:79,121 `max(cosine_similarity(mem_grads, grad))`).
 * workspace: ocl_cosine_max_workspace_bytes(k) bytes. out: one float. */
int64_t ocl_cosine_max_workspace_bytes(int k);
int ocl_cosine_max(const float* mem, int k, int64_t n, const float* g, float eps, float* out, void* workspace, void* stream);

/* ---- K13: SCR view augmentation --------------------------------------------------------------------
 * stands in for the kornia pipeline of agents/scr.py:18-24 (RandomResizedCrop -> HorizontalFlip ->
 * ColorJitter -> RandomGrayscale); kornia 0.4.1's RNG parameterisation is unpinned, so
 * the per-sample parameters are drawn on the host and passed in.  params: n rows of
 * OCL_AUG_NPARAM floats: [y0,x0,crop_h,crop_w (input pixels, fractional), flip(0/1),
 * jitter_on(0/1), brightness, contrast, saturation, hue (fraction of a turn), order (0..23 index
 * of the permutation of the four jitter ops), gray(0/1)]. x, out: [n,3,h,w] fp32 in [0,1]. */
#define OCL_AUG_NPARAM 12
int ocl_scr_augment(const float* x, float* out, int n, int h, int w, const float* params,
                    void* stream);
/* Same, with the parameter arithmetic on the device: u holds n rows of OCL_AUG_NUNIFORM raw U[0,1) draws (host generator,
 * one torch.rand call as before): [0,10) crop areas, [10,20) crop log-ratios of the 10 attempts RandomResizedCrop makes,
 * [20,30) position y / x, flip, jitter-on, brightness, contrast, saturation, hue, order, gray.  cfg12 (HOST array):
 * scale_lo, scale_hi, ratio_lo, ratio_hi, brightness, contrast, saturation, hue, p_jitter, p_gray, fallback crop w, h.
 * params [n, OCL_AUG_NPARAM] (device) receives the derived per-image parameters (the layout ocl_scr_augment takes). */
#define OCL_AUG_NUNIFORM 30
int ocl_scr_augment_uniform(const float* x, float* out, int n, int h, int w, const float* u, const double* cfg12,
                            float* params, void* stream);

/* ---- small dense GEMM (K5 helper; exposed for tests) ----------------------------------------------
 * C[m,n] = A(m,k) * B(k,n) (+ bias[n]) (relu) with arbitrary element strides, exact-fp32 MFMA
 * 16x16x4.  Used for nn.Linear fwd/bwd (models/resnet.py:79,103,148-152). */
int ocl_gemm_small(const float* a, int64_t a_rs, int64_t a_cs, const float* b, int64_t b_rs,
                   int64_t b_cs, float* c, int64_t c_rs, int m, int n, int k, const float* bias,
                   int relu, int accumulate, void* stream);

/* ---- K1-K5: Reduced-ResNet18 engine ----------------------------------------------------------------
 * One object per model (models/resnet.py:69-116 Reduced_ResNet18; :140-168 SupConResNet).  The
 * parameter order/layout is exactly PyTorch's named_parameters() order of the reference module, as
 * one flat fp32 array (so the flat gradient IS the vector get_grad_vector builds,
 * utils/buffer/buffer_utils.py:58-71).  All BatchNorm running statistics live in one flat array:
 * per BN, running_mean[C] then running_var[C], in module order; num_batches_tracked is int64[n_bn].
 */
typedef struct ocl_net ocl_net;

typedef struct {
    int32_t in_h, in_w;     /* 32x32 (CIFAR) or 84x84 (Mini-ImageNet), utils/setup_elements.py:11-17 */
    int32_t nf;             /* 20 (Reduced_ResNet18, models/resnet.py:112-116) */
    int32_t n_classes;      /* size of the encoder's `linear` (always present as parameters) */
    int32_t head;           /* 0: logits = linear(features)           (ResNet.forward, :106-109)
                               1: normalize(mlp(features))            (SupConResNet head='mlp')
                               2: normalize(linear_head(features))    (head='linear')
                               3: normalize(features)                 (head='None') */
    int32_t feat_dim;       /* 128: SupCon projection size (head 1,2) */
    int32_t max_batch;      /* largest n ever passed to forward */
    int32_t n_slots;        /* activation tapes kept alive for backward (>=1) */
} ocl_net_desc;

int ocl_net_create(const ocl_net_desc* desc, ocl_net** out);
void ocl_net_destroy(ocl_net* net);

int64_t ocl_net_param_count(const ocl_net* net);       /* floats in the flat parameter array */
int32_t ocl_net_num_tensors(const ocl_net* net);       /* parameter tensors, named_parameters() order */
/* name (<=63 chars + NUL), flat offset, ndim<=4, shape */
int ocl_net_tensor_info(const ocl_net* net, int i, char* name64, int64_t* offset, int32_t* ndim,
                        int64_t* shape4);
int32_t ocl_net_num_bn(const ocl_net* net);
int64_t ocl_net_bn_stat_count(const ocl_net* net);     /* floats in the flat running-stat array */
int ocl_net_bn_info(const ocl_net* net, int i, char* name64, int64_t* offset, int32_t* channels);
int32_t ocl_net_feature_dim(const ocl_net* net);        /* 160 (32x32) / 640 (84x84) */
int32_t ocl_net_out_dim(const ocl_net* net);            /* n_classes, feat_dim or feature_dim */
int64_t ocl_net_workspace_bytes(const ocl_net* net);

/* Binds caller-owned storage. params/grads: param_count floats; running: bn_stat_count floats;
 * nbt: int64[num_bn]; workspace: workspace_bytes, 256-B aligned. */
int ocl_net_bind(ocl_net* net, float* params, float* grads, float* running, int64_t* nbt,
                 void* workspace, int64_t workspace_bytes);

#define OCL_FWD_TRAIN 1u          /* BatchNorm uses batch statistics (model.train()) */
#define OCL_FWD_SAVE_TAPE 2u      /* keep activations in `slot` for ocl_net_backward */
#define OCL_FWD_UPDATE_RUNNING 4u /* momentum-0.1 running-stat update (nn.BatchNorm2d default);
                                     applied once per group, in group order */
#define OCL_FWD_FROZEN_BN 8u      /* with OCL_FWD_SAVE_TAPE and without OCL_FWD_TRAIN: eval-mode BatchNorm (running statistics) but
                                     the activations are kept, so that ocl_net_backward gives the gradients of an eval-mode
                                     forward: model.eval() followed by loss.backward(), utils/buffer/gss_greedy_update.py:16,
                                     77-79,97-100,116-118 */
#define OCL_FWD_SAME_WEIGHTS 16u  /* the caller asserts that the parameter array of this call has not been written since this net's
                                     previous forward read it (several forwards between two optimiser steps: the ASER retrieval's
                                     feature pass, the memory pass and the combined pass of agents/exp_replay.py:49-84): the engine
                                     reuses the weight packs it made then instead of re-packing (it still re-packs when its arena
                                     holds another array's packs, e.g. after a params_override call) */
#define OCL_FWD_PACK_ALL 32u      /* with a pass that packs (no OCL_FWD_SAME_WEIGHTS, or refused): also write the data-gradient packs
                                     although this pass keeps no tape -- a taped pass on the same weights will follow */
/* x: [n,3,H,W] fp32 NCHW (what the reference's agents hand to model.forward).
 * groups: the batch is `groups` equal consecutive sub-batches that the reference would have run as
 * separate forward calls (SCR's two views, agents/scr.py:55): BatchNorm statistics are per group.
 * params_override: NULL = bound params; else another flat parameter array (MIR's virtual model,
 * mir_retrieve.py:21,25).  feat_out [n,feature_dim] and out [n,out_dim] may each be NULL. */
int ocl_net_forward(ocl_net* net, const float* x, int n, int groups, uint32_t flags,
                    const float* params_override, float* feat_out, float* out, int slot,
                    void* stream);
/* The same pass over a batch given as `nseg` (1..8) separate tensors xs[i] of ns[i] images each, in batch order: the reference's
 * torch.cat((mem_x, batch_x)) (agents/exp_replay.py:78, agents/scr.py:52) and its one-forward-call-per-view (agents/scr.py:55),
 * torch.cat((eval_x, cand_x)) of utils/buffer/aser_utils.py:73 -- without materialising the concatenation: the engine's layout
 * conversion reads the segments where they are. */
int ocl_net_forward_segments(ocl_net* net, const float* const* xs, const int32_t* ns, int nseg, int groups, uint32_t flags,
                             const float* params_override, float* feat_out, float* out, int slot, void* stream);
/* Backward of the forward recorded in `slot`. dout: [n,out_dim] = d(loss)/d(out).
 * accumulate=0 overwrites the bound flat gradient, 1 adds to it (loss.backward() twice,
 * agents/exp_replay.py:55,77).  Tensors that take no part in the forward (SupConResNet's
 * encoder.linear) get zero / are left untouched respectively. */
int ocl_net_backward(ocl_net* net, int slot, const float* dout, int accumulate, void* stream);

/* Test hooks. debug_stop: make ocl_net_backward return right after stage block*10+step (step 1: bn2 backward,
 * 2: conv2 data gradient, 3: bn1 backward, 4: conv1 data gradient, 5: block input gradient complete; 990: after the
 * head; -1: off).  debug_copy what: 0 raw conv output (NHWC) of conv `index`; 1 output of block `index`; 2 gradient
 * scratch buffer `index`; 3 gradient buffer by role (0..4 = gA..gE) at the last stop; 4 activation a1 of block `index`; 5 stem output. */
int ocl_net_debug_stop(ocl_net* net, int stage);
int ocl_net_debug_copy(ocl_net* net, int slot, int what, int index, float* dst, int64_t max_floats,
                       int64_t* n_written, void* stream);

/* ---- kernel-level entry points (single layers; tests and micro-benchmarks) ------------------------------
 * BatchNorm2d backward (train mode) fused with the ReLU mask that follows it: dpre = dz * (zmask > 0) (zmask NULL =
 * no ReLU); dy = gamma*invstd*(dpre - mean(dpre) - xhat*mean(dpre*xhat)); dgamma = sum(dpre*xhat), dbeta = sum(dpre).
 * Tensors are NHWC [groups*m_per_group, c]; mean/invstd are [groups, c]. scratch: groups*2*c accumulator cells of 16 bytes
 * (= groups*4*c doubles, 16-byte aligned): the batch sums are accumulated as integers, independent of the order in which the
 * workgroups finish. */
int ocl_bn_bwd_nhwc(const float* dz, const float* zmask, const float* y, const float* mean, const float* invstd,
                    const float* gamma, int64_t m_per_group, int groups, int c, float* dy, float* dgamma,
                    float* dbeta, int accumulate, double* scratch, void* stream);

/* Run-to-run reproducibility.  The BatchNorm batch sums (forward statistics, backward reductions) are the only accumulations of a
 * step whose order depends on scheduling.  on = 1: they are accumulated as fixed-point integers (associative): every weight is
 * bit-identical from run to run, as the reference's CPU path is at a fixed thread count; costs ~12 % per step (two atomics per
 * partial sum).  on = 0 (default; OCL_DETERMINISTIC=1 in the environment starts with 1): fp64 atomics.  Synchronises the device;
 * call it between steps, not inside one. */
int ocl_set_deterministic(int on);

/* ---- measurement helpers --------------------------------------------------------------------------
 * HIP-event timing on the caller's stream (bench.py's roofline leg: torch.cuda.Event only sees
 * torch's current stream).  Kernel-class accumulators are filled when profiling is enabled. */
int ocl_prof_enable(int on);
int ocl_prof_reset(void);
/* cls: 0 conv fwd/dgrad GEMM, 1 conv wgrad, 2 batchnorm/elementwise, 3 head/loss, 4 kNN/buffer.
 * Returns accumulated milliseconds and launch count since reset (synchronises the device). */
int ocl_prof_query(int cls, double* ms, int64_t* launches);
/* What the fp32 MFMA pipe of THIS box delivers right now: a register-only stream of v_mfma_f32_16x16x4_f32 (four independent
 * accumulators per wave, one wave per SIMD on every CU, `iters` rounds; no memory traffic), timed with HIP events on `stream`
 * (synchronises the host).  `scratch` receives 256 * n_cu floats (n_cu <= 1024).  bench.py reports it as `roofline.calibrated_peak`
 * next to the nominal peak so that a clock- or power-limited box shows in the record (boxes of this pool differ by up to ~19 % on the
 * MFMA-dense kernels).  There is no reference counterpart: measurement only. */
int ocl_mfma_calibrate(int iters, float* scratch, double* tflops, double* us, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OCL_HIP_H */
