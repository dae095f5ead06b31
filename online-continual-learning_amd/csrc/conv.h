// Internal interface of the conv / batch-norm kernels (conv.hip) used by the network engine (net.hip).
// Activations are NHWC fp32 with C in {4,20,40,80,160}; weights are re-packed per call from the
// PyTorch OIHW master copy.  Everything is exact fp32 (v_mfma_f32_16x16x4_f32).
#pragma once
#include "common.h"
#include <vector>

namespace ocl {

constexpr int kStatReps = 8;   // replicas of the BatchNorm statistic accumulators (atomics contention)

// One batch-sum accumulator.  Workgroups add their partial sums with atomics in whatever order they finish.  Default: the first word
// holds a double (fp64 atomics): the total -- and with it the normalised activations, the gradients and the stepped weights -- depends
// on that order in the last bit.  Deterministic mode (ocl_set_deterministic(1) / OCL_DETERMINISTIC=1): the partial sums are converted
// to 2^-40 fixed point and added as INTEGERS (two 64-bit words: the low 32 bits of the fixed-point value and the rest), which is
// associative: the totals, and everything downstream, are bit-identical from run to run
// (tests/test_gpu_parity2.py::test_training_steps_are_bit_reproducible).  Resolution 2^-40 (9e-13) absolute per partial sum, range
// |sum| < 2^47; a non-finite partial sum poisons the cell (reads back as NaN).  Two atomics per sum instead of one: +12 % per step.
struct alignas(16) StatCell {
    unsigned long long lo;   // sum of the low 32 bits of the addends' fixed-point values
    long long hi;            // sum of the remaining bits (floor(v * 2^8))
};

enum ConvEpi : int {
    EPI_STORE = 0,       // out = acc
    EPI_STATS = 1,       // + per-(group,channel) sum / sum-of-squares into `stats` (fp64 atomics)
    EPI_AFFINE = 2,      // out = acc*scale[c] + shift[c]        (eval-mode BN folded)
    EPI_RES = 4,         // out += res[same offset]
    EPI_RESMASK = 8,     // out += res * (resmask > 0)           (ReLU-masked gradient of an identity shortcut)
    EPI_RELU = 16,       // out = max(out, 0)
    EPI_ACCUM = 32,      // out = out_old + value                (second gradient contribution)
    EPI_BNB = 64,        // data gradient that feeds a BatchNorm backward: ReLU mask + the BatchNorm's two batch sums here (ConvArgs::bnb_*)
};

struct ConvArgs {
    const float* in;
    float* out;
    const float* scale;
    const float* shift;
    const float* res;
    const float* resmask;
    StatCell* stats;      // [kStatReps][groups][2][Cout], replica stride stat_rep_stride cells
    int64_t stat_rep_stride;
    int N, Hin, Win, Cin;
    int Hout, Wout, Cout;
    int CoutP;                  // n_splits * 16*MT (channels the grid covers)
    int LH, LW, os, oy0, ox0;   // output lattice: (oy,ox) = (ly*os+oy0, lx*os+ox0)
    int is;                     // input step per lattice step
    int ntaps;
    int tdy[9], tdx[9], tw[9];  // input offset of each tap and its index in the weight pack
    int tpo[9];                 // LDS patch offset of each tap (floats)
    int d_c4, d_pc, d_row;      // patch-staging walk: 256 units = d_row rows + d_pc pixels + d_c4 float4s
    int min_dy, min_dx, max_dy, max_dx;
    int KC, CP, PC, PR;         // channels per LDS chunk, LDS pixel stride, patch cols, patch rows
    int ppi, imgs, tiles_per_img;
    int groups, group_size, tiles_per_group;
    int flags;
    int n_splits;               // grid.y
    // ---- channels x pixels orientation, K-contiguous operands --------------------------------------------------------
    const float* wT;            // K-grouped weight pack [tap][Cin/4][WPT][4]
    int C4tot, WPT;             // Cin/4 of the whole convolution; row stride (channels) of the pack
    int Qc, Qpad;               // (tap, channel-quad) groups of one channel chunk; rounded up to whole rounds of 4
    int QS, nstage, wres;       // groups per weight stage, stages per chunk; 1: all weights stay in LDS for the workgroup's lifetime
    int pipe;                   // staged weights through the three-buffer ring (conv_t_kernel<..., PIPE>; default, OCL_CONV_PIPE=0: two buffers)
    int aligned;                // 1: every tile starts at a lattice row and holds whole rows / whole images: a lane's pixel geometry is tile-invariant
    // output classes sharing one launch (the four parity classes of a stride-2 data gradient: same input window, disjoint taps and
    // output lattices).  cls_pack = ncls | ntaps(class 0) << 4 | ntaps(class 1) << 8 | ...: the taps are listed class by class;
    // cls_oyx bit 2c = oy0 of class c, bit 2c+1 = ox0.  One class (ncls = 1): an ordinary convolution.
    int cls_pack, cls_oyx;
    unsigned m_tpg, m_tpi, m_lw, m_ppi, m_kc4, m_pc, m_pr;   // ceil(2^32 / d) of the plan's divisors (exact quotients by one v_mul_hi)
    // ---- input transform (xf = 1): the train-mode BatchNorm + ReLU of the PRODUCING convolution applied while this convolution stages
    // its input patch (models/resnet.py:33: out = relu(bn1(conv1(x))) feeding conv2): `in` is the producer's raw output, the batch
    // statistics are final when this launch starts, every workgroup folds them into a per-(group, channel) scale / shift table in
    // LDS, and workgroup (0, 0) does what the BatchNorm kernel's first block did (saved mean / invstd for the backward, running
    // statistics).  Halo positions outside the image stay zero (the convolution pads the ACTIVATION, not the raw output).
    int xf;
    int patch_floats;           // LDS floats of the patch area (the transform table follows it)
    int qstat_off;              // conv_q_kernel: byte offset of its statistics scratch in LDS ([4 waves][4 rows + 1][2 * 20] floats)
    const StatCell* xf_stats;   // producer's statistics [kStatReps][groups][2][Cin], replica stride xf_rep_stride cells
    int64_t xf_rep_stride;
    int64_t xf_m_per_group;     // pixels per BatchNorm group of the producer's output
    const float* xf_gamma;
    const float* xf_beta;
    float* xf_save_mean;        // [groups][Cin]
    float* xf_save_invstd;
    float* xf_running_mean;     // null: no running-statistics update
    float* xf_running_var;
    int64_t* xf_nbt;
    float xf_momentum, xf_eps;
    // plan-constant tables in device memory (conv_plan_finalize): [ctab 16 | qoff Qpad | qrow Qpad | pad] [tile descriptors ntiles x 8]
    // [patch units 3 * PF x 256] [output pixels 3 * NT x 256], offsets in ints
    // ---- BatchNorm backward, reduction half, in this epilogue (EPI_BNB; autograd of models/resnet.py:33-36).  `out` is the gradient
    // w.r.t. the ReLU'd output of a train-mode BatchNorm whose raw input is bnb_y (same shape as `out`): the epilogue applies the ReLU
    // mask (from the post-ReLU activation bnb_z, or, when that was never written because the forward applied the BatchNorm inside the
    // consuming convolution, recomputed as fma(y, scale, shift) > 0 with the forward's own arithmetic), stores the MASKED gradient d
    // and adds sum(d) and sum(d * (y - mean)) per (group, channel) to `stats` (the forward's replicated fp64 scheme): what
    // bn_bwd_fused_kernel needed a grid-wide arrival for.  bn_bwd_apply_e_kernel then only streams: dy = f(d, y, sums).
    const float* bnb_y;
    const float* bnb_z;         // null: mask recomputed from bnb_y
    const float* bnb_mean;      // [groups][Cout]
    const float* bnb_invstd;    // [groups][Cout]
    const float* bnb_gamma;     // [Cout]
    const float* bnb_beta;      // [Cout]
    int bnb_lds;                // float offset (from the end of the patch area) of the epilogue's LDS table [groups][Cout/4][3][4]: scale, shift, mean quads (planner; -1: no room reserved)
    const int* blob;
    int off_tdesc, off_pu, off_loc, blob_ints;
    unsigned long long* trace;  // measurement only (kbench): per workgroup 64 s_memtime stamps of wave 0 at the phase boundaries
};

struct ConvPlan {
    ConvArgs a;
    int cs;                     // 1: conv_s_kernel (16-pixel tiles, the four waves split the input channels; MT = NT = 1)
    int cw;                     // 1: conv_w_kernel (convw.hip: 512-thread workgroups, weights resident, every wave stages and multiplies its own pixel tiles)
    int q4;                     // > 0: conv_q_kernel<q4, ...> (4x4x1 MFMA, <= 20 output channels), q4 = 64-pixel sets per wave; MT = 5 blocks of 4 channels, NT = q4
    int MT, NT;                 // 16-channel tiles and 16-pixel tiles per wave (conv_t_kernel<MT, NT, ...>)
    int grid_x, grid_y;
    size_t lds_bytes;
    // the tables' upload (conv_plan_finalize with an arena: asynchronous, on the stream of the plan's first launch): a launch on ANOTHER
    // stream waits for this event first
    hipEvent_t ready;
    hipStream_t ready_stream;
};

// geometry description used by the planner
struct ConvGeomDesc {
    int N, groups;
    int Hin, Win, Cin;          // input tensor
    int Hout, Wout, Cout;       // output tensor (storage dims)
    int LH, LW, os, oy0, ox0, is;
    int ntaps;
    int tdy[9], tdx[9], tw[9];
    int force_MT, force_NT, force_bpc;   // 0 = planner's choice (benchmarks / tests)
    int force_cs;                        // conv_s_kernel for few output pixels: 0 = planner (OCL_CONV_S, default on), 1 = always where it fits, -1 = never
    int force_q4;                        // conv_q_kernel for <= 20 output channels: 0 = planner (OCL_CONV_Q4, default on), 1 = always where it fits, -1 = never
    int force_pipe;                      // staged-weight schedule: 0 = environment (OCL_CONV_PIPE, default on), 1 = ring, -1 = two-buffer
    int force_cw;                        // conv_w_kernel (wave-autonomous tiles, convw.hip): 0 = planner (OCL_CONV_W), 1 = always where it fits (force_MT / force_NT honoured), -1 = never
    int WPT;                    // row stride of the K-grouped weight pack (0: the plan's own CoutP)
    int xf;                     // reserve LDS for the input-transform table (ConvArgs::xf may then be set at launch)
    int bnb;                    // reserve LDS for the BatchNorm-backward epilogue table (EPI_BNB may then be set at launch)
    int ncls;                   // > 1: output classes of one launch, taps listed class by class
    int cls_ntaps[4], cls_oy[4], cls_ox[4];
};

int plan_conv(const ConvGeomDesc& g, ConvPlan* p);
// the plan's tables as the kernel reads them (host arithmetic only); conv_plan_finalize puts them into device memory and sets
// p->a.blob -- a plan must be finalized before launch_conv (the engine does it at a plan's first launch)
void conv_plan_tables(const ConvPlan& p, std::vector<int>* out);
// Device memory for plan tables: 8 MB chunks (hipMalloc only when a chunk fills up -- batch shapes first seen in the steady state, like
// the varying evaluation-set sizes of the ASER update, must not pay an allocation + a blocking copy per plan), uploads are asynchronous
// on the stream the plan is about to be launched on, the host copies stay alive with the arena.
struct PlanArena {
    std::vector<void*> chunks;
    size_t used = 0, cap = 0;
    std::vector<std::vector<int>*> host_keep;
    std::vector<hipEvent_t> events;   // one per plan: its tables have landed
};
void plan_arena_release(PlanArena* a);
// arena == nullptr: one hipMalloc + one blocking copy for this plan (measurement tools)
int conv_plan_finalize(ConvPlan* p, PlanArena* arena = nullptr, hipStream_t s = nullptr);
void conv_plan_release(ConvPlan* p);   // plans finalized without an arena

// One convolution layer of the network (models/resnet.py:10-12,25-30): shapes and weight-pack row strides.
struct ConvShape {
    int Cin, CinT, Cout, k, stride, Hin, Win, Ho, Wo;  // CinT: channels of the NHWC input tensor (stem: 3 -> 4)
    int CoutP, CiP;                                    // row strides of the forward / data-gradient weight packs
};
int pack_width(int channels);   // row stride of a weight pack with `channels` columns (multiple of 16)
// forward geometry; data-gradient geometry ("input" = dy, "output" = dx): one launch for stride 1 and for the 1x1
// stride-2 shortcut, four dense parity classes of the dx lattice for 3x3 stride 2 (merge_classes: as ONE description with
// four output classes when the lattices coincide, i.e. even input height and width).
void geom_fwd(const ConvShape& c, int N, int groups, ConvGeomDesc* g);
// groups > 1: the tiles follow the BatchNorm groups of the pass (no tile straddles two groups: the EPI_BNB epilogue sums per group)
void geom_dgrad(const ConvShape& c, int N, std::vector<ConvGeomDesc>* out, bool merge_classes = false, int groups = 1);
int launch_conv(const ConvPlan& p, hipStream_t s);
// conv_w_kernel (convw.hip): its planner (OCL_ERR_ARG: the geometry does not fit the form), its tables, its launch, its per-device set-up
int plan_conv_w(const ConvGeomDesc& g, ConvPlan* p);
void conv_w_tables(const ConvPlan& p, std::vector<int>* out);
int launch_conv_w(const ConvPlan& p, hipStream_t s);
int convw_kernels_init();
int convw_set_det(int on);   // this translation unit's copy of the batch-sum mode flag (conv_stats_dev.h)

// ---- wgrad -------------------------------------------------------------------------------------------
struct WgradArgs {
    const float* x;     // conv input  [N,Hin,Win,Cin]
    const float* dy;    // grad wrt conv output [N,Ho,Wo,Cout]
    float* partial;     // [S][Mrows][CoutP]   Mrows = nchunks*ntaps*KC rounded up per block
    int N, Hin, Win, Cin, Ho, Wo, Cout, CoutP;
    int stride, ntaps;
    int tdy[9], tdx[9];
    int min_dy, min_dx, max_dy, max_dx;
    int KC, nchunks;            // channel chunking of Cin
    int d_c4, d_pc, d_row;      // patch-prefetch walk (as ConvArgs)
    float inv_PR;
    int CP, PC, PR, DP;         // LDS strides
    int KP;                     // pixels per tile (multiple of 4)
    int ppi, imgs, tiles_per_img, total_tiles;
    int S;                      // pixel splits (grid.x)
    int mblocks_per_chunk;      // blocks along (tap,cc) per chunk
    int nblocks;                // blocks along Cout
    int Mchunk;                 // ntaps*KC
    int Mrows_total;            // nchunks * mblocks_per_chunk * (64*MTW)
    // input transform (as ConvArgs::xf): x is the raw output of the convolution in front of a BatchNorm + ReLU; the patch staging
    // applies max(fma(x, scale, shift), 0) with scale / shift from the saved mean / invstd, gamma, beta of that BatchNorm
    int xf, xf_groups, xf_group_size;   // BatchNorm groups of the pass, images per group
    const float* xf_mean;       // [groups][Cin]
    const float* xf_invstd;
    const float* xf_gamma;
    const float* xf_beta;
    int xcd_by;                 // > 0: one-dimensional launch of S * xcd_by workgroups in the XCD-aware order (conv_wgrad_kernel), = grid_y of the plan
    unsigned long long* trace;  // measurement only (kbench wgradtrace): per workgroup 64 s_memtime stamps of thread 0 at the phase boundaries
};
struct WgradPlan {
    WgradArgs a;
    int MTW, NTW;
    int q_rgw;                  // > 0: the 4x4x1 form (conv_wgrad_kernel<1, 1, PF, q_rgw>: row groups per wave), layer 1 of the large passes
    int grid_x, grid_y;
    size_t lds_bytes;
    size_t partial_floats;
};
// all layers' weight gradients of a replay-sized pass in one launch (conv_wgrad_multi_kernel)
constexpr int kMaxWgradMulti = 24;
struct WgradMultiEntry {
    WgradArgs a;
    int variant, grid_x;
};
struct WgradMultiArgs {
    const WgradMultiEntry* tab;     // device table, one entry per layer
    int n;
    int start[kMaxWgradMulti + 1];  // first workgroup of each layer
};
struct WgradMultiTable {            // the device table + the host copy it was last written from
    void* dev = nullptr;
    std::vector<unsigned char> host;
    int uploads = 0;
};
int wgrad_multi_variant(const WgradPlan& p);   // form index inside conv_wgrad_multi_kernel, -1: this plan launches on its own
int launch_wgrad_multi(const WgradPlan* plans, int n, WgradMultiTable* t, hipStream_t s);
void wgrad_multi_release(WgradMultiTable* t);
// xf_groups > 0: reserve LDS for the input-transform table of that many BatchNorm groups
// wg_target > 0: workgroups the pixel split aims at (default 512: a launch of its own fills the machine)
int plan_wgrad(int N, int Hin, int Win, int Cin, int Ho, int Wo, int Cout, int ksize, int stride, WgradPlan* p, int xf_groups = 0, int wg_target = 0);
int launch_wgrad(const WgradPlan& p, hipStream_t s);
// sums the S partials and writes/accumulates the OIHW gradient
int launch_wgrad_reduce(const WgradPlan& p, float* grad_oihw, int accumulate, hipStream_t s);

// all layers' reductions in one launch (single-stream backward of replay-sized batches)
constexpr int kMaxReduceLayers = 24;
struct WgradReduceLayer {
    int64_t partial_off, grad_off;   // floats from `partial` / `grads`
    int S, Mrows_total, CoutP, mrows_chunk, KC, ntaps, CinReal, Cout, block0;
};
struct WgradReduceMulti {
    const float* partial;
    float* grads;
    int accumulate, n;
    WgradReduceLayer L[kMaxReduceLayers];
};
void wgrad_reduce_layer(const WgradPlan& p, int64_t partial_off, int64_t grad_off, WgradReduceLayer* d);
int launch_wgrad_reduce_multi(WgradReduceMulti m, hipStream_t s);

// ---- weight packing -----------------------------------------------------------------------------------
// K-grouped packs (the A operand of conv_t_kernel: 4 consecutive input channels of a (tap, channel quad) group contiguous):
// fwd:   wTf[t][ci/4][coP][ci%4] = w[co][ci][t]  (ci padded with zero quads up to CinP)      dgrad: wTd[t][co/4][ciP][co%4] = w[co][ci][t]
struct PackDesc {
    int64_t w_off;      // offset of the OIHW tensor in the flat parameter array
    int64_t tf_off;     // fwd pack in the arena, -1 = none
    int64_t td_off;     // data-gradient pack, -1 = none
    int Cout, Cin, ntaps, CinP, CoutP, CiP;
};
enum { PACK_TF = 1, PACK_TD = 2, PACK_ALL = 3 };   // which packs a launch writes
// zero_a / zero_b: two arrays of accumulator cells cleared by the same launch (the BatchNorm statistics arenas of the pass), may be null / 0
int launch_pack_weights(const float* params, float* arena, const PackDesc* descs_dev, int n_layers, int max_elems,
                        hipStream_t s, int mask = PACK_ALL, StatCell* zero_a = nullptr, int64_t zero_a_n = 0, StatCell* zero_b = nullptr,
                        int64_t zero_b_n = 0);

// ---- layout / elementwise ------------------------------------------------------------------------------
int launch_nchw3_to_nhwc4(const float* x, float* out, int N, int H, int W, hipStream_t s);
constexpr int kMaxInputSegments = 8;
struct InputSegments {
    const float* x[kMaxInputSegments];   // [n_i, 3, H, W] each
    int first[kMaxInputSegments];        // index of the segment's first image in the batch
    int n;                               // number of segments
};
int launch_nchw3_to_nhwc4_segments(const InputSegments& sg, float* out, int N, int H, int W, hipStream_t s);

struct BnFwdArgs {
    const float* y;       // raw conv output [M,C]
    float* z;             // output
    const float* res;     // optional residual (already normalised), same shape
    const StatCell* stats;  // [kStatReps][G][2][C] (replica stride stat_rep_stride cells), summed here
    int64_t stat_rep_stride;
    const float* gamma;
    const float* beta;
    float* running_mean;  // may be null (no update)
    float* running_var;
    int64_t* nbt;         // num_batches_tracked, may be null
    float* save_mean;     // [G][C]
    float* save_invstd;   // [G][C]
    int64_t m_per_group;  // pixels per group
    int G, C, relu;
    float momentum, eps;
    const float* frozen_mean;   // non-null: normalise with these statistics instead of the batch's (eval-mode BatchNorm kept on the
    const float* frozen_var;    // tape: the forward of model.eval() under autograd, utils/buffer/gss_greedy_update.py:16,77-79)
    // Second BatchNorm whose normalised output is this one's residual (the projection shortcut's, models/resnet.py:27-30,35:
    // out += self.shortcut(x)): z = relu(bn(y) + bn_b(yb)) in one launch -- the shortcut's own normalise launch and its output tensor
    // are gone.  Same arithmetic in the same order as the two launches (fma, then add): same bits.  yb == nullptr: none.
    const float* yb;
    const StatCell* stats_b;
    const float *gamma_b, *beta_b;
    float *running_mean_b, *running_var_b;
    int64_t* nbt_b;
    float *save_mean_b, *save_invstd_b;
    const float *frozen_mean_b, *frozen_var_b;
};
int launch_bn_fwd(const BnFwdArgs& a, hipStream_t s);
// z = relu(fma(y, scale, shift)) from saved statistics (the activation a fused pass never wrote: debug copies, tests)
int launch_bn_apply_saved(const float* y, const float* mean, const float* invstd, const float* gamma, const float* beta, float* z,
                          int64_t m_per_group, int G, int C, hipStream_t s);

// eval-mode fold: scale = gamma/sqrt(rv+eps), shift = beta - rm*scale for every BN at once
struct BnFoldDesc {
    int64_t gamma_off, beta_off, stat_off, out_off;
    int C;
};
int launch_bn_fold(const float* params, const float* running, float* out, const BnFoldDesc* descs_dev, int n_bn, float eps,
                   hipStream_t s);

struct BnBwdArgs {
    const float* dz;      // grad wrt block output (post-ReLU)   [M,C]
    const float* z;       // post-ReLU output (mask); null = no ReLU mask
    int64_t m_per_group;
    int G, C;
    int nsets;            // 1 or 2 BatchNorms sharing dz (main path + projection shortcut)
    const float* y[2];
    const float* mean[2];
    const float* invstd[2];
    const float* gamma[2];
    float* dy[2];         // grad wrt raw conv output
    float* dgamma[2];
    float* dbeta[2];
    StatCell* sums;       // scratch [nsets][G][2][C], zeroed by the caller
    unsigned* barrier;    // zeroed by the caller, or null: arrival counter of the one-pass kernel (nsets == 1, G <= 2, see launch_bn_bwd)
    StatCell* fsums;      // one-pass kernel: zeroed accumulators [8 replicas][G][2][C]
    StatCell* fsums_b;    // the same for the second BatchNorm (nsets == 2), null: two-kernel path for two sets
    int accumulate;       // dgamma/dbeta += (1) or = (0)
    unsigned* err;        // host-mapped asynchronous error word (one-pass kernel: arrival time-out), may be null
    int frozen;           // 1: the forward normalised with constant (running) statistics: dy = gamma*invstd*dpre, no mean terms
    // mask_from_y: the post-ReLU activation was never written (its BatchNorm + ReLU ran inside the consuming convolution's patch
    // staging): the ReLU mask is recomputed as fma(y, scale, shift) > 0 with exactly the arithmetic of that staging (nsets == 1, z null)
    int mask_from_y;
    const float* beta[2];
};
int launch_bn_bwd(const BnBwdArgs& a, hipStream_t s);
// Apply half of a BatchNorm backward whose reduction ran in the producing data gradient's epilogue (EPI_BNB): d is the masked gradient,
// esums the replicated sums [kStatReps][G][2][C] (replica stride esums_rep_stride cells) of d and d * (y - mean);
// dy = gamma * invstd * (d - mean(d) - xhat * mean(d * xhat)); block 0 writes dgamma / dbeta.  Pure streaming: no atomics, no arrival.
struct BnApplyEArgs {
    const float* d;
    const float* y;
    const float* mean;      // [G][C]
    const float* invstd;    // [G][C]
    const float* gamma;     // [C]
    float* dy;
    float* dgamma;
    float* dbeta;
    const StatCell* esums;
    int64_t esums_rep_stride;
    int64_t m_per_group;
    int G, C, accumulate;
};
int launch_bn_apply_e(const BnApplyEArgs& a, hipStream_t s);
void bn_bwd_tune(int block_cap, int unroll, int phase);
void bn_bwd_fused_enable(int on);   // one-pass kernel on/off (-1: OCL_BN_FUSED from the environment, default on)   // micro-benchmark overrides; 0 = default (phase 1 reduce only, 2 apply only)

// avg_pool2d(k=4) + flatten in PyTorch's (C,ph,pw) order; and its backward
int launch_avgpool_fwd(const float* z, float* feat, int N, int H, int W, int C, hipStream_t s);
int launch_avgpool_bwd(const float* dfeat, float* dz, int N, int H, int W, int C, hipStream_t s);

// F.normalize(dim=1) forward/backward
int launch_l2norm_fwd(const float* v, float* out, float* norms, int n, int d, hipStream_t s, float* out2 = nullptr);
int launch_l2norm_bwd(const float* out, const float* norms, const float* dout, float* dv, int n, int d, hipStream_t s);
// dx = dy * (a > 0)
int launch_relu_bwd(const float* dy, const float* a, float* dx, int64_t n, hipStream_t s);
// out[c] (+)= sum_r m[r][c]
int launch_colsum(const float* m, int rows, int cols, float* out, int accumulate, hipStream_t s);
int launch_fill(float* p, int64_t n, float v, hipStream_t s);

int conv_kernels_init();
int wgrad_kernels_init();   // wgrad.hip; called by conv_kernels_init
// batch sums as order-independent integers (1) or fp64 atomics (0, default): see StatCell.  Synchronises the device.
int set_deterministic_sums(int on);

}  // namespace ocl
