// K1-K4: Reduced-ResNet18 convolution / batch-norm kernels for gfx950.
//
//  conv_t_kernel      implicit-GEMM 3x3 / 1x1 convolution on exact-fp32 MFMA (v_mfma_f32_16x16x4_f32), D[channel][pixel] tiles with
//                     K-grouped operands.  One generic "tap list + output lattice" geometry covers forward (stride 1/2), data
//                     gradient (stride 1; stride 2 as four parity classes, in one launch where the lattices coincide) and the
//                     1x1 shortcut.  The input patch (with halo) of a 64/128-pixel tile is staged ONCE in LDS and reused by
//                     all taps; weights are resident in LDS or stream through a double-buffered stage.  Epilogues from
//                     registers: BN batch statistics (fp64 atomics), folded eval-mode BN, residual, ReLU, masked residual.
//  (conv_wgrad_kernel, the weight gradient, lives in wgrad.hip; device helpers shared with it in conv_dev.h)
//  bn_*               train-mode BatchNorm forward (normalise+residual+ReLU, running-stat update) and backward.
//
// Replaces the ATen sequences behind models/resnet.py:10-12,32-37,90-99 and their autograd.
#include "conv_dev.h"
#include "conv_stats_dev.h"
#include <string.h>
#include <algorithm>
#include <type_traits>
#include <cmath>

namespace ocl {




// The host's copy of g_det_sums, PER DEVICE (conv_s_kernel's instantiation is chosen by it; the __constant__ lives per device, so a
// process-wide host flag could disagree with it as soon as a second device is touched: cells written as fixed point and read as doubles)
static const int kMaxDevices = 64;
static int g_det_dev[kMaxDevices] = {0};
static int det_host() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return 0;
    return g_det_dev[dev];
}
int set_deterministic_sums(int on) {
    const int v = on ? 1 : 0;
    int dev = 0;
    OCL_HIP(hipGetDevice(&dev));
    OCL_REQUIRE(dev >= 0 && dev < kMaxDevices, "set_deterministic: device %d", dev);
    OCL_HIP(hipDeviceSynchronize());   // (no launch may straddle the switch: the cells are interpreted by the flag)
    OCL_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_det_sums), &v, sizeof(int)));
    if (int rc = convw_set_det(v)) return rc;   // (convw.hip's copy of the flag)
    g_det_dev[dev] = v;                // the current device only: the mode is a per-device state, like the symbol
    return OCL_OK;
}


// =====================================================================================================
// conv_t_kernel: channels x pixels orientation with K-grouped operands
// =====================================================================================================
// D[channel][pixel] tiles: the MFMA's A operand is the weight (row = output channel), B the input patch (column = output pixel).
//  * K runs over (tap, channel quad) GROUPS q; a round of 4 MFMAs covers 4 groups, one per lane quarter g = lane >> 4, and MFMA
//    j of the round multiplies channel 4*c4(q_g) + j.  Any one-to-one assignment of k slots works as long as A and B agree, and
//    this one makes the 4 operands a lane needs for a round ONE 16-byte LDS read each: B from the pixel-major patch (channels
//    contiguous), A from the K-grouped pack [q][channel][4].  Per round a wave issues 1 + NT + MT LDS reads for 4*MT*NT MFMAs
//    (the round-1 kernel, pixels x channels tiles: 4*(MT+NT) 4-byte reads and their address arithmetic).
//  * A lane's 4 accumulator registers are 4 CONSECUTIVE output channels of one pixel: the epilogue (statistics, folded BatchNorm,
//    residual, mask, ReLU, accumulate) works on registers and stores 16-byte vectors straight to the NHWC tensor: no LDS
//    transpose, no barriers after the MFMAs.
//  * Weights of the small layers (<= kResidentBytes per channel split) are copied to LDS ONCE per persistent workgroup; the others
//    stream through a double-buffered stage of QS groups, fetched one stage ahead into registers.
//  * BatchNorm statistics: fp32 per-lane partials over the workgroup's tiles, fp64 from the cross-lane reduction on, flushed with
//    one fp64 atomic per channel per workgroup (8 replicas, as above).
constexpr int kWPF = 4;                        // float4 weight-prefetch registers per thread (staged weights)
constexpr size_t kResidentBytes = 80 * 1024;   // weights of one channel split kept in LDS for the workgroup's lifetime up to this

constexpr int kMaxWgTiles = 64;                // tile descriptors a workgroup keeps in LDS

// PIPE variant of the staged-weight path (the default since round 3; OCL_CONV_PIPE=0 / ConvGeomDesc::force_pipe = -1 select the
// two-buffer schedule).  Bit-identical to it on the whole network (tests/test_gpu_ring.py), 18 - 21 % faster per staged launch.  The two-buffer
// schedule pays, per stage and with one workgroup per CU, a serial section nothing overlaps: the table look-ups and loads of the
// next stage (4 dependent LDS round trips), the commit, a barrier and the first operand reads (~1900 of ~3800 cycles around 60
// MFMAs, profiles/r2_kbench_conv_staged_trace.txt).  Here the stages of a (tile, class, chunk) form ONE software-pipelined round
// sequence: weights go through a ring of three stage buffers, the registers hold the stage after next, and the stage's single
// barrier sits in the middle of its first round (after the commit of the next stage), so operand reads run across stage boundaries:
//    first round of stage t:  operand reads of round 1 | commit regs -> buffer (t+1)%3, look up the rows of stage t+2 |
//                             MFMAs of round 0 | loads of stage t+2 -> regs, barrier | ...
//  * buffer (t+1)%3 was last read in stage t-2, which every wave left before the barrier of stage t-1;
//  * stage t+1 is read after the barrier of stage t, which follows every wave's commit.
// Stage geometry by MT: QS groups with 256 * WPF == QS * 16 * MT units (every thread commits WPF whole units) and an even number
// of rounds per stage (the two operand register sets then alternate the same way in every stage).
#ifndef OCL_RING_SPREAD
#define OCL_RING_SPREAD 1
#endif
__host__ __device__ constexpr int pipe_qs(int MT) { return MT == 1 ? 64 : MT == 2 ? 32 : 16; }
__host__ __device__ constexpr int pipe_wpf(int MT) { return pipe_qs(MT) * 16 * MT / 256; }

// BNB: instantiated with the EPI_BNB epilogue (its registers must not weigh on the other launches: the forward instantiations sit at the
// edge of their occupancy step)
template <int MT, int NT, int PF, bool RES, bool CLS = false, bool PIPE = false, bool BNB = false>   // CLS: several output classes per tile (merged parity classes of a stride-2 data gradient)
__global__ void __launch_bounds__(256, PIPE ? 1 : 2) conv_t_kernel(const ConvArgs a) {   // PIPE plans run one workgroup per CU (three stage buffers): all 512 registers
    static_assert(!(PIPE && RES), "the ring is a schedule of the staged-weight path");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    constexpr int COPW = 16 * MT;              // channels per workgroup (one channel split)
    constexpr int WPF = RES ? 1 : (PIPE ? pipe_wpf(MT) : kWPF);   // float4 weight-prefetch registers per thread
    constexpr int QSP = pipe_qs(MT);           // PIPE: groups per stage (== a.QS)
    int* tdesc = (int*)lds_raw;                // [kMaxWgTiles][8] per-tile geometry of this workgroup's tile range
    int* ctab = tdesc + kMaxWgTiles * 8;       // [4][4] per output class: first group, groups (padded to rounds), output offset, weight stages
    int* qoff = ctab + 16;                     // [Qpad] patch offset (floats) of group q relative to a pixel's origin
    int* qrow = qoff + a.Qpad;                 // [Qpad] row of the K-grouped pack (tap * C4tot + channel quad), -1: padding group
    float* wl = (float*)(qrow + a.Qpad);       // resident: [Qpad][COPW][4]; staged: [2][QS][COPW][4]; PIPE: [3][QS][COPW][4]
    float* patch = wl + (size_t)(RES ? a.Qpad : (PIPE ? 3 : 2) * a.QS) * COPW * 4;   // [imgs][PR][PC][CP]
    float* xft = patch + a.patch_floats;       // input transform: [groups][Cin/4][2][4] scale quads / shift quads
    const float* bnt = xft + (a.bnb_lds > 0 ? a.bnb_lds : 0);   // EPI_BNB: [groups][Cout/4][3][4] scale, shift, mean quads of the BatchNorm being differentiated

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.y * COPW;
    const int LP = a.LH * a.LW;
    const int ntiles_all = a.groups * a.tiles_per_group;
    // contiguous tile range of this workgroup: neighbouring tiles share halo rows (L2) and one BatchNorm group
    const int t_begin = (int)(((int64_t)blockIdx.x * ntiles_all) / gridDim.x), t_end = (int)(((int64_t)(blockIdx.x + 1) * ntiles_all) / gridDim.x);
    const int nwt = t_end - t_begin;
    if (nwt <= 0) return;
    const int flags = BNB ? a.flags : (a.flags & ~EPI_BNB);
    int tr_n = 0;
    auto stamp = [&]() __attribute__((always_inline)) {
        if (a.trace && tid == 0 && tr_n < 64) a.trace[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 64 + tr_n++] = __builtin_amdgcn_s_memtime();
    };
    stamp();   // 0: start
    // ---- tables ------------------------------------------------------------------------------------------------------------------
    // Everything that depends only on the plan -- the K-group tables, the geometry of every tile, every thread's patch units and
    // output pixels -- is computed ONCE on the host when the plan is made (conv_plan_tables) and sits in device memory next to the
    // plan: the prologue is a handful of independent loads instead of ~8 k cycles of integer arithmetic, dependent LDS round trips
    // and kernel-argument fetches per launch (profiles/r3_kbench_conv_220_trace.txt; rounds 1 - 2 built them here, per workgroup).
    const int ncls = CLS ? (a.cls_pack & 15) : 1;
    const int* __restrict__ blob = a.blob;
    int pu_goff[PF], pu_lds[PF], pu_rp[PF];   // per-thread patch units (float4 along the channels): global byte offset from the patch origin; LDS float offset; row | pr << 16
    {
        const int* pu = blob + a.off_pu + tid;
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            pu_goff[i] = pu[(3 * i + 0) * 256];
            pu_lds[i] = pu[(3 * i + 1) * 256];
            pu_rp[i] = pu[(3 * i + 2) * 256];
        }
    }
    // the lane's NT pixels relative to the tile origin (aligned plans: tile-invariant)
    int loc_p[NT], loc_o[NT], loc_il[NT];
    {
        const int* lc = blob + a.off_loc + tid;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            loc_p[nt] = lc[(3 * nt + 0) * 256];
            loc_o[nt] = lc[(3 * nt + 1) * 256];
            loc_il[nt] = lc[(3 * nt + 2) * 256];
        }
    }
    const int4 tile0 = *(const int4*)(blob + a.off_tdesc + (size_t)t_begin * 8);   // first tile: in_base, iy0, nrows, obase (block-uniform)
    // class table + group tables (contiguous in the blob and in LDS: 16 + 2 * Qpad <= 768 ints, checked by the planner) and this
    // workgroup's tile descriptors (<= kMaxWgTiles * 8 = 512 ints): predicated loads, requested BEFORE the first patch (loads return in order: the stores
    // below then wait for the tables only, not for the patch)
    const int ntab = 16 + 2 * a.Qpad, ntd = nwt * 8;
    const int* td = blob + a.off_tdesc + (size_t)t_begin * 8;
    const int tab0 = tid < ntab ? blob[tid] : 0, tab1 = tid + 256 < ntab ? blob[tid + 256] : 0, tab2 = tid + 512 < ntab ? blob[tid + 512] : 0;
    const int td0 = tid < ntd ? td[tid] : 0, td1 = tid + 256 < ntd ? td[tid + 256] : 0;
    const __amdgpu_buffer_rsrc_t rs_in = make_rsrc(a.in), rs_w = make_rsrc(a.wT);
    float4 pv[PF];
    unsigned okm = 0;   // bit i: unit i of the patch in flight lies inside the image (input transform: the others stay zero)
    auto load_patch_d = [&](const int4 d, int c0) __attribute__((always_inline)) {   // d: in_base, iy0, nrows, obase
        const int base = d.x + c0 * 4;
        okm = 0;
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            const int row = pu_rp[i] & 0xffff, pr = (pu_rp[i] >> 16) & 0xff;
            const bool ok = (row < d.z) & ((unsigned)(d.y + pr) < (unsigned)a.Hin) & (pu_goff[i] >= 0);
            pv[i] = buf_load16(rs_in, ok ? base + pu_goff[i] : kOob);
            okm |= ok ? (1u << i) : 0u;
        }
    };
    // grp / c0: BatchNorm group of the tile and channel origin of the chunk being stored (input transform only)
    auto store_patch = [&](int nrows, int grp, int c0) __attribute__((always_inline)) {
        if (a.xf) {   // block-uniform
            const float* tb = xft + (size_t)(grp * a.C4tot + (c0 >> 2)) * 8;
#pragma unroll
            for (int i = 0; i < PF; ++i)
                if ((pu_rp[i] & 0xffff) < nrows) {
                    const float* t = tb + (pu_rp[i] >> 24) * 8;
                    const float4 sc = *(const float4*)t, sh = *(const float4*)(t + 4);
                    float4 v = pv[i];
                    v.x = fmaxf(__fmaf_rn(v.x, sc.x, sh.x), 0.f); v.y = fmaxf(__fmaf_rn(v.y, sc.y, sh.y), 0.f);
                    v.z = fmaxf(__fmaf_rn(v.z, sc.z, sh.z), 0.f); v.w = fmaxf(__fmaf_rn(v.w, sc.w, sh.w), 0.f);
                    if (!((okm >> i) & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f);
                    *(float4*)(patch + pu_lds[i]) = v;
                }
            return;
        }
#pragma unroll
        for (int i = 0; i < PF; ++i)
            if ((pu_rp[i] & 0xffff) < nrows) {   // CP % 4 == 0: 16-byte aligned
                *(float4*)(patch + pu_lds[i]) = pv[i];
            }
    };
    auto load_patch = [&](int k, int c0) __attribute__((always_inline)) { load_patch_d(*(const int4*)(tdesc + k * 8), c0); };
    load_patch_d(tile0, 0);
    if (a.xf) {   // the producer's BatchNorm folded into scale / shift per (group, channel); see ConvArgs::xf
        const int C = a.Cin;
        const double M = (double)a.xf_m_per_group;
        const bool lead = blockIdx.x == 0 && blockIdx.y == 0;
        for (int j = tid; j < a.groups * C; j += 256) {
            const int gq = j / C, c = j - gq * C;
            double mean, var;
            bn_batch_moments(a.xf_stats, a.xf_rep_stride, gq, c, C, M, a.xf_eps, mean, var);
            // 1 / sqrt(var + eps) without the fp64 divide / square-root sequences (every workgroup of the launch runs this prologue):
            // fp32 rsqrt seed + two Newton steps in fp64 (relative error < 1e-15: the float it is rounded to is the exact one)
            const double xv = var + (double)a.xf_eps;
            double invstd = (double)rsqrtf((float)xv);
            invstd = invstd * (1.5 - 0.5 * xv * invstd * invstd);
            invstd = invstd * (1.5 - 0.5 * xv * invstd * invstd);
            float sc, sh;
            bn_scale_shift(a.xf_gamma[c], a.xf_beta[c], (float)mean, (float)invstd, sc, sh);
            float* t = xft + (size_t)(gq * (C >> 2) + (c >> 2)) * 8 + (c & 3);
            t[0] = sc;
            t[4] = sh;
            if (lead) {
                a.xf_save_mean[j] = (float)mean;
                a.xf_save_invstd[j] = (float)invstd;
            }
        }
        if (lead && a.xf_running_mean)
            bn_running_update(a.xf_stats, a.xf_rep_stride, a.groups, C, M, a.xf_momentum, a.xf_eps, a.xf_running_mean, a.xf_running_var, a.xf_nbt, tid, 256);
    }
    if (BNB && (flags & EPI_BNB)) bnb_table(a, const_cast<float*>(bnt), tid, 256);
    if (tid < ntab) ctab[tid] = tab0;
    if (tid + 256 < ntab) ctab[tid + 256] = tab1;
    if (tid + 512 < ntab) ctab[tid + 512] = tab2;
    if (tid < ntd) tdesc[tid] = td0;
    if (tid + 256 < ntd) tdesc[tid + 256] = td1;

    stamp();   // P1: tables written, first patch requested
    __syncthreads();   // tables visible
    stamp();   // P2: barrier
    // ---- weights ----------------------------------------------------------------------------------------------------------
    const int wcol_ok = a.WPT - n0;   // columns of this split that exist in the pack
    if (RES) {
        // global -> LDS without registers (buffer_load ... lds): a wave instruction fills 64 consecutive 16-byte units (LDS address =
        // wave-uniform base + lane * 16, global address per lane); everything is in flight at once, one wait at the end.  Padding
        // groups / channels past the pack address the descriptor's out-of-range area, which reads as zeros.
        const int units = a.Qpad * COPW;
#pragma unroll 4
        for (int u0 = wave * 64; u0 < units; u0 += 256) {
            const int u = u0 + lane;
            const int q = min(u, units - 1) / COPW, c = min(u, units - 1) - q * COPW;
            const int row = qrow[q];
            const int off = (u < units && row >= 0 && c < wcol_ok) ? ((row * a.WPT + n0 + c) * 4) * 4 : kOob;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(wl + (size_t)u0 * 4), 16, off, 0, 0, 0);
        }
    }
    stamp();   // P3: weight DMA issued
    // staged: stage s of chunk c0 covers groups [s*QS, s*QS + QS); unit u = tid + i*256 -> (group in stage, channel)
    float4 wv[WPF];
    auto w_prefetch = [&](int s_, int c0_, int cls) __attribute__((always_inline)) {
        const int q0 = (CLS ? ctab[cls * 4] : 0) + s_ * a.QS, qend = CLS ? ctab[cls * 4] + ctab[cls * 4 + 1] : a.Qpad;
        const int c4base = c0_ >> 2;
#pragma unroll
        for (int i = 0; i < WPF; ++i) {
            const int u = tid + i * 256;
            const int qq = u / COPW, c = u - qq * COPW;
            const int q = q0 + qq;
            const int row = (qq < a.QS && q < qend) ? qrow[q] : -1;
            wv[i] = buf_load16(rs_w, (row >= 0 && c < wcol_ok) ? (((row + c4base) * a.WPT + n0 + c) * 4) * 4 : kOob);
        }
    };
    auto w_commit = [&](int buf) __attribute__((always_inline)) {
        float* dst = wl + (size_t)buf * a.QS * COPW * 4;
#pragma unroll
        for (int i = 0; i < WPF; ++i) {
            const int u = tid + i * 256;
            if (u < a.QS * COPW) *(float4*)(dst + (size_t)u * 4) = wv[i];
        }
    };
    // ---- PIPE: the prefetch cursor runs two stages ahead of the MFMAs (stage in class-chunk, chunk origin, class, tile; the class's
    // first group / group count / stage count); the look-up, the loads and the commit are separate steps so that each sits where its
    // latency is covered (see the schedule above).  Past the workgroup's last stage the cursor simply wraps to the first tile's stages
    // (two stages of loads nobody reads).
    int pf_s = 0, pf_c0 = 0, pf_cls = 0, pf_q0 = 0, pf_nq = a.Qpad, pf_nst = a.nstage;
    int xb = 0;                                // ring buffer of the stage whose MFMAs issue
    // The look-up only READS the table (its consumers come after a round of MFMAs: no wait in between).  No bounds beyond the table's:
    // groups past the class's last one (partial last stage) or past the workgroup's last stage fetch rows no MFMA reads.
    int prow[PIPE ? WPF : 1];
    auto pf_lookup = [&]() __attribute__((always_inline)) {
        const int qs0 = pf_q0 + pf_s * QSP;
#pragma unroll
        for (int i = 0; i < (PIPE ? WPF : 1); ++i) prow[i] = qrow[min(qs0 + (tid + i * 256) / COPW, a.Qpad - 1)];
    };
    auto pf_issue = [&]() __attribute__((always_inline)) {
        const int cb = (pf_c0 >> 2) * a.WPT * 16;   // chunk origin in the pack, bytes
#pragma unroll
        for (int i = 0; i < (PIPE ? WPF : 1); ++i) {
            const int u = tid + i * 256;
            const int c = u - (u / COPW) * COPW;
            wv[i] = buf_load16(rs_w, (prow[i] + cb + (n0 + c) * 16) | (c < wcol_ok ? 0 : (int)0x80000000));
        }
        if (++pf_s >= pf_nst) {   // block-uniform
            pf_s = 0;
            pf_c0 += a.KC;
            if (pf_c0 >= a.Cin) {
                pf_c0 = 0;
                if (CLS) {   // next class, or the first class of the next tile
                    if (++pf_cls >= ncls) pf_cls = 0;
                    pf_q0 = __builtin_amdgcn_readfirstlane(ctab[pf_cls * 4]);
                    pf_nq = __builtin_amdgcn_readfirstlane(ctab[pf_cls * 4 + 1]);
                    pf_nst = (pf_nq + QSP - 1) / QSP;
                }
            }
        }
    };
    auto pf_commit = [&](int buf) __attribute__((always_inline)) {   // 256 * WPF == QSP * COPW: every unit exists
        float* dst = wl + (size_t)buf * QSP * COPW * 4;
#pragma unroll
        for (int i = 0; i < (PIPE ? WPF : 1); ++i) *(float4*)(dst + (size_t)(tid + i * 256) * 4) = wv[i];
    };
    if (PIPE) {   // stage 0 is requested here: its latency runs under the per-lane set-up below
        if (CLS) {
            pf_q0 = __builtin_amdgcn_readfirstlane(ctab[0]);
            pf_nq = __builtin_amdgcn_readfirstlane(ctab[1]);
            pf_nst = (pf_nq + QSP - 1) / QSP;
        }
        pf_lookup();
        pf_issue();
    }

    const int nchunks = a.Cin / a.KC;
    float s1[MT][4], s2[MT][4];   // BatchNorm partial sums of this lane's channels over this workgroup's tiles
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int e = 0; e < 4; ++e) s1[mt][e] = s2[mt][e] = 0.f;
    // PIPE (one workgroup per CU: the AccVGPR half of the register file is free): the statistics partials sit in AccVGPRs while a
    // tile's MFMA sequence runs -- 8*MT ArchVGPRs fewer live across the loop, which is what lets the register allocator keep the two
    // operand sets in place instead of squeezing temporaries into them (copies + early waits: profiles/r2_kbench_ring_trace.txt)
    constexpr bool PARK = PIPE && !CLS;
    float park[PARK ? 8 * MT : 1];
    auto park_stats = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(park[PARK ? (mt * 4 + e) * 2 : 0]) : "v"(s1[mt][e]));
                asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(park[PARK ? (mt * 4 + e) * 2 + 1 : 0]) : "v"(s2[mt][e]));
            }
    };
    auto unpark_stats = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(s1[mt][e]) : "a"(park[PARK ? (mt * 4 + e) * 2 : 0]));
                asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(s2[mt][e]) : "a"(park[PARK ? (mt * 4 + e) * 2 + 1 : 0]));
            }
    };
    int run_grp = -1;
    auto flush_stats = [&]() __attribute__((always_inline)) {
        // lanes with the same g hold the same channels for 16 different pixels: fp32 butterfly over them (a lane's partial covers at
        // most a few dozen values), then fp64: the 4 waves through LDS (`patch` is free here: a barrier precedes), one atomic per channel
        double* red = (double*)patch;   // [4 waves][2][COPW]
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x = row16_sum(s1[mt][e]), y = row16_sum(s2[mt][e]);
                if (r16 == 0) {
                    red[(wave * 2 + 0) * COPW + mt * 16 + 4 * g + e] = (double)x;
                    red[(wave * 2 + 1) * COPW + mt * 16 + 4 * g + e] = (double)y;
                }
                s1[mt][e] = s2[mt][e] = 0.f;
            }
        __syncthreads();
        if (tid < 2 * COPW && run_grp >= 0) {
            const int which = tid / COPW, c = tid - which * COPW;
            const int co = n0 + c;
            if (co < a.Cout) {
                const double v = (red[(0 * 2 + which) * COPW + c] + red[(1 * 2 + which) * COPW + c]) +
                                 (red[(2 * 2 + which) * COPW + c] + red[(3 * 2 + which) * COPW + c]);
                StatCell* st_ = a.stats + (int64_t)(blockIdx.x % kStatReps) * a.stat_rep_stride;
                fx_add(&st_[((int64_t)run_grp * 2 + which) * a.Cout + co], v);
            }
        }
        __syncthreads();
    };

    int st = 0;
    if (PIPE) {   // stage 0 into buffer 0 (published by the barriers of the first tile), stage 1 into the registers
        pf_commit(0);
        pf_lookup();
        pf_issue();
    } else if (!RES) {
        w_prefetch(0, 0, 0);
    }
    // the resident weights (LDS-DMA) were in flight during the per-lane set-up above; every wave waits for ITS OWN DMA writes here
    // (a barrier does not wait for vector-memory operations), the barriers of the first tile publish them
    if (RES) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp();   // P5: set-up done
    for (int k = 0; k < nwt; ++k) {
        const int4 d0 = *(const int4*)(tdesc + k * 8);       // in_base, iy0, nrows, obase
        const int4 d1 = *(const int4*)(tdesc + k * 8 + 4);   // nimg, grp, p0, img0 | ly0 << 20
        if ((flags & (EPI_STATS | EPI_BNB)) && d1.y != run_grp) {   // block-uniform; the tile range is in ascending group order
            if (run_grp >= 0) flush_stats();
            run_grp = d1.y;
        }
        // this lane's NT output pixels: LDS patch offset of the pixel's origin, output element offset (-1: not a pixel)
        int pbase[NT], ooff[NT];
        if (a.aligned) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const bool v = loc_il[nt] < d1.x;
                pbase[nt] = v ? loc_p[nt] : 0;
                ooff[nt] = v ? d0.w + loc_o[nt] : -1;
            }
        } else {
            const int img0 = d1.w & 0xfffff, ly0 = d1.w >> 20;
            const int grp_end = min(a.N, (d1.y + 1) * a.group_size);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int r = wave * 16 * NT + nt * 16 + r16;
                int pl, lx;
                const int il = mdiv(r, a.m_ppi, a.ppi, pl);
                const int p = d1.z + pl;
                const int n = img0 + il;
                const bool v = (il < a.imgs) & (n < grp_end) & (p < LP);
                const int ly = mdiv(p, a.m_lw, a.LW, lx);
                pbase[nt] = v ? ((il * a.PR + (ly - ly0) * a.is) * a.PC + lx * a.is) * a.CP : 0;
                ooff[nt] = v ? ((n * a.Hout + ly * a.os + a.oy0) * a.Wout + lx * a.os + a.ox0) * a.Cout : -1;
            }
        }
        f32x4 acc[MT][NT];

        // operands of round rho+1 are read from LDS while the MFMAs of round rho issue (two register sets).  What the ring's loop taught
        //  applies here too: the patch-offset table entry of a fetch is read TWO fetches ahead (its wait never falls on
        // reads that have just been issued -- the round-2 loop waited for the entry right behind its ds_read, an exposed LDS round trip
        // per round pair), the operand reads are unconditional (past the last round they fetch registers nobody uses, from addresses
        // inside the weight / patch area) so that a round pair is ONE straight-line body, and sched_barriers keep every read in front
        // of the MFMAs whose register set it does not touch.
        auto rounds = [&](const float* wbase, int q0, int nq) __attribute__((always_inline)) {
            const float* wb = wbase + (size_t)(g * COPW + r16) * 4;
            const int nr = nq >> 2;
            float4 bv[2][NT], av[2][MT];
            int fR = 0;
            int po = qoff[q0 + g], po1 = qoff[q0 + 4 * min(1, nr - 1) + g];
            auto fetch = [&](int set) __attribute__((always_inline)) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bv[set][nt] = *(const float4*)(patch + pbase[nt] + po);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) av[set][mt] = *(const float4*)(wb + (size_t)fR * 4 * COPW * 4 + mt * 64);
                ++fR;
                po = po1;
                po1 = qoff[q0 + 4 * min(fR + 1, nr - 1) + g];
            };
            auto fma4 = [&](int set) __attribute__((always_inline)) {   // k component outermost: consecutive MFMAs accumulate into different tiles
#define OCL_KSTEP(E)                                                                                                              \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                           \
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[set][mt].E, bv[set][nt].E, acc[mt][nt], 0, 0, 0);
                OCL_KSTEP(x) OCL_KSTEP(y) OCL_KSTEP(z) OCL_KSTEP(w)
#undef OCL_KSTEP
            };
            fetch(0);
            int rho = 0;
            for (; rho + 2 <= nr; rho += 2) {
                fetch(1);
                __builtin_amdgcn_sched_barrier(0);
                fma4(0);
                __builtin_amdgcn_sched_barrier(0);
                fetch(0);
                __builtin_amdgcn_sched_barrier(0);
                fma4(1);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (rho < nr) fma4(0);
        };
        // PIPE: the nrs rounds of one (class, chunk) as ONE pipelined sequence over its weight stages (ring buffers xb, xb+1, ...)
        auto seq = [&](int q0, int nrs) __attribute__((always_inline)) {
            constexpr int RPS = QSP / 4;                 // rounds per stage (even)
            const float* wlane = wl + (size_t)(g * COPW + r16) * 4;
            float4 bv[2][NT], av[2][MT];
            int fR = 0, fr = 0, fb = xb;                 // fetch cursor: round of the sequence, round of its stage, ring buffer
            // patch offsets of the next two fetches: a table entry is consumed two fetches (one loop iteration, 2 x 4*MT*NT MFMAs) after it
            // is read, so the wait in front of its address arithmetic never falls on reads that have just been issued
            int po = qoff[q0 + g], po1 = qoff[q0 + 4 * min(1, nrs - 1) + g];
            auto fetch = [&](int set) __attribute__((always_inline)) {
                const float* wb = wlane + (size_t)fb * (QSP * COPW * 4) + fr * (16 * COPW);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bv[set][nt] = *(const float4*)(patch + pbase[nt] + po);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) av[set][mt] = *(const float4*)(wb + mt * 64);
                ++fR;
                po = po1;
                po1 = qoff[q0 + 4 * min(fR + 1, nrs - 1) + g];
                if (++fr == RPS) { fr = 0; fb = fb == 2 ? 0 : fb + 1; }
            };
            // k component outermost: consecutive MFMAs accumulate into different tiles
            auto fma4 = [&](int set) __attribute__((always_inline)) {
#define OCL_KSTEP(E, F)                                                                                                           \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                           \
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[set][mt].E, bv[set][nt].F, acc[mt][nt], 0, 0, 0);
                OCL_KSTEP(x, x) OCL_KSTEP(y, y) OCL_KSTEP(z, z) OCL_KSTEP(w, w)
#undef OCL_KSTEP
            };
            // (operand reads are unconditional: past the sequence's last round they fetch registers nobody uses, from addresses inside the
            // ring and the patch.  The sched_barriers keep every read where it is written: hoisted into MFMAs that still read the
            // register set it refills, a read gets other registers and a copy -- with an early wait -- behind it.)
            // One wave per SIMD: every instruction that is not an MFMA costs the MFMA stream an issue slot unless it falls into the
            // 32-cycle shadow of an MFMA (about four per gap, cdna guide: issue slots).  The stage's bookkeeping is ~45 instructions
            // (commit, table look-ups) plus ~40 (addresses, loads, cursor): left to the scheduler they form two bursts in front of the
            // first MFMAs of each round (ISA of round 2's build: 45 instructions inside the first k-step of round 0) and the MFMA pipe
            // starves for ~900 cycles per stage (profiles/r3_kbench_conv_220_trace.txt: 44.9 cycles per MFMA against 33.8).  The
            // group barriers below spread them: after every MFMA of the round at most kFill other instructions.
            constexpr int kFillMask = 0x002 | 0x004 | 0x010 | 0x080;   // VALU | SALU | VMEM | DS
            auto spread = [&](int fill) __attribute__((always_inline)) {
#pragma unroll
                for (int i = 0; i < 4 * MT * NT; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (fill == 2) __builtin_amdgcn_sched_group_barrier(kFillMask, 2, 0);
                    else __builtin_amdgcn_sched_group_barrier(kFillMask, 3, 0);
                }
            };
            auto first_pair = [&]() __attribute__((always_inline)) {   // rounds 0, 1 of a stage, with the stage's bookkeeping
                fetch(1);
                pf_commit(xb == 2 ? 0 : xb + 1);
                pf_lookup();
                fma4(0);
                if (OCL_RING_SPREAD) spread(3);
                __builtin_amdgcn_sched_barrier(0);       // the loads (and their table values) stay behind the first round's MFMAs
                fetch(0);
                __builtin_amdgcn_sched_barrier(0);       // operand reads first: they have the whole second round to land
                pf_issue();
                fma4(1);
                if (OCL_RING_SPREAD) spread(2);
                __builtin_amdgcn_sched_barrier(0);       // (the barrier is not hoisted into the MFMAs: its wait would cover the reads above)
                __syncthreads();                         // before the first read of stage t+1 (last round pair of this stage)
            };
            // (a variant with each round's reads split into three pieces between the k-steps of the round before -- at most three LDS
            // instructions per gap -- measured the same: profiles/r2_kbench_ring_v3.txt; the simpler form is kept)
            auto pair = [&]() __attribute__((always_inline)) {
                fetch(1);
                __builtin_amdgcn_sched_barrier(0);
                fma4(0);
                __builtin_amdgcn_sched_barrier(0);
                fetch(0);
                __builtin_amdgcn_sched_barrier(0);
                fma4(1);
                __builtin_amdgcn_sched_barrier(0);
            };
            fetch(0);
            // whole stages: ONE straight-line loop body (RPS rounds), so the two operand sets keep their registers around the back edge
            const int nfull = nrs / RPS;
            for (int t = 0; t < nfull; ++t) {
                first_pair();
#pragma unroll
                for (int p = 1; p < RPS / 2; ++p) pair();
                xb = xb == 2 ? 0 : xb + 1;
            }
            // the class-chunk's last, partial stage (fewer than RPS rounds)
            const int rem = nrs - nfull * RPS;
            if (rem > 0) {
                int R = 0;
                if (rem >= 2) {
                    first_pair();
                    for (R = 2; R + 2 <= rem; R += 2) pair();
                }
                if (R < rem) {   // odd last round
                    if (R == 0) {
                        pf_commit(xb == 2 ? 0 : xb + 1);
                        pf_lookup();
                        fma4(0);
                        __builtin_amdgcn_sched_barrier(0);
                        pf_issue();
                        __syncthreads();
                    } else {
                        fma4(0);
                    }
                }
                xb = xb == 2 ? 0 : xb + 1;
            }
        };

        // output classes (one for an ordinary convolution): with a single channel chunk they share the tile's patch; with several
        // chunks every (class, chunk) stages its own
        if (PARK) park_stats();
        for (int cls = 0; cls < ncls; ++cls) {
        const int4 ct = CLS ? *(const int4*)(ctab + cls * 4) : make_int4(0, a.Qpad, 0, a.nstage);   // first group, groups, output offset, weight stages
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int chunk = 0; chunk < nchunks; ++chunk) {
            const int c0 = chunk * a.KC;
            const bool fresh = (nchunks > 1) | (cls == 0);   // block-uniform
            if (fresh) {
                stamp();   // tile + 0: tile set-up done
                __syncthreads();   // consumers of the previous patch are done
                stamp();   // tile + 1: barrier passed
                store_patch(d0.z, d1.y, c0);
                stamp();   // tile + 2: patch arrived and written to LDS
                if (chunk + 1 < nchunks) load_patch(k, c0 + a.KC);
                else if (nchunks > 1 && cls + 1 < ncls) load_patch(k, 0);
                else if (k + 1 < nwt) load_patch(k + 1, 0);
            }
            if (RES) {
                if (fresh) {
                    __syncthreads();   // patch (and, the first time, the resident weights) visible
                    stamp();   // tile + 3: second barrier passed
                }
                rounds(wl + (size_t)ct.x * COPW * 4, ct.x, ct.y);
                if (fresh) stamp();   // tile + 4: MFMAs issued
            } else if (PIPE) {
                if (fresh) {
                    __syncthreads();   // patch visible (a stage's weights: published by the barrier that follows their commit)
                    stamp();
                }
                seq(ct.x, ct.y >> 2);
                stamp();   // (ring) MFMAs of the class-chunk issued
            } else {
                for (int s_ = 0; s_ < ct.w; ++s_, ++st) {
                    w_commit(st & 1);
                    stamp();   // (staged) weights of the stage arrived and written
                    __syncthreads();   // stage st's weights (and the patch) visible; everyone is done with stage st-1
                    stamp();   // (staged) barrier passed
                    {   // the stage after this one: next stage of the class, next chunk, next class, next tile
                        int ns = s_ + 1, nc0 = c0, ncl = cls, nk = k;
                        if (ns >= ct.w) {
                            ns = 0; nc0 = c0 + a.KC;
                            if (nc0 >= a.Cin) {
                                nc0 = 0; ncl = cls + 1;
                                if (ncl >= ncls) { ncl = 0; nk = k + 1; }
                            }
                        }
                        if (nk < nwt) w_prefetch(ns, nc0, ncl);
                    }
                    const int q0 = ct.x + s_ * a.QS;
                    rounds(wl + (size_t)(st & 1) * a.QS * COPW * 4, q0, min(a.QS, ct.x + ct.y - q0));
                    stamp();   // (staged) MFMAs of the stage issued
                }
            }
        }

        if (PARK) unpark_stats();
        // ---- epilogue from registers: lane (r16 = pixel, g) holds channels n0 + mt*16 + 4g .. +3 of its NT pixels -----------------
        // the two flag sets of a training step (forward: statistics only; plain data gradient: nothing) run without per-store branches
        if (flags == EPI_STATS || flags == 0) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const bool pv_ok = ooff[nt] >= 0;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int co = n0 + mt * 16 + 4 * g;
                    if (pv_ok && co < a.Cout) {
                        const float4 v = make_float4(acc[mt][nt][0], acc[mt][nt][1], acc[mt][nt][2], acc[mt][nt][3]);
                        s1[mt][0] += v.x; s1[mt][1] += v.y; s1[mt][2] += v.z; s1[mt][3] += v.w;
                        s2[mt][0] = fmaf(v.x, v.x, s2[mt][0]); s2[mt][1] = fmaf(v.y, v.y, s2[mt][1]);
                        s2[mt][2] = fmaf(v.z, v.z, s2[mt][2]); s2[mt][3] = fmaf(v.w, v.w, s2[mt][3]);
                        *(float4*)(a.out + (int64_t)ooff[nt] + ct.z + co) = v;
                    }
                }
            }
        } else
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const bool pv_ok = ooff[nt] >= 0;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int co = n0 + mt * 16 + 4 * g;
                if (!pv_ok || co >= a.Cout) continue;
                float4 v = make_float4(acc[mt][nt][0], acc[mt][nt][1], acc[mt][nt][2], acc[mt][nt][3]);
                if (flags & EPI_STATS) {
                    s1[mt][0] += v.x; s1[mt][1] += v.y; s1[mt][2] += v.z; s1[mt][3] += v.w;
                    s2[mt][0] = fmaf(v.x, v.x, s2[mt][0]); s2[mt][1] = fmaf(v.y, v.y, s2[mt][1]);
                    s2[mt][2] = fmaf(v.z, v.z, s2[mt][2]); s2[mt][3] = fmaf(v.w, v.w, s2[mt][3]);
                }
                float* op = a.out + (int64_t)ooff[nt] + ct.z + co;
                if (flags & EPI_AFFINE) {
                    const float4 sc = *(const float4*)(a.scale + co), sh = *(const float4*)(a.shift + co);
                    v.x = fmaf(v.x, sc.x, sh.x); v.y = fmaf(v.y, sc.y, sh.y); v.z = fmaf(v.z, sc.z, sh.z); v.w = fmaf(v.w, sc.w, sh.w);
                }
                if (flags & EPI_RES) {
                    const float4 r = *(const float4*)(a.res + (int64_t)ooff[nt] + ct.z + co);
                    v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
                }
                if (flags & EPI_RESMASK) {
                    const float4 r = *(const float4*)(a.res + (int64_t)ooff[nt] + ct.z + co);
                    const float4 m = *(const float4*)(a.resmask + (int64_t)ooff[nt] + ct.z + co);
                    v.x += m.x > 0.f ? r.x : 0.f; v.y += m.y > 0.f ? r.y : 0.f; v.z += m.z > 0.f ? r.z : 0.f; v.w += m.w > 0.f ? r.w : 0.f;
                }
                if (flags & EPI_ACCUM) {
                    const float4 o = *(const float4*)op;
                    v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                }
                if (BNB && (flags & EPI_BNB)) {   // ReLU mask + the two batch sums of the BatchNorm this gradient enters (ConvArgs::bnb_*)
                    const int64_t eo = (int64_t)ooff[nt] + ct.z + co;
                    const float* t = bnt + (size_t)(d1.y * (a.Cout >> 2) + (co >> 2)) * 12;
                    bnb_apply(a, *(const float4*)t, *(const float4*)(t + 4), *(const float4*)(t + 8), eo, v, s1[mt], s2[mt]);
                }
                if (flags & EPI_RELU) {
                    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                }
                *(float4*)op = v;
            }
        }
        }   // classes
        stamp();   // tile + 5: epilogue issued
    }
    if ((flags & (EPI_STATS | EPI_BNB)) && run_grp >= 0) flush_stats();
    stamp();
}


// =====================================================================================================
// conv_q_kernel: the convolutions with at most 20 output channels (stem, layer 1, their data gradients) on v_mfma_f32_4x4x1_16b_f32
// =====================================================================================================
// A 16x16x4 tile pads 20 output channels to 32 and 45 (tap, channel-quad) groups to 48: 41 % of the MFMAs issued by conv_t_kernel on
// layer 1 multiply zeros, and layer 1 is the largest single item of a replay step (8 launches, 28 % of the convolution time).  The
// 4x4x1 form is sixteen independent 4x4 outer products per instruction at the same MACs per cycle (profiles/r3_mfma_4x4x1_calibration.txt:
// 10.5 - 12 cycles against 8 ideal with this kernel's operand traffic):
//   block b = 4 consecutive pixels of the wave's 64-pixel set;  B: lane L supplies ITS pixel's input value x[pixel L][k];
//   A: lane L supplies w[channel 4m + (L & 3)][k] (every block multiplies the same four channels);  D: register i of lane L is
//   output channel 4m + i of pixel L.
// So a lane owns one pixel per set and, per block m of four channels, the same "4 consecutive channels of one pixel" accumulator
// layout as conv_t_kernel: the register epilogue carries over.  Nothing is padded: K runs over the 45 groups themselves (one group =
// one 16-byte read of the lane's pixel + 5 broadcast reads of the weights' k-quads for 4 * 5 * NTQ MFMAs), channels over 5 blocks.
// The operand traffic per MFMA is what limits the form (the weights are re-read per 64-pixel set), hence NTQ >= 2 sets per wave and one
// workgroups per CU kept at two by LDS and registers.  Weights are always resident (<= 14.4 KB); tables, patch staging, input transform and epilogue flags as in
// conv_t_kernel.
// TRACE = 1 (measurement build, launched when ConvArgs::trace is set: kbench KBENCH_TRACE): s_memtime stamps of thread 0 -- start |
// tables + weight DMA + first patch landed | per tile: passed barrier 1, patch stored + next patch requested + passed barrier 2, K loop
// done, epilogue done | statistics flushed.
template <int NTQ, int PF, int STATS, int TRACE = 0>   // STATS 0: no sums; 1: forward batch statistics (EPI_STATS); 2: BatchNorm-backward sums (EPI_BNB)
__global__ void __launch_bounds__(256, 2) conv_q_kernel(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    constexpr int MB = kQBlocks, COPW = 4 * MB;
    int* tdesc = (int*)lds_raw;
    int* ctab = tdesc + kMaxWgTiles * 8;
    int* qoff = ctab + 16;
    int* qrow = qoff + a.Qpad;
    float* wl = (float*)(qrow + a.Qpad);                  // [Qpad][COPW][4]
    float* patch = wl + (size_t)a.Qpad * COPW * 4;
    float* xft = patch + a.patch_floats;
    const float* bnt = xft + (a.bnb_lds > 0 ? a.bnb_lds : 0);   // EPI_BNB table (see conv_t_kernel)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int LP = a.LH * a.LW;
    const int ntiles_all = a.groups * a.tiles_per_group;
    const int t_begin = (int)(((int64_t)blockIdx.x * ntiles_all) / gridDim.x), t_end = (int)(((int64_t)(blockIdx.x + 1) * ntiles_all) / gridDim.x);
    const int nwt = t_end - t_begin;
    if (nwt <= 0) return;
    int tr_n = 0;
    auto stamp = [&]() __attribute__((always_inline)) {
        if constexpr (TRACE) {
            if (tid == 0 && tr_n < 64) a.trace[(size_t)blockIdx.x * 64 + tr_n++] = __builtin_amdgcn_s_memtime();
        }
    };
    stamp();
    const int flags = STATS == 1 ? (a.flags & ~EPI_BNB) : STATS == 2 ? (a.flags & ~EPI_STATS) : (a.flags & ~(EPI_STATS | EPI_BNB));   // (instantiated without the statistics: no partial sums in registers)
    // ---- plan tables (conv_plan_tables) ----------------------------------------------------------------------------------------
    const int* __restrict__ blob = a.blob;
    // Register budget (two workgroups per CU: 256 registers, accumulators in ArchVGPRs so that the K loop carries no accvgpr copies
    // across its back edge): of the per-thread patch units only the LDS offset and the row word stay resident; the global offsets are
    // re-read from the plan tables whenever a patch is requested (12 coalesced loads from L2, a whole tile of MFMAs ahead of their use).
    int pu_lds[PF], pu_rp[PF];
    const int* pu_tab = blob + a.off_pu + tid;
    {
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            pu_lds[i] = pu_tab[(3 * i + 1) * 256];
            pu_rp[i] = pu_tab[(3 * i + 2) * 256];
        }
    }
    int loc_p[NTQ], loc_o[NTQ], loc_il[NTQ];
    {
        const int* lc = blob + a.off_loc + tid;
#pragma unroll
        for (int nt = 0; nt < NTQ; ++nt) {
            loc_p[nt] = lc[(3 * nt + 0) * 256];
            loc_o[nt] = lc[(3 * nt + 1) * 256];
            loc_il[nt] = lc[(3 * nt + 2) * 256];
        }
    }
    const int4 tile0 = *(const int4*)(blob + a.off_tdesc + (size_t)t_begin * 8);
    const int ntab = 16 + 2 * a.Qpad, ntd = nwt * 8;
    const int* td = blob + a.off_tdesc + (size_t)t_begin * 8;
    const int tab0 = tid < ntab ? blob[tid] : 0, tab1 = tid + 256 < ntab ? blob[tid + 256] : 0;
    const int td0 = tid < ntd ? td[tid] : 0, td1 = tid + 256 < ntd ? td[tid + 256] : 0;
    const __amdgpu_buffer_rsrc_t rs_in = make_rsrc(a.in), rs_w = make_rsrc(a.wT);
    float4 pv[PF];
    unsigned okm = 0;
    auto load_patch_d = [&](const int4 d) __attribute__((always_inline)) {
        okm = 0;
        int goff[PF];
#pragma unroll
        for (int i = 0; i < PF; ++i) goff[i] = pu_tab[(3 * i + 0) * 256];
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            const int row = pu_rp[i] & 0xffff, pr = (pu_rp[i] >> 16) & 0xff;
            const bool ok = (row < d.z) & ((unsigned)(d.y + pr) < (unsigned)a.Hin) & (goff[i] >= 0);
            pv[i] = buf_load16(rs_in, ok ? d.x + goff[i] : kOob);
            okm |= ok ? (1u << i) : 0u;
        }
    };
    auto store_patch = [&](int nrows, int grp) __attribute__((always_inline)) {
        const float* tb = xft + (size_t)(grp * a.C4tot) * 8;
#pragma unroll
        for (int i = 0; i < PF; ++i)
            if ((pu_rp[i] & 0xffff) < nrows) {
                float4 v = pv[i];
                if (a.xf) {   // block-uniform (ConvArgs::xf)
                    const float* t = tb + (pu_rp[i] >> 24) * 8;
                    const float4 sc = *(const float4*)t, sh = *(const float4*)(t + 4);
                    v.x = fmaxf(__fmaf_rn(v.x, sc.x, sh.x), 0.f); v.y = fmaxf(__fmaf_rn(v.y, sc.y, sh.y), 0.f);
                    v.z = fmaxf(__fmaf_rn(v.z, sc.z, sh.z), 0.f); v.w = fmaxf(__fmaf_rn(v.w, sc.w, sh.w), 0.f);
                    if (!((okm >> i) & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f);
                }
                *(float4*)(patch + pu_lds[i]) = v;
            }
    };
    load_patch_d(tile0);
    if (a.xf) {
        const int C = a.Cin;
        const double M = (double)a.xf_m_per_group;
        const bool lead = blockIdx.x == 0;
        for (int j = tid; j < a.groups * C; j += 256) {
            const int gq = j / C, c = j - gq * C;
            double mean, var;
            bn_batch_moments(a.xf_stats, a.xf_rep_stride, gq, c, C, M, a.xf_eps, mean, var);
            const double xv = var + (double)a.xf_eps;
            double invstd = (double)rsqrtf((float)xv);
            invstd = invstd * (1.5 - 0.5 * xv * invstd * invstd);
            invstd = invstd * (1.5 - 0.5 * xv * invstd * invstd);
            float sc, sh;
            bn_scale_shift(a.xf_gamma[c], a.xf_beta[c], (float)mean, (float)invstd, sc, sh);
            float* t = xft + (size_t)(gq * (C >> 2) + (c >> 2)) * 8 + (c & 3);
            t[0] = sc;
            t[4] = sh;
            if (lead) {
                a.xf_save_mean[j] = (float)mean;
                a.xf_save_invstd[j] = (float)invstd;
            }
        }
        if (lead && a.xf_running_mean)
            bn_running_update(a.xf_stats, a.xf_rep_stride, a.groups, C, M, a.xf_momentum, a.xf_eps, a.xf_running_mean, a.xf_running_var, a.xf_nbt, tid, 256);
    }
    if (STATS == 2 && (flags & EPI_BNB)) bnb_table(a, const_cast<float*>(bnt), tid, 256);
    if (tid < ntab) ctab[tid] = tab0;
    if (tid + 256 < ntab) ctab[tid + 256] = tab1;
    if (tid < ntd) tdesc[tid] = td0;
    if (tid + 256 < ntd) tdesc[tid + 256] = td1;
    __syncthreads();
    {   // resident weights: global -> LDS without registers, as conv_t_kernel (pack rows [tap * C4tot + c4][WPT][4], columns < Cout <= WPT)
        const int units = a.Qpad * COPW;
#pragma unroll 4
        for (int u0 = wave * 64; u0 < units; u0 += 256) {
            const int u = u0 + lane;
            const int q = min(u, units - 1) / COPW, c = min(u, units - 1) - q * COPW;
            const int row = qrow[q];
            const int off = (u < units && row >= 0 && c < a.WPT) ? ((row * a.WPT + c) * 4) * 4 : kOob;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(wl + (size_t)u0 * 4), 16, off, 0, 0, 0);
        }
    }
    // BatchNorm statistics without partial sums in registers (they would cost 40 registers across the K loop and, with them, the second
    // workgroup per CU): after every tile a wave reduces its 2 * 20 values over its 64 pixels -- DPP over the 16-lane rows, the four
    // row sums through a wave-private LDS slot -- and adds them to its own accumulator slot in a fixed order (deterministic); the
    // flush sums the four waves' slots in fp64 and issues one atomic per channel, as conv_t_kernel does.
    float* qrows = (float*)(lds_raw + a.qstat_off);   // [4 waves][4 rows][2 * COPW]
    float* qacc = qrows + 4 * 4 * 2 * COPW;            // [4 waves][2 * COPW]
    if (STATS && tid < 4 * 2 * COPW) qacc[tid] = 0.f;
    int run_grp = -1;
    auto flush_stats = [&]() __attribute__((always_inline)) {
        __syncthreads();
        if (STATS && tid < 2 * COPW && run_grp >= 0) {
            const int which = tid / COPW, c = tid - which * COPW;
            if (c < a.Cout) {
                const double v = ((double)qacc[0 * 2 * COPW + tid] + (double)qacc[1 * 2 * COPW + tid]) +
                                 ((double)qacc[2 * 2 * COPW + tid] + (double)qacc[3 * 2 * COPW + tid]);
                StatCell* st_ = a.stats + (int64_t)(blockIdx.x % kStatReps) * a.stat_rep_stride;
                fx_add(&st_[((int64_t)run_grp * 2 + which) * a.Cout + c], v);
            }
        }
        __syncthreads();
        if (STATS && tid < 4 * 2 * COPW) qacc[tid] = 0.f;
        __syncthreads();
    };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // own weight DMA (and the first patch) landed; the tile loop's barriers publish
    stamp();
    const float* wlane = wl + (size_t)(lane & 3) * 4;
    for (int k = 0; k < nwt; ++k) {
        const int4 d0 = *(const int4*)(tdesc + k * 8);
        const int4 d1 = *(const int4*)(tdesc + k * 8 + 4);
        if ((flags & (EPI_STATS | EPI_BNB)) && d1.y != run_grp) {
            if (run_grp >= 0) flush_stats();
            run_grp = d1.y;
        }
        int pbase[NTQ], ooff[NTQ];
        if (a.aligned) {
#pragma unroll
            for (int nt = 0; nt < NTQ; ++nt) {
                const bool v = loc_il[nt] < d1.x;
                pbase[nt] = v ? loc_p[nt] : 0;
                ooff[nt] = v ? d0.w + loc_o[nt] : -1;
            }
        } else {
            const int img0 = d1.w & 0xfffff, ly0 = d1.w >> 20;
            const int grp_end = min(a.N, (d1.y + 1) * a.group_size);
#pragma unroll
            for (int nt = 0; nt < NTQ; ++nt) {
                const int r = wave * 64 * NTQ + nt * 64 + lane;
                int pl, lx;
                const int il = mdiv(r, a.m_ppi, a.ppi, pl);
                const int p = d1.z + pl;
                const int n = img0 + il;
                const bool v = (il < a.imgs) & (n < grp_end) & (p < LP);
                const int ly = mdiv(p, a.m_lw, a.LW, lx);
                pbase[nt] = v ? ((il * a.PR + (ly - ly0) * a.is) * a.PC + lx * a.is) * a.CP : 0;
                ooff[nt] = v ? ((n * a.Hout + ly * a.os + a.oy0) * a.Wout + lx * a.os + a.ox0) * a.Cout : -1;
            }
        }
        f32x4 acc[MB][NTQ];
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int nt = 0; nt < NTQ; ++nt) acc[m][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        __syncthreads();   // consumers of the previous patch are done
        stamp();
        store_patch(d0.z, d1.y);
        if (k + 1 < nwt) load_patch_d(*(const int4*)(tdesc + (k + 1) * 8));
        __syncthreads();   // patch (and, the first time, the weights and the transform table) visible
        stamp();
        {   // K loop: one (tap, channel quad) group per step, operands of group q + 1 read while the MFMAs of group q issue
            const int nq = a.Qc;
            float4 bv[2][NTQ], av[2][MB];
            int fq = 0;
            int po = qoff[0], po1 = qoff[min(1, nq - 1)];
            auto fetch = [&](int set) __attribute__((always_inline)) {
#pragma unroll
                for (int nt = 0; nt < NTQ; ++nt) bv[set][nt] = *(const float4*)(patch + pbase[nt] + po);
#pragma unroll
                for (int m = 0; m < MB; ++m) av[set][m] = *(const float4*)(wlane + (size_t)(fq * COPW + 4 * m) * 4);
                ++fq;
                po = po1;
                po1 = qoff[min(fq + 1, nq - 1)];
            };
            auto fma = [&](int set) __attribute__((always_inline)) {
#define OCL_QSTEP(E)                                                                                                      \
    _Pragma("unroll") for (int m = 0; m < MB; ++m) _Pragma("unroll") for (int nt = 0; nt < NTQ; ++nt)                     \
        acc[m][nt] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[set][m].E, bv[set][nt].E, acc[m][nt], 0, 0, 0);
                OCL_QSTEP(x) OCL_QSTEP(y) OCL_QSTEP(z) OCL_QSTEP(w)
#undef OCL_QSTEP
            };
            fetch(0);
            int q = 0;
            for (; q + 2 <= nq; q += 2) {
                fetch(1);
                __builtin_amdgcn_sched_barrier(0);
                fma(0);
                __builtin_amdgcn_sched_barrier(0);
                fetch(0);
                __builtin_amdgcn_sched_barrier(0);
                fma(1);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (q < nq) fma(0);
        }
        stamp();
        if (STATS == 1 && (flags & EPI_STATS)) {   // this tile's sums over the wave's pixels -> the wave's accumulator slot
            float* rw = qrows + (size_t)(wave * 4 + (lane >> 4)) * 2 * COPW;
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float t1 = 0.f, t2 = 0.f;
#pragma unroll
                    for (int nt = 0; nt < NTQ; ++nt) {
                        const float v = ooff[nt] >= 0 ? acc[m][nt][e] : 0.f;
                        t1 += v;
                        t2 = fmaf(v, v, t2);
                    }
                    t1 = row16_sum(t1);
                    t2 = row16_sum(t2);
                    if ((lane & 15) == 0) {
                        rw[m * 4 + e] = t1;
                        rw[COPW + m * 4 + e] = t2;
                    }
                }
            if (lane < 2 * COPW) {   // (same wave: the writes above are ordered before these reads)
                const float* r0 = qrows + (size_t)(wave * 4) * 2 * COPW + lane;
                qacc[wave * 2 * COPW + lane] += (r0[0] + r0[2 * COPW]) + (r0[4 * COPW] + r0[6 * COPW]);
            }
        }
        // ---- epilogue from registers: the lane holds channels 4m .. 4m + 3 of its NTQ pixels ---------------------------------------
        // one (pixel set, channel block) of the tile: the flag-driven register epilogue
        auto epi_one = [&](int nt, int m, float (&b1)[4], float (&b2)[4]) __attribute__((always_inline)) {
            const int co = 4 * m;
            if (ooff[nt] < 0 || co >= a.Cout) return;
            float4 v = make_float4(acc[m][nt][0], acc[m][nt][1], acc[m][nt][2], acc[m][nt][3]);
            float* op = a.out + (int64_t)ooff[nt] + co;
            if (flags & EPI_AFFINE) {
                const float4 sc = *(const float4*)(a.scale + co), sh = *(const float4*)(a.shift + co);
                v.x = fmaf(v.x, sc.x, sh.x); v.y = fmaf(v.y, sc.y, sh.y); v.z = fmaf(v.z, sc.z, sh.z); v.w = fmaf(v.w, sc.w, sh.w);
            }
            if (flags & EPI_RES) {
                const float4 r = *(const float4*)(a.res + (int64_t)ooff[nt] + co);
                v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
            }
            if (flags & EPI_RESMASK) {
                const float4 r = *(const float4*)(a.res + (int64_t)ooff[nt] + co);
                const float4 mk = *(const float4*)(a.resmask + (int64_t)ooff[nt] + co);
                v.x += mk.x > 0.f ? r.x : 0.f; v.y += mk.y > 0.f ? r.y : 0.f; v.z += mk.z > 0.f ? r.z : 0.f; v.w += mk.w > 0.f ? r.w : 0.f;
            }
            if (flags & EPI_ACCUM) {
                const float4 o = *(const float4*)op;
                v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
            }
            if (STATS == 2 && (flags & EPI_BNB)) {
                const float* t = bnt + (size_t)(d1.y * (a.Cout >> 2) + m) * 12;
                bnb_apply(a, *(const float4*)t, *(const float4*)(t + 4), *(const float4*)(t + 8), (int64_t)ooff[nt] + co, v, b1, b2);
            }
            if (flags & EPI_RELU) {
                v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            }
            *(float4*)op = v;
        };
        if (STATS == 2 && (flags & EPI_BNB)) {
            // channel block by channel block (8 sum registers at a time): the block's sums over the wave's pixels -> the wave's
            // accumulator slot, as the forward's statistics
            float* rw = qrows + (size_t)(wave * 4 + (lane >> 4)) * 2 * COPW;
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                float b1[4] = {0.f, 0.f, 0.f, 0.f}, b2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int nt = 0; nt < NTQ; ++nt) epi_one(nt, m, b1, b2);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float r1 = row16_sum(b1[e]), r2 = row16_sum(b2[e]);
                    if ((lane & 15) == 0) {
                        rw[m * 4 + e] = r1;
                        rw[COPW + m * 4 + e] = r2;
                    }
                }
            }
            if (lane < 2 * COPW) {   // (same wave: the writes above are ordered before these reads)
                const float* r0 = qrows + (size_t)(wave * 4) * 2 * COPW + lane;
                qacc[wave * 2 * COPW + lane] += (r0[0] + r0[2 * COPW]) + (r0[4 * COPW] + r0[6 * COPW]);
            }
        } else {
            float nb1[4], nb2[4];   // (unused)
#pragma unroll
            for (int nt = 0; nt < NTQ; ++nt)
#pragma unroll
                for (int m = 0; m < MB; ++m) epi_one(nt, m, nb1, nb2);
        }
        stamp();
    }
    if ((flags & (EPI_STATS | EPI_BNB)) && run_grp >= 0) flush_stats();
    stamp();
}


// =====================================================================================================
// conv_s_kernel: few output pixels behind a deep K (layer 4 at every batch size, layer 3 below ~200 images)
// =====================================================================================================
// A 20-image pass has 320 output pixels on layer 4 and 1280 on layer 3: five / twenty 64-pixel tiles.  conv_t_kernel gives every
// wave 16 of a tile's pixels and the WHOLE K dimension -- 360 dependent-chain MFMAs per wave on layer 4, on 20 - 60 workgroups of
// the 256 CUs: 14 - 20 us for 0.15 GFLOP (profiles/r3_aser_kernel_stats_v2_single_stream.csv: 25 such launches per ASER step).
// Here a workgroup owns 16 NT pixels x 16 channels and its four waves split K by INPUT CHANNELS (wave w: channels [w, w + 1) * Cin / 4,
// all taps): 4x the workgroups, a quarter of the chain; each wave stages its own channel slice of the (shared-halo) patch, takes its
// weights straight from the pack in global memory / L2 into registers (16 bytes per lane and round, one loop body of four rounds
// ahead: nothing about them is shared between waves, so LDS would only add a copy), and the four partial tiles meet in LDS, where
// wave j adds those of pixel tile j in a fixed order and runs the usual register epilogue.  Tables, input transform and epilogue flags
// as in conv_t_kernel.  NT = 16-pixel tiles per workgroup: at NT = 2 every weight quad and every table entry feeds two MFMAs, for
// twice the patch per wave -- it pays on layer 3's 8x8 lattices from ~100 images on and on the 84x84 input's lattices, not on
// layer 4's 4x4 images (profiles/r3_conv_s_ab.md, which also has the per-wave phase traces and the counter passes).
// The kernel must stay free of scratch: a build with 10 spilled VGPRs was 1 - 4 us per launch slower than the one before it.
constexpr int kDepthS = 4;    // weight rounds in flight per wave
constexpr int kPFS = 7;       // patch units (16 bytes) per lane and staging pass: a wave stages 448 units per pass (layer 4 needs 360 - 405; 8 would spill at 96 VGPRs)
template <int NT, bool TRACE, bool BNB = false, bool DET = false>   // BNB: instantiated with the EPI_BNB epilogue; DET: for the deterministic batch sums
__global__ void __launch_bounds__(256, NT == 1 ? 5 : 4) conv_s_kernel(const ConvArgs a) {
    constexpr int FXM = DET ? 1 : 0;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    int* ctab = (int*)lds_raw;
    int* qoff = ctab + 16;                           // [4][Qpad / 4] patch offset of group q = 4 rho + g, stored [g][rho] (one wave's channel slice)
    int* qrow = qoff + a.Qpad;                       // [4][Qpad / 4 + 4] pack row of group q relative to the slice's first channel quad, same order; each row ends in four -1 ("no load")
    float* patch0 = (float*)(qrow + a.Qpad + 16);    // [4 waves][patch_floats]; after the K loop each wave's slice holds its partial tiles [NT][64 lanes][4]
    float* xft = patch0 + (size_t)4 * a.patch_floats;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.y * 16;
    const int flags = BNB ? a.flags : (a.flags & ~EPI_BNB);
    const int* __restrict__ blob = a.blob;
    const int tile = blockIdx.x;
    const int c0 = wave * a.KC;                      // this wave's channel slice
    float* patch = patch0 + (size_t)wave * a.patch_floats;
    // (TRACE, a measurement build launched when ConvArgs::trace is set: s_memtime stamps of lane 0 of every wave, 8 slots per wave -- kbench KBENCH_TRACE)
    unsigned long long* trp = TRACE ? a.trace + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 64 + wave * 8 : nullptr;
    auto stamp = [&](int i) __attribute__((always_inline)) { if (TRACE && lane == 0) trp[i] = __builtin_amdgcn_s_memtime(); };
    stamp(0);
    const int4 d0 = *(const int4*)(blob + a.off_tdesc + (size_t)tile * 8);       // in_base, iy0, nrows, obase
    const int4 d1 = *(const int4*)(blob + a.off_tdesc + (size_t)tile * 8 + 4);   // nimg, grp, p0, img0 | ly0 << 20
    const __amdgpu_buffer_rsrc_t rs_in = make_rsrc(a.in), rs_w = make_rsrc(a.wT);
    const int c4base = c0 >> 2;
    // ---- the lane's patch units (64-lane walk), kPFS per staging pass: table entries -> loads -> (transform) -> the wave's LDS slice ------
    int pu_lds[kPFS], pu_rp[kPFS];
    float4 pv[kPFS];
    unsigned okm = 0;
    auto stage_load = [&](int pass) __attribute__((always_inline)) {
        const int* pu = blob + a.off_pu + pass * (3 * kPFS * 256) + lane;
        okm = 0;
#pragma unroll
        for (int i = 0; i < kPFS; ++i) {
            const int goff = pu[(3 * i + 0) * 256];
            pu_lds[i] = pu[(3 * i + 1) * 256];
            pu_rp[i] = pu[(3 * i + 2) * 256];
            const int row = pu_rp[i] & 0xffff, pr = (pu_rp[i] >> 16) & 0xff;
            const bool ok = (row < d0.z) & ((unsigned)(d0.y + pr) < (unsigned)a.Hin) & (goff >= 0);
            pv[i] = buf_load16(rs_in, ok ? d0.x + c0 * 4 + goff : kOob);
            okm |= ok ? (1u << i) : 0u;
        }
    };
    auto stage_store = [&]() __attribute__((always_inline)) {
        const float* tb = xft + (size_t)(d1.y * a.C4tot + c4base) * 8;
#pragma unroll
        for (int i = 0; i < kPFS; ++i)
            if ((pu_rp[i] & 0xffff) < d0.z) {
                float4 v = pv[i];
                if (a.xf) {
                    const float* t = tb + (pu_rp[i] >> 24) * 8;
                    const float4 sc = *(const float4*)t, sh = *(const float4*)(t + 4);
                    v.x = fmaxf(__fmaf_rn(v.x, sc.x, sh.x), 0.f); v.y = fmaxf(__fmaf_rn(v.y, sc.y, sh.y), 0.f);
                    v.z = fmaxf(__fmaf_rn(v.z, sc.z, sh.z), 0.f); v.w = fmaxf(__fmaf_rn(v.w, sc.w, sh.w), 0.f);
                    if (!((okm >> i) & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f);
                }
                *(float4*)(patch + pu_lds[i]) = v;
            }
    };
    stage_load(0);
    // ---- the lane's output pixels, the group tables ------------------------------------------------------------------------------------------
    int loc_p[NT], loc_o[NT], loc_il[NT];
    {
        const int* lc = blob + a.off_loc + r16;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            loc_p[nt] = lc[(3 * nt + 0) * 256];
            loc_o[nt] = lc[(3 * nt + 1) * 256];
            loc_il[nt] = lc[(3 * nt + 2) * 256];
        }
    }
    const int ntab = 16 + 2 * a.Qpad;
    const int tab0 = tid < ntab ? blob[tid] : 0, tab1 = tid + 256 < ntab ? blob[tid + 256] : 0;
    if (a.xf) {
        const int C = a.Cin;
        const double M = (double)a.xf_m_per_group;
        const bool lead = blockIdx.x == 0 && blockIdx.y == 0;
        // A tile lies inside one BatchNorm group (d1.y) and stage_store reads that group's rows only: the ~2000 workgroups of a layer-4
        // launch each build ONE group's table, two replica loads in flight per thread (every group's with one load in flight was 16
        // dependent L2 round trips per entry and two entries per thread: 6.4 us of a 32 us launch, profiles/r6_convs_xf_prologue_ab.txt);
        // the lead workgroup builds every group's (it saves mean / invstd for the backward).
        const int j_end = lead ? a.groups * C : (d1.y + 1) * C;
        for (int j = (lead ? 0 : d1.y * C) + tid; j < j_end; j += 256) {
            const int gq = j / C, c = j - gq * C;
            double mean, var;
            bn_batch_moments<2, FXM>(a.xf_stats, a.xf_rep_stride, gq, c, C, M, a.xf_eps, mean, var);
            const double xv = var + (double)a.xf_eps;
            double invstd = (double)rsqrtf((float)xv);
            invstd = invstd * (1.5 - 0.5 * xv * invstd * invstd);
            invstd = invstd * (1.5 - 0.5 * xv * invstd * invstd);
            float sc, sh;
            bn_scale_shift(a.xf_gamma[c], a.xf_beta[c], (float)mean, (float)invstd, sc, sh);
            float* t = xft + (size_t)(gq * (C >> 2) + (c >> 2)) * 8 + (c & 3);
            t[0] = sc;
            t[4] = sh;
            if (lead) {
                a.xf_save_mean[j] = (float)mean;
                a.xf_save_invstd[j] = (float)invstd;
            }
        }
        if (lead && a.xf_running_mean)
            bn_running_update<FXM>(a.xf_stats, a.xf_rep_stride, a.groups, C, M, a.xf_momentum, a.xf_eps, a.xf_running_mean, a.xf_running_var, a.xf_nbt, tid, 256);
    }
    stamp(1);
    const int nr = a.Qpad >> 2;                      // rounds of 4 groups; a multiple of 4 (the planner pads with zero-weight groups)
    {   // the group tables transposed to [g][rho]: a lane fetches four rounds of its g with one 16-byte read
        auto tpos = [&](int t) __attribute__((always_inline)) -> int {
            if (t < 16) return t;
            int e = t - 16, base = 16;
            if (e >= a.Qpad) return 16 + a.Qpad + ((e - a.Qpad) & 3) * (nr + 4) + ((e - a.Qpad) >> 2);
            return base + (e & 3) * nr + (e >> 2);
        };
        if (tid < ntab) ctab[tpos(tid)] = tab0;
        if (tid + 256 < ntab) ctab[tpos(tid + 256)] = tab1;
        if (tid < 16) qrow[(tid >> 2) * (nr + 4) + nr + (tid & 3)] = -1;
    }
    __syncthreads();   // group tables (and the transform table) visible
    stamp(2);
    // ---- weights: round rho of this wave = groups 4 rho + g, one 16-byte load per lane, four rounds (one loop body) ahead -----------------
    const int wcol = n0 + r16;
    const int* qoffT = qoff + g * nr;
    const int* qrowT = qrow + g * (nr + 4);
    const bool wok = wcol < a.WPT;
    const int wbase = (c4base * a.WPT + wcol) * 16, wstride = a.WPT * 16;
    auto w_addr = [&](int row) __attribute__((always_inline)) -> int { return (row >= 0 && wok) ? row * wstride + wbase : kOob; };
    int4 qr = *(const int4*)qrowT;                   // pack rows of rounds 0 .. 3
    float4 aw[kDepthS];
    aw[0] = buf_load16(rs_w, w_addr(qr.x)); aw[1] = buf_load16(rs_w, w_addr(qr.y));
    aw[2] = buf_load16(rs_w, w_addr(qr.z)); aw[3] = buf_load16(rs_w, w_addr(qr.w));
    qr = *(const int4*)(qrowT + 4);                  // rounds 4 .. 7: the loads the first body issues (past the last round: -1, no load)
    int4 qo = *(const int4*)qoffT;                   // patch offsets of rounds 0 .. 3
    // ---- this wave's patch slice (private to the wave: no workgroup barrier, its own LDS writes are ordered before its reads) -------------
    stamp(3);
    stage_store();
    for (int pass = 1; pass < a.nstage; ++pass) {
        stage_load(pass);
        stage_store();
    }
    stamp(4);
    int pbase[NT], ooff[NT];
    if (a.aligned) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const bool pix_ok = loc_il[nt] < d1.x;
            pbase[nt] = pix_ok ? loc_p[nt] : 0;
            ooff[nt] = pix_ok ? d0.w + loc_o[nt] : -1;
        }
    } else {   // tiles that start inside a lattice row (11 x 11, 21 x 21 lattices of the 84 x 84 input): one image per tile
        const int img0 = d1.w & 0xfffff, ly0 = d1.w >> 20;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int p = d1.z + nt * 16 + r16;
            const bool v = p < a.LH * a.LW;
            int lx;
            const int ly = mdiv(p, a.m_lw, a.LW, lx);
            pbase[nt] = v ? (((ly - ly0) * a.is) * a.PC + lx * a.is) * a.CP : 0;
            ooff[nt] = v ? ((img0 * a.Hout + ly * a.os + a.oy0) * a.Wout + lx * a.os + a.ox0) * a.Cout : -1;
        }
    }
    f32x4 acc[NT][2];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt][0] = acc[nt][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // One body = four rounds, no branches: the B operands of the body and the tables of the next are requested at its top, each
    // weight register is refilled (for the next body) right after it is read, and the MFMAs of two rounds alternate between two
    // accumulators per pixel tile.  With one wave per SIMD (a 20-image pass) the loop ran at 578 cycles per round of 4 MFMAs -- two
    // dependent LDS round trips (table, then operand) and a 4-MFMA chain per round; profiles/r3_conv_s_ab.md.
    for (int rho = 0; rho < nr; rho += 4) {
        float4 bv[NT][4];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            bv[nt][0] = *(const float4*)(patch + pbase[nt] + qo.x); bv[nt][1] = *(const float4*)(patch + pbase[nt] + qo.y);
            bv[nt][2] = *(const float4*)(patch + pbase[nt] + qo.z); bv[nt][3] = *(const float4*)(patch + pbase[nt] + qo.w);
        }
        const int4 qo_n = *(const int4*)(qoffT + min(rho + 4, nr - 4));
        const int4 qr_n = *(const int4*)(qrowT + min(rho + 8, nr));
        const int qrv[4] = {qr.x, qr.y, qr.z, qr.w};
#pragma unroll
        for (int i = 0; i < 4; i += 2) {
            const float4 a0 = aw[i], a1 = aw[i + 1];
            aw[i] = buf_load16(rs_w, w_addr(qrv[i]));
            aw[i + 1] = buf_load16(rs_w, w_addr(qrv[i + 1]));
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                acc[nt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, bv[nt][i].x, acc[nt][0], 0, 0, 0);
                acc[nt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, bv[nt][i + 1].x, acc[nt][1], 0, 0, 0);
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                acc[nt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, bv[nt][i].y, acc[nt][0], 0, 0, 0);
                acc[nt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, bv[nt][i + 1].y, acc[nt][1], 0, 0, 0);
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                acc[nt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, bv[nt][i].z, acc[nt][0], 0, 0, 0);
                acc[nt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, bv[nt][i + 1].z, acc[nt][1], 0, 0, 0);
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                acc[nt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, bv[nt][i].w, acc[nt][0], 0, 0, 0);
                acc[nt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, bv[nt][i + 1].w, acc[nt][1], 0, 0, 0);
            }
        }
        qo = qo_n;
        qr = qr_n;
    }
    stamp(5);
    // (the wave's own patch slice is dead once its K loop is done: the partial tiles go there, no extra buffer and no extra barrier)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const f32x4 t = acc[nt][0] + acc[nt][1];
        *(float4*)(patch + (size_t)(nt * 64 + lane) * 4) = make_float4(t[0], t[1], t[2], t[3]);
    }
    __syncthreads();
    stamp(6);
    if (wave >= NT) return;   // wave j adds the four partial tiles of pixel tile j in a fixed order and runs its epilogue
    float4 v;
    {
        const float* rj = patch0 + (size_t)(wave * 64 + lane) * 4;
        const size_t ws = (size_t)a.patch_floats;
        const float4 p0 = *(const float4*)(rj), p1 = *(const float4*)(rj + ws);
        const float4 p2 = *(const float4*)(rj + 2 * ws), p3 = *(const float4*)(rj + 3 * ws);
        v = make_float4((p0.x + p1.x) + (p2.x + p3.x), (p0.y + p1.y) + (p2.y + p3.y), (p0.z + p1.z) + (p2.z + p3.z), (p0.w + p1.w) + (p2.w + p3.w));
    }
    // ---- epilogue: lane (r16 = pixel of tile `wave`, g) holds channels n0 + 4g .. + 3 -------------------------------------------------------
    int oo = ooff[0];
#pragma unroll
    for (int nt = 1; nt < NT; ++nt) oo = wave == nt ? ooff[nt] : oo;
    const int co = n0 + 4 * g;
    const bool live = oo >= 0 && co < a.Cout;
    if (flags & EPI_STATS) {   // sums over the tile's pixels (DPP row of 16 lanes), one fp64 atomic per channel
        const float4 z = live ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        const float s1x = row16_sum(z.x), s1y = row16_sum(z.y), s1z = row16_sum(z.z), s1w = row16_sum(z.w);
        const float s2x = row16_sum(z.x * z.x), s2y = row16_sum(z.y * z.y), s2z = row16_sum(z.z * z.z), s2w = row16_sum(z.w * z.w);
        if (r16 < 8 && co < a.Cout) {   // (every lane of the row holds the eight sums: lane j adds sum j -- one accumulation per lane)
            const int j = r16;
            const float v = j == 0 ? s1x : j == 1 ? s1y : j == 2 ? s1z : j == 3 ? s1w : j == 4 ? s2x : j == 5 ? s2y : j == 6 ? s2z : s2w;
            StatCell* st_ = a.stats + (int64_t)((blockIdx.x + blockIdx.y + wave) % kStatReps) * a.stat_rep_stride + ((int64_t)d1.y * 2) * a.Cout + co;
            fx_add<FXM>(st_ + (j >> 2) * a.Cout + (j & 3), (double)v);
        }
    }
    float* op = a.out + (int64_t)oo + co;
    if (live) {
        if (flags & EPI_AFFINE) {
            const float4 sc = *(const float4*)(a.scale + co), sh = *(const float4*)(a.shift + co);
            v.x = fmaf(v.x, sc.x, sh.x); v.y = fmaf(v.y, sc.y, sh.y); v.z = fmaf(v.z, sc.z, sh.z); v.w = fmaf(v.w, sc.w, sh.w);
        }
        if (flags & EPI_RES) {
            const float4 r = *(const float4*)(a.res + (int64_t)oo + co);
            v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
        }
        if (flags & EPI_RESMASK) {
            const float4 r = *(const float4*)(a.res + (int64_t)oo + co);
            const float4 mk = *(const float4*)(a.resmask + (int64_t)oo + co);
            v.x += mk.x > 0.f ? r.x : 0.f; v.y += mk.y > 0.f ? r.y : 0.f; v.z += mk.z > 0.f ? r.z : 0.f; v.w += mk.w > 0.f ? r.w : 0.f;
        }
        if (flags & EPI_ACCUM) {
            const float4 o = *(const float4*)op;
            v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
        }
    }
    if (BNB && (flags & EPI_BNB)) {   // ReLU mask + the two batch sums of the BatchNorm this gradient enters; one channel quad per lane: the
                             // BatchNorm's parameters come straight from memory (no table), after the K loop (no registers across it)
        float b1[4] = {0.f, 0.f, 0.f, 0.f}, b2[4] = {0.f, 0.f, 0.f, 0.f};
        if (live) {
            const int j = d1.y * a.Cout + co;
            const float4 mu = *(const float4*)(a.bnb_mean + j);
            float4 sc = make_float4(0.f, 0.f, 0.f, 0.f), sh = sc;
            if (!a.bnb_z) {
                const float4 is = *(const float4*)(a.bnb_invstd + j), gm = *(const float4*)(a.bnb_gamma + co), bt = *(const float4*)(a.bnb_beta + co);
                bn_scale_shift(gm.x, bt.x, mu.x, is.x, sc.x, sh.x); bn_scale_shift(gm.y, bt.y, mu.y, is.y, sc.y, sh.y);
                bn_scale_shift(gm.z, bt.z, mu.z, is.z, sc.z, sh.z); bn_scale_shift(gm.w, bt.w, mu.w, is.w, sc.w, sh.w);
            }
            bnb_apply(a, sc, sh, mu, (int64_t)oo + co, v, b1, b2);
        }
        const float s1x = row16_sum(b1[0]), s1y = row16_sum(b1[1]), s1z = row16_sum(b1[2]), s1w = row16_sum(b1[3]);
        const float s2x = row16_sum(b2[0]), s2y = row16_sum(b2[1]), s2z = row16_sum(b2[2]), s2w = row16_sum(b2[3]);
        if (r16 < 8 && co < a.Cout) {   // (every lane of the row holds the eight sums: lane j adds sum j -- one accumulation per lane)
            const int j = r16;
            const float v = j == 0 ? s1x : j == 1 ? s1y : j == 2 ? s1z : j == 3 ? s1w : j == 4 ? s2x : j == 5 ? s2y : j == 6 ? s2z : s2w;
            StatCell* st_ = a.stats + (int64_t)((blockIdx.x + blockIdx.y + wave) % kStatReps) * a.stat_rep_stride + ((int64_t)d1.y * 2) * a.Cout + co;
            fx_add<FXM>(st_ + (j >> 2) * a.Cout + (j & 3), (double)v);
        }
    }
    if (!live) return;
    if (flags & EPI_RELU) {
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    }
    *(float4*)op = v;
    stamp(7);
}

typedef void (*conv_fn_t)(const ConvArgs);
static conv_fn_t convs_fn(int nt, bool trace = false, bool bnb = false, bool det = false) {
    if (det) {
        if (bnb) return nt == 2 ? conv_s_kernel<2, false, true, true> : conv_s_kernel<1, false, true, true>;
        return nt == 2 ? conv_s_kernel<2, false, false, true> : conv_s_kernel<1, false, false, true>;
    }
    if (bnb) return nt == 2 ? conv_s_kernel<2, false, true> : conv_s_kernel<1, false, true>;
    if (trace) return nt == 2 ? conv_s_kernel<2, true> : conv_s_kernel<1, true>;
    return nt == 2 ? conv_s_kernel<2, false> : conv_s_kernel<1, false>;
}

#define OCL_CONVT_TILINGS(X) X(1, 1) X(2, 1) X(3, 1) X(4, 1) X(5, 1) X(1, 2) X(2, 2) X(3, 2) X(4, 2) X(5, 2)
static conv_fn_t convt_fn(int MT, int NT, int PF, int res, int cls = 0, int pipe = 0, int bnb = 0) {
    if (bnb) {   // the EPI_BNB epilogue: stride-1 data gradients only (no output classes)
        if (cls) return nullptr;
        if (pipe) {
            if (res || NT != 1) return nullptr;
#define OCL_CASE(M)                                                                                     \
    if (MT == M) {                                                                                      \
        if (PF == 4) return conv_t_kernel<M, 1, 4, false, false, true, true>;                           \
        if (PF == 8) return conv_t_kernel<M, 1, 8, false, false, true, true>;                           \
    }
            OCL_CASE(1) OCL_CASE(2) OCL_CASE(3) OCL_CASE(4) OCL_CASE(5)
#undef OCL_CASE
            return nullptr;
        }
#define OCL_CASE(M, N)                                                                                                              \
    if (MT == M && NT == N) {                                                                                                       \
        if (PF == 4) return res ? conv_t_kernel<M, N, 4, true, false, false, true> : conv_t_kernel<M, N, 4, false, false, false, true>;   \
        if (PF == 8) return res ? conv_t_kernel<M, N, 8, true, false, false, true> : conv_t_kernel<M, N, 8, false, false, false, true>;   \
    }
        OCL_CONVT_TILINGS(OCL_CASE)
#undef OCL_CASE
        return nullptr;
    }
    if (pipe) {   // staged weights through the ring: one pixel tile per wave
        if (res || NT != 1) return nullptr;
#define OCL_CASE(M)                                                                                                                  \
    if (MT == M) {                                                                                                                   \
        if (PF == 4) return cls ? conv_t_kernel<M, 1, 4, false, true, true> : conv_t_kernel<M, 1, 4, false, false, true>;            \
        if (PF == 8) return cls ? conv_t_kernel<M, 1, 8, false, true, true> : conv_t_kernel<M, 1, 8, false, false, true>;            \
    }
        OCL_CASE(1) OCL_CASE(2) OCL_CASE(3) OCL_CASE(4) OCL_CASE(5)
#undef OCL_CASE
        return nullptr;
    }
    if (cls) {   // output classes: one pixel tile per wave (the class lattices are the small ones)
#define OCL_CASE(M)                                                                                              \
    if (MT == M && NT == 1) {                                                                                    \
        if (PF == 4) return res ? conv_t_kernel<M, 1, 4, true, true> : conv_t_kernel<M, 1, 4, false, true>;      \
        if (PF == 8) return res ? conv_t_kernel<M, 1, 8, true, true> : conv_t_kernel<M, 1, 8, false, true>;      \
    }
        OCL_CASE(1) OCL_CASE(2) OCL_CASE(3) OCL_CASE(4) OCL_CASE(5)
#undef OCL_CASE
        return nullptr;
    }
#define OCL_CASE(M, N)                                                                              \
    if (MT == M && NT == N) {                                                                       \
        if (PF == 4) return res ? conv_t_kernel<M, N, 4, true> : conv_t_kernel<M, N, 4, false>;     \
        if (PF == 8) return res ? conv_t_kernel<M, N, 8, true> : conv_t_kernel<M, N, 8, false>;     \
    }
    OCL_CONVT_TILINGS(OCL_CASE)
#undef OCL_CASE
    return nullptr;
}
static int convt_pf_for(int units) { return units <= 1024 ? 4 : 8; }

// ---- conv_t_kernel layout: fills the tile-dependent fields for (MT channel tiles, NT pixel tiles); returns LDS bytes (0: no fit)
static size_t convt_layout(const ConvGeomDesc& g, ConvArgs& a, int MT, int NT, bool pipe = false) {
    const int nt16 = cdiv(g.Cout, 16);
    const int splits = cdiv(nt16, MT);
    const int COPW = 16 * MT;
    a.n_splits = splits;
    a.CoutP = splits * COPW;
    a.group_size = g.N / g.groups;
    const int LP = g.LH * g.LW;
    const int BM = 64 * NT;
    if (LP >= BM) {
        a.imgs = 1; a.ppi = BM; a.tiles_per_img = cdiv(LP, BM);
    } else {
        a.imgs = std::min(BM / LP, a.group_size); a.ppi = LP; a.tiles_per_img = 1;
    }
    a.PC = (g.LW - 1) * g.is + (a.max_dx - a.min_dx) + 1;
    const int rows_l = (a.imgs == 1 && LP >= BM) ? std::min(g.LH, (BM + g.LW - 2) / g.LW + 1) : g.LH;
    a.PR = (rows_l - 1) * g.is + (a.max_dy - a.min_dy) + 1;
    a.C4tot = g.Cin / 4;
    size_t bytes = 0;
    for (;;) {
        for (int KC = g.Cin; KC >= 4; KC -= 4) {
            if (g.Cin % KC) continue;
            a.KC = KC;
            a.CP = ((KC / 4) & 1) ? KC : KC + 4;     // 16-byte pixel slots, an odd number of them per pixel: b128 reads of 16 pixels spread over all banks
            a.Qc = g.ntaps * (KC / 4);
            a.Qpad = 0;   // every class's groups are padded to whole rounds of 4
            for (int c = 0; c < std::max(1, g.ncls); ++c) a.Qpad += (int)round_up((g.ncls > 1 ? g.cls_ntaps[c] : g.ntaps) * (KC / 4), 4);
            const size_t w_all = (size_t)a.Qpad * COPW * 16;
            a.wres = (KC == g.Cin && w_all <= kResidentBytes) ? 1 : 0;
            a.pipe = (pipe && !a.wres && NT == 1) ? 1 : 0;   // ring of three stage buffers of pipe_qs(MT) groups (conv_t_kernel<..., PIPE>)
            a.QS = a.wres ? a.Qpad : a.pipe ? pipe_qs(MT) : std::min(a.Qpad, ((256 * kWPF) / COPW) & ~3);
            a.nstage = cdiv(a.Qpad, a.QS);
            const size_t patch_b = (size_t)round_up(std::max((size_t)a.imgs * a.PR * a.PC * a.CP * 4, (size_t)8 * COPW * 8), 16);
            a.patch_floats = (int)(patch_b / 4);
            const size_t xf_b = g.xf ? (size_t)g.groups * g.Cin * 8 : 0;   // input transform: scale / shift per (group, channel)
            const size_t bnb_b = g.bnb ? (size_t)g.groups * g.Cout * 12 : 0;   // EPI_BNB: scale / shift / mean per (group, output channel)
            a.bnb_lds = g.bnb ? (int)(xf_b / 4) : -1;
            bytes = (size_t)kMaxWgTiles * 32 + 64 + (size_t)2 * a.Qpad * 4 + (a.wres ? w_all : (size_t)(a.pipe ? 3 : 2) * a.QS * COPW * 16) + patch_b + xf_b + bnb_b;
            const bool units_ok = a.imgs * a.PR * a.PC * (KC / 4) <= 256 * kConvPatchPF;
            if (units_ok && bytes <= kLdsLimit - 2048 && (bytes <= 100 * 1024 || KC <= 20)) goto found;
        }
        if (a.imgs > 1) {   // shrink the tile (fewer images per workgroup) and retry
            a.imgs -= 1;
            continue;
        }
        return 0;
    }
found:
    if (a.imgs > 127 || a.PR >= 256 || a.PC >= 256 || a.KC / 4 >= 256) return 0;
    a.tiles_per_group = cdiv(a.group_size, a.imgs) * a.tiles_per_img;
    return bytes;
}

// ---- conv_q_kernel plan: 5 blocks of 4 channels, NTQ 64-pixel sets per wave (tile = 256 * NTQ pixels), weights resident, one chunk
static int plan_conv_q_ntq(const ConvGeomDesc& g, ConvPlan* p, const int NTQ, bool* too_wide = nullptr) {
    ConvArgs& a = p->a;
    constexpr int COPW = 4 * kQBlocks;
    if (g.Cout > COPW || g.Cout % 4 || g.ncls > 1) return OCL_ERR_ARG;
    a.n_splits = 1;
    a.CoutP = COPW;
    a.group_size = g.N / g.groups;
    const int LP = g.LH * g.LW, BM = 256 * NTQ;
    if (LP >= BM) {
        a.imgs = 1; a.ppi = BM; a.tiles_per_img = cdiv(LP, BM);
    } else {
        a.imgs = std::min(BM / LP, a.group_size); a.ppi = LP; a.tiles_per_img = 1;
    }
    a.PC = (g.LW - 1) * g.is + (a.max_dx - a.min_dx) + 1;
    // lattice rows a tile can touch: exactly BM / LW when tiles start at row boundaries, one more when they straddle
    const bool whole_rows = BM % g.LW == 0 && LP % BM == 0;
    const int rows_l = (a.imgs == 1 && LP >= BM) ? (whole_rows ? BM / g.LW : std::min(g.LH, (BM + g.LW - 2) / g.LW + 1)) : g.LH;
    a.PR = (rows_l - 1) * g.is + (a.max_dy - a.min_dy) + 1;
    a.C4tot = g.Cin / 4;
    a.KC = g.Cin;
    a.CP = ((a.KC / 4) & 1) ? a.KC : a.KC + 4;
    a.Qc = g.ntaps * (a.KC / 4);
    a.Qpad = (int)round_up(a.Qc, 4);
    a.wres = 1; a.pipe = 0; a.QS = a.Qpad; a.nstage = 1;
    const int units = a.imgs * a.PR * a.PC * (a.KC / 4);
    if (too_wide) *too_wide = units > 256 * 12;
    if (units > 256 * 12 || a.imgs > 127 || a.PR >= 256 || a.PC >= 256 || a.KC / 4 >= 64) return OCL_ERR_ARG;
    const int PF = (units <= 1024 && NTQ == 2) ? 4 : 12;
    const size_t patch_b = (size_t)round_up(std::max((size_t)a.imgs * a.PR * a.PC * a.CP * 4, (size_t)8 * COPW * 8), 16);
    a.patch_floats = (int)(patch_b / 4);
    size_t lds = (size_t)kMaxWgTiles * 32 + 64 + (size_t)2 * a.Qpad * 4 + (size_t)a.Qpad * COPW * 16 + patch_b + (g.xf ? (size_t)g.groups * g.Cin * 8 : 0) +
                 (g.bnb ? (size_t)g.groups * g.Cout * 12 : 0);
    a.bnb_lds = g.bnb ? (g.xf ? g.groups * g.Cin * 2 : 0) : -1;
    a.qstat_off = (int)round_up(lds, 16);
    lds = (size_t)a.qstat_off + (size_t)(4 * 4 + 4) * 2 * COPW * 4;   // statistics scratch: row sums + per-wave accumulators
    if (lds > kLdsLimit - 2048) return OCL_ERR_ARG;
    a.tiles_per_group = cdiv(a.group_size, a.imgs) * a.tiles_per_img;
    const int ntiles = g.groups * a.tiles_per_group;
    // the form wants the machine full of whole tiles: below ~half a tile per CU the 64-pixel tiles of conv_t_kernel spread better
    if (ntiles < 128 && g.force_q4 <= 0) return OCL_ERR_ARG;
    a.cls_pack = 1 | (g.ntaps << 4);
    a.cls_oyx = 0;
    p->q4 = NTQ; p->MT = kQBlocks; p->NT = NTQ;
    p->lds_bytes = lds;
    a.WPT = g.WPT > 0 ? g.WPT : (int)round_up(g.Cout, 16);
    for (int t = 0; t < 9; ++t) a.tpo[t] = t < a.ntaps ? ((a.tdy[t] - a.min_dy) * a.PC + (a.tdx[t] - a.min_dx)) * a.CP : 0;
    {
        const int kc4 = a.KC / 4;
        a.d_c4 = 256 % kc4;
        const int d_pix = 256 / kc4;
        a.d_pc = d_pix % a.PC;
        a.d_row = d_pix / a.PC;
    }
    a.groups = g.groups;
    a.aligned = a.imgs > 1 ? 1 : ((BM % g.LW == 0 && LP % BM == 0) ? 1 : 0);
    {
        auto magic = [](int d) { return d <= 1 ? 0u : (unsigned)(((1ull << 32) + (unsigned)d - 1) / (unsigned)d); };
        a.m_tpg = magic(a.tiles_per_group); a.m_tpi = magic(a.tiles_per_img); a.m_lw = magic(a.LW); a.m_ppi = magic(a.ppi);
        a.m_kc4 = magic(a.KC / 4); a.m_pc = magic(a.PC); a.m_pr = magic(a.PR);
        const int64_t xmax = std::max<int64_t>(std::max<int64_t>(ntiles, (int64_t)LP + BM), 4096);
        const int64_t dmax = std::max(std::max(a.tiles_per_group, a.tiles_per_img), std::max(std::max(a.LW, a.ppi), std::max(a.PC, a.PR)));
        if (xmax * dmax >= (1ll << 32)) return OCL_ERR_ARG;
    }
    {   // two workgroups per CU where the LDS allows it: the second wave per SIMD covers the other's operand reads
        int bpc = (int)std::min<size_t>(2, kLdsLimit / (lds + 512));
        if (g.force_bpc) bpc = g.force_bpc;
        p->grid_x = std::max(1, std::min(ntiles, 256 * std::max(1, bpc)));
    }
    p->grid_x = std::max(p->grid_x, cdiv(ntiles, kMaxWgTiles));
    p->grid_y = 1;
    a.off_tdesc = (int)round_up(16 + 2 * a.Qpad, 4);
    a.off_pu = a.off_tdesc + ntiles * 8;
    a.off_loc = a.off_pu + 3 * PF * 256;
    a.blob_ints = a.off_loc + 3 * NTQ * 256;
    a.blob = nullptr;
    return OCL_OK;
}

static int plan_conv_q(const ConvGeomDesc& g, ConvPlan* p) {
    // 512-pixel tiles; 256-pixel tiles where the patch of 512 pixels has more units than a workgroup stages (84-pixel-wide rows)
    ConvPlan q = *p;
    bool too_wide = false;
    int r = plan_conv_q_ntq(g, &q, 2, &too_wide);
    if (r != OCL_OK && too_wide) {   // (not where 512-pixel tiles are merely too few: there conv_t_kernel's 64-pixel tiles spread better)
        q = *p;
        r = plan_conv_q_ntq(g, &q, 1);
    }
    if (r == OCL_OK) *p = q;
    return r;
}

// ---- conv_s_kernel plan: (16 NT)-pixel x 16-channel workgroups, input channels split over the four waves --------------------------
static int plan_conv_s_nt(const ConvGeomDesc& g, ConvPlan* p, int NT) {
    ConvArgs& a = p->a;
    if (g.ncls > 1 || g.Cin % 16 || g.Cout % 4) return OCL_ERR_ARG;
    const int LP = g.LH * g.LW, TP = 16 * NT;
    // a tile = TP consecutive lattice pixels of one image (whole rows where the lattice allows: the lane -> pixel map is then the same
    // for every tile and comes from the plan's table), or whole images
    a.n_splits = cdiv(g.Cout, 16);
    a.CoutP = a.n_splits * 16;
    a.group_size = g.N / g.groups;
    if (LP >= TP) {
        a.imgs = 1; a.ppi = TP; a.tiles_per_img = cdiv(LP, TP);
    } else {
        a.imgs = std::min(TP / LP, a.group_size); a.ppi = LP; a.tiles_per_img = 1;
    }
    a.PC = (g.LW - 1) * g.is + (a.max_dx - a.min_dx) + 1;
    const bool whole_rows = TP % g.LW == 0 && LP % TP == 0;
    const int rows_l = LP >= TP ? (whole_rows ? TP / g.LW : std::min(g.LH, (TP + g.LW - 2) / g.LW + 1)) : g.LH;
    a.PR = (rows_l - 1) * g.is + (a.max_dy - a.min_dy) + 1;
    a.C4tot = g.Cin / 4;
    a.KC = g.Cin / 4;                                   // one wave's channel slice
    a.CP = ((a.KC / 4) & 1) ? a.KC : a.KC + 4;
    a.Qc = g.ntaps * (a.KC / 4);
    a.Qpad = (int)round_up(a.Qc, 16);                   // whole loop bodies of four rounds (the padding groups carry zero weights)
    if (16 + 2 * a.Qpad > 512) return OCL_ERR_ARG;
    a.wres = 0; a.pipe = 0; a.QS = a.Qpad;
    const int units = a.imgs * a.PR * a.PC * (a.KC / 4);
    a.nstage = cdiv(units, 64 * kPFS);                  // staging passes of 64 kPFS units per wave
    if (a.nstage > 3 || a.imgs > 127 || a.PR >= 256 || a.PC >= 256 || a.KC / 4 >= 64) return OCL_ERR_ARG;
    a.patch_floats = std::max((int)round_up((int64_t)a.imgs * a.PR * a.PC * a.CP, 4), NT * 64 * 4);   // (>= the partial tiles it holds at the end)
    a.bnb_lds = g.bnb ? 0 : -1;   // (conv_s_kernel reads the BatchNorm's parameters straight from memory: no table)
    a.tiles_per_group = cdiv(a.group_size, a.imgs) * a.tiles_per_img;
    const int ntiles = g.groups * a.tiles_per_group;
    const size_t lds = 64 + (size_t)(2 * a.Qpad + 16) * 4 + (size_t)4 * a.patch_floats * 4 + (g.xf ? (size_t)g.groups * g.Cin * 8 : 0);
    if (lds > 64 * 1024) return OCL_ERR_ARG;
    // Worth it (profiles/r3_conv_s_ab.md) where conv_t_kernel's 64-pixel tiles leave most of the machine idle behind a long K chain:
    // lattices of <= 16 pixels per image (layer 4: a 64-pixel tile is four images, each with its own halo, and K = 720 - 1440 behind
    // every wave) at any batch size; larger lattices (layer 3) below 1000 units of conv_t_kernel work (< 200 images), where that
    // kernel's resident-weight plan takes over (26.9 vs 30.2 us at 220 images).
    const int64_t tiles64 = (int64_t)g.groups * (LP >= 64 ? (int64_t)a.group_size * cdiv(LP, 64) : cdiv(a.group_size, std::max(1, 64 / LP)));
    if (g.force_cs <= 0 && ((LP > 16 && tiles64 * cdiv(g.Cout, 16) >= 1000) || a.Qc <= 36)) return OCL_ERR_ARG;
    a.cls_pack = 1 | (g.ntaps << 4);
    a.cls_oyx = 0;
    p->cs = 1; p->q4 = 0; p->MT = 1; p->NT = NT;
    p->lds_bytes = lds;
    a.WPT = g.WPT > 0 ? g.WPT : a.CoutP;
    for (int t = 0; t < 9; ++t) a.tpo[t] = t < a.ntaps ? ((a.tdy[t] - a.min_dy) * a.PC + (a.tdx[t] - a.min_dx)) * a.CP : 0;
    {   // a wave's 64 lanes walk the patch units
        const int kc4 = a.KC / 4;
        a.d_c4 = 64 % kc4;
        const int d_pix = 64 / kc4;
        a.d_pc = d_pix % a.PC;
        a.d_row = d_pix / a.PC;
    }
    a.groups = g.groups;
    a.aligned = (LP < TP || whole_rows) ? 1 : 0;
    {
        auto magic = [](int d) { return d <= 1 ? 0u : (unsigned)(((1ull << 32) + (unsigned)d - 1) / (unsigned)d); };
        a.m_tpg = a.m_tpi = a.m_kc4 = a.m_pc = a.m_pr = 0;
        a.m_lw = magic(a.LW); a.m_ppi = magic(a.ppi);
        if ((int64_t)(LP + TP) * std::max(a.LW, a.ppi) >= (1ll << 32)) return OCL_ERR_ARG;
    }
    p->grid_x = ntiles;
    p->grid_y = a.n_splits;
    a.off_tdesc = (int)round_up(16 + 2 * a.Qpad, 4);
    a.off_pu = a.off_tdesc + ntiles * 8;
    a.off_loc = a.off_pu + 3 * kPFS * a.nstage * 256;
    a.blob_ints = a.off_loc + 3 * NT * 256;
    a.blob = nullptr;
    return OCL_OK;
}

static int plan_conv_s(const ConvGeomDesc& g, ConvPlan* p) {
    // One pixel tile per workgroup; two (every weight quad feeds two MFMAs: half the weight and table bytes per MFMA, twice the patch per
    // wave) on the 8x8 lattices of layer 3 once the one-tile grid has >= 800 workgroups -- 100 images: 14.6 vs 17.4 us, 150: 21.8 vs
    // 24.1; on layer 4 (two whole images per workgroup) it only pays around 150 images (profiles/r3_conv_s_ab.md).
    static const int env_nt = [] { const char* e = getenv("OCL_CONV_S_NT"); return e ? atoi(e) : 0; }();   // measurement knob: 1 / 2 = always
    ConvPlan q1 = *p;
    const int r1 = plan_conv_s_nt(g, &q1, 1);
    const bool want2 = env_nt ? env_nt == 2 : (r1 == OCL_OK && g.LH * g.LW > 16 && (int64_t)q1.grid_x * q1.grid_y >= 800);
    if (want2) {
        ConvPlan q2 = *p;
        if (plan_conv_s_nt(g, &q2, 2) == OCL_OK) {
            *p = q2;
            return OCL_OK;
        }
    }
    if (r1 == OCL_OK) *p = q1;
    return r1;
}

static int plan_conv_t(const ConvGeomDesc& g, ConvPlan* p) {
    ConvArgs& a = p->a;
    p->cs = 0;
    p->cw = 0;
    {   // wave-autonomous tiles over resident weights (convw.hip): no workgroup barrier between the prologue and the statistics flush
        // OCL_CONV_W: 0 = never, 1 (default) = where its specialised form exists and measured faster (profiles/r6_convw_vs_planner.txt: the 3x3
        // stride-1 convolutions of 40 and 80 channels from ~100 images on), 2 = wherever it fits (A/B reference of tests/test_gpu_ring.py)
        static const int env_cw = [] { const char* e = getenv("OCL_CONV_W"); return e ? atoi(e) : 1; }();
        if (g.force_cw > 0 || (g.force_cw == 0 && env_cw > 0 && !g.force_MT && !g.force_NT && g.force_cs <= 0 && g.force_q4 <= 0)) {
            ConvPlan q = *p;
            if (plan_conv_w(g, &q) == OCL_OK) {
                const int64_t units = (int64_t)(g.N / q.a.imgs) * q.a.tiles_per_img * q.a.n_splits;   // (pixel tile, channel split) units of the launch
                // (not with the input transform: conv_wx_kernel's register pipeline for it is slower than conv_t_kernel's staging -- 33.7 vs 33.3 us
                // on layer 2, 196 us on layer 3: profiles/r6_convw_in_network.txt)
                const bool hot = q.cw == 2 && !g.xf && g.ntaps == 9 && g.is == 1 && (g.Cout == 40 || g.Cout == 80) && g.Cin == g.Cout && units >= 2048;
                if (g.force_cw > 0 || env_cw >= 2 || hot) {
                    *p = q;
                    return OCL_OK;
                }
            }
        }
    }
    {   // few output pixels behind a deep K (layers 3 - 4 of a replay-sized pass): K split over the waves
        static const bool env_cs = [] { const char* e = getenv("OCL_CONV_S"); return !(e && atoi(e) == 0); }();
        if (g.force_cs > 0 || (g.force_cs == 0 && env_cs && !g.force_MT && !g.force_NT)) {
            ConvPlan q = *p;
            if (plan_conv_s(g, &q) == OCL_OK) {
                *p = q;
                return OCL_OK;
            }
        }
    }
    {   // <= 20 output channels: the 4x4x1 form (no channel / K padding) where it fits and the launch is large enough
        static const bool env_q4 = [] { const char* e = getenv("OCL_CONV_Q4"); return !(e && atoi(e) == 0); }();
        if (g.force_q4 > 0 || (g.force_q4 == 0 && env_q4 && !g.force_MT && !g.force_NT)) {
            ConvPlan q = *p;
            if (plan_conv_q(g, &q) == OCL_OK) {
                *p = q;
                return OCL_OK;
            }
        }
    }
    p->q4 = 0;
    const int nt16 = cdiv(g.Cout, 16);
    const int LPx = g.LH * g.LW;
    const int64_t tiles64 = (int64_t)g.groups * (LPx >= 64 ? (int64_t)(g.N / g.groups) * cdiv(LPx, 64)
                                                            : cdiv(g.N / g.groups, std::max(1, 64 / LPx)));
    // channel tiles per workgroup: all of them up to 5 (a lane's A reads are reused NT times, its B reads MT times); fewer when
    // the layer has too few pixel tiles to give every CU a workgroup
    int MT = std::min(5, nt16);
    if (nt16 > 5) MT = cdiv(nt16, cdiv(nt16, 5));                    // balanced splits (10 tiles -> 2 x 5)
    while (MT > 1 && tiles64 * cdiv(nt16, MT) < 200) --MT;   // kbench sweep: 4x55 workgroups of 3 channel tiles beat 5x55 of 2 on layer 4
    if (nt16 > MT) MT = cdiv(nt16, cdiv(nt16, MT));
    // 160 output channels behind a deep K that conv_s_kernel does not take (layer 4 of a 50-image 84x84 pass: 100 pixel tiles): two
    // splits of five channel tiles are 200 workgroups with a 360-round chain each; five splits of two fill the machine twice
    // (kbench sweep, profiles/r3_kbench_sweep_84.txt: 37.7 vs 48.6 us; three or four tiles per workgroup pad 10 tiles to 12)
    if (nt16 == 10 && g.ntaps * g.Cin >= 1280 && tiles64 * 2 < 400) MT = 2;
    int NT = tiles64 * cdiv(nt16, MT) >= 2048 ? 2 : 1;
    if (g.ncls > 1) NT = 1;
    if (g.force_MT) MT = g.force_MT;
    if (g.force_NT) NT = g.force_NT;
    if (MT < 1 || MT > 5 || NT < 1 || NT > 2 || MT > nt16) return OCL_ERR_ARG;
    // staged-weight schedule: the three-buffer ring unless OCL_CONV_PIPE=0 asks for the two-buffer one (plan constant: read once)
    static const bool env_pipe = [] { const char* e = getenv("OCL_CONV_PIPE"); return !(e && atoi(e) == 0); }();
    const bool pipe = g.force_pipe > 0 || (g.force_pipe == 0 && env_pipe);
    size_t lds = convt_layout(g, a, MT, NT, pipe);
    if (!lds && NT == 2 && !g.force_NT) { NT = 1; lds = convt_layout(g, a, MT, NT, pipe); }
    if (!lds && pipe) lds = convt_layout(g, a, MT, NT, false);
    if (!lds) return OCL_ERR_ARG;
    if (a.pipe) {
        // the ring's third buffer can cost the second workgroup per CU; when the launch has more workgroups than CUs that matters more
        // than the schedule (layer 4's merged data gradient at 220 images: 275 workgroups, 36.7 us with two buffers and two workgroups
        // per CU, 40.6 us with the ring and one: profiles/r2_kbench_ring_v2.txt) -- keep the two-buffer plan there
        ConvArgs b = a;
        const size_t lds2 = convt_layout(g, b, MT, NT, false);
        const int bpc_ring = (int)std::min<size_t>(2, kLdsLimit / (lds + 512));
        const int bpc_two = lds2 ? (int)std::min<size_t>(2, kLdsLimit / (lds2 + 512)) : 0;
        const int64_t wgs = (int64_t)g.groups * (cdiv(a.group_size, a.imgs) * a.tiles_per_img) * a.n_splits;
        if (bpc_two > bpc_ring && wgs > 256 * bpc_ring) {
            a = b;
            lds = lds2;
        }
    }
    a.cls_pack = std::max(1, g.ncls);
    a.cls_oyx = 0;
    for (int c = 0; c < std::max(1, g.ncls); ++c) {
        a.cls_pack |= (g.ncls > 1 ? g.cls_ntaps[c] : g.ntaps) << (4 + 4 * c);
        if (g.ncls > 1) a.cls_oyx |= (g.cls_oy[c] << (2 * c)) | (g.cls_ox[c] << (2 * c + 1));
    }
    p->MT = MT; p->NT = NT;
    p->lds_bytes = lds;
    a.WPT = g.WPT > 0 ? g.WPT : a.CoutP;
    for (int t = 0; t < 9; ++t) a.tpo[t] = t < a.ntaps ? ((a.tdy[t] - a.min_dy) * a.PC + (a.tdx[t] - a.min_dx)) * a.CP : 0;
    {
        const int kc4 = a.KC / 4;
        a.d_c4 = 256 % kc4;
        const int d_pix = 256 / kc4;
        a.d_pc = d_pix % a.PC;
        a.d_row = d_pix / a.PC;
    }
    a.groups = g.groups;
    {
        const int BMp = 64 * NT, LPp = g.LH * g.LW;
        a.aligned = a.imgs > 1 ? 1 : ((BMp % g.LW == 0 && LPp % BMp == 0) ? 1 : 0);
    }
    const int ntiles = g.groups * a.tiles_per_group;
    {
        auto magic = [](int d) { return d <= 1 ? 0u : (unsigned)(((1ull << 32) + (unsigned)d - 1) / (unsigned)d); };
        a.m_tpg = magic(a.tiles_per_group); a.m_tpi = magic(a.tiles_per_img); a.m_lw = magic(a.LW); a.m_ppi = magic(a.ppi);
        a.m_kc4 = magic(a.KC / 4); a.m_pc = magic(a.PC); a.m_pr = magic(a.PR);
        // exactness of x / d by one multiply-high needs x * d < 2^32: the largest dividends are tile and pixel indices
        const int64_t xmax = std::max<int64_t>(std::max<int64_t>(ntiles, (int64_t)g.LH * g.LW + 64 * NT), 4096);
        const int64_t dmax = std::max(std::max(a.tiles_per_group, a.tiles_per_img), std::max(std::max(a.LW, a.ppi), std::max(a.PC, a.PR)));
        if (xmax * dmax >= (1ll << 32)) return OCL_ERR_ARG;
    }
    int bpc = (int)std::min<size_t>(2, kLdsLimit / (lds + 512));
    if (g.force_bpc) bpc = g.force_bpc;
    p->grid_x = std::max(1, std::min(ntiles, (256 * std::max(1, bpc)) / a.n_splits));
    p->grid_x = std::max(p->grid_x, cdiv(ntiles, kMaxWgTiles));   // a workgroup keeps at most kMaxWgTiles tile descriptors
    p->grid_y = a.n_splits;
    // layout of the plan's device tables (conv_plan_tables)
    if (16 + 2 * a.Qpad > 768) return OCL_ERR_ARG;   // the prologue copies the group tables with three predicated loads per thread
    const int PF = convt_pf_for(a.imgs * a.PR * a.PC * (a.KC / 4));
    a.off_tdesc = (int)round_up(16 + 2 * a.Qpad, 4);
    a.off_pu = a.off_tdesc + ntiles * 8;
    a.off_loc = a.off_pu + 3 * PF * 256;
    a.blob_ints = a.off_loc + 3 * NT * 256;
    a.blob = nullptr;
    return OCL_OK;
}

// ---- the plan's tables: every value the kernel's prologue used to compute per workgroup and per launch -------------------------
void conv_plan_tables(const ConvPlan& p, std::vector<int>* out) {
    if (p.cw) {
        conv_w_tables(p, out);
        return;
    }
    const ConvArgs& a = p.a;
    const int NT = p.NT, MT = p.MT;
    (void)MT;
    const int kc4 = a.KC / 4;
    const int ncls = a.cls_pack & 15;
    const bool pipe = a.pipe != 0, res = a.wres != 0;
    std::vector<int>& b = *out;
    b.assign((size_t)a.blob_ints, 0);
    int* ctab = b.data();
    int* qoff = ctab + 16;
    int* qrow = qoff + a.Qpad;
    // K groups: class by class, each class padded to whole rounds of 4 groups
    {
        int q0 = 0, t0 = 0;
        for (int c = 0; c < ncls; ++c) {
            const int ntc = (a.cls_pack >> (4 + 4 * c)) & 15, nq = (ntc * kc4 + 3) & ~3;
            for (int ql = 0; ql < nq; ++ql) {
                const int q = q0 + ql;
                const bool ok = ql < ntc * kc4;
                const int t = t0 + ql / kc4, c4 = ql % kc4;
                qoff[q] = ok ? a.tpo[t] + 4 * c4 : 0;
                // ring: the BYTE offset of the pack row (bit 31 = past every buffer descriptor: such a load returns zeros)
                qrow[q] = ok ? (a.tw[t] * a.C4tot + c4) * (pipe ? a.WPT * 16 : 1) : (pipe ? (int)0x80000000 : -1);
            }
            if (ncls > 1) {
                const int oy = (a.cls_oyx >> (2 * c)) & 1, ox = (a.cls_oyx >> (2 * c + 1)) & 1;
                ctab[c * 4 + 0] = q0;
                ctab[c * 4 + 1] = nq;
                ctab[c * 4 + 2] = (oy * a.Wout + ox) * a.Cout;
                ctab[c * 4 + 3] = res ? 1 : (nq + a.QS - 1) / a.QS;
            }
            q0 += nq; t0 += ntc;
        }
        for (int q = q0; q < a.Qpad; ++q) {   // (conv_s_kernel pads to whole loop bodies: groups that load no weights)
            qoff[q] = 0;
            qrow[q] = pipe ? (int)0x80000000 : -1;
        }
        if (ncls > 1)
            for (int c = ncls; c < 4; ++c) ctab[c * 4 + 0] = q0;   // (classes past the last: first group = end, no groups)
    }
    // tile descriptors
    const int LP = a.LH * a.LW;
    const int ntiles = a.groups * a.tiles_per_group;
    int* td = b.data() + a.off_tdesc;
    for (int tile = 0; tile < ntiles; ++tile) {
        const int grp = tile / a.tiles_per_group, tg_ = tile % a.tiles_per_group;
        const int ti = tg_ / a.tiles_per_img, tp = tg_ % a.tiles_per_img;
        const int img0 = grp * a.group_size + ti * a.imgs;
        const int p0 = tp * a.ppi;
        const int grp_end = std::min(a.N, (grp + 1) * a.group_size);
        const int ly0 = p0 / a.LW;
        const int pend = std::min(p0 + a.ppi, LP);
        const int ly1 = (pend - 1) / a.LW;
        const int nimg = std::min(a.imgs, grp_end - img0);
        const int nrows = a.imgs > 1 ? nimg * a.PR : (ly1 - ly0) * a.is + (a.max_dy - a.min_dy) + 1;
        const int iy0 = ly0 * a.is + a.min_dy;
        int* d = td + (size_t)tile * 8;
        d[0] = (((img0 * a.Hin + iy0) * a.Win + a.min_dx) * a.Cin) * 4;                // input byte offset of the patch origin
        d[1] = iy0;
        d[2] = nrows;
        d[3] = ((img0 * a.Hout + ly0 * a.os + a.oy0) * a.Wout + a.ox0) * a.Cout;         // output element offset of the tile origin
        d[4] = nimg;
        d[5] = grp;
        d[6] = p0;
        d[7] = img0 | (ly0 << 20);
    }
    // per-thread patch units: unit u = tid + i * 256 of the flat [row][pc][c4] patch
    const int PF = (a.off_loc - a.off_pu) / (3 * 256);
    int* pu = b.data() + a.off_pu;
    for (int tid = 0; tid < (p.cs ? 64 : 256); ++tid) {   // (conv_s_kernel: a wave's 64 lanes walk the units, stride 64)
        const int pix = tid / kc4;
        int c4 = tid % kc4, row = pix / a.PC, pc = pix % a.PC;
        for (int i = 0; i < PF; ++i) {
            int il = 0, pr = row;
            if (a.imgs > 1) { il = row / a.PR; pr = row % a.PR; }
            const int ix = a.min_dx + pc;
            const bool xok = ix >= 0 && ix < a.Win;               // columns of the halo outside the image: zeros (never loaded, still stored)
            pu[(3 * i + 0) * 256 + tid] = xok ? (((il * a.Hin + pr) * a.Win + pc) * a.Cin + c4 * 4) * 4 : -1;
            pu[(3 * i + 1) * 256 + tid] = (row * a.PC + pc) * a.CP + c4 * 4;
            pu[(3 * i + 2) * 256 + tid] = (il < 128 && pr < 256) ? (row | (pr << 16) | (c4 << 24)) : 0x7fff;   // row 0x7fff: past every tile's last row; bits 24+: channel quad
            c4 += a.d_c4;
            pc += a.d_pc;
            if (c4 >= kc4) { c4 -= kc4; pc += 1; }
            row += a.d_row;
            if (pc >= a.PC) { pc -= a.PC; row += 1; }
        }
    }
    // per-lane output pixels relative to the tile origin
    int* lc = b.data() + a.off_loc;
    for (int tid = 0; tid < 256; ++tid) {
        const int wave = tid >> 6, r16 = tid & 15, lane = tid & 63;
        for (int nt = 0; nt < NT; ++nt) {
            // conv_t_kernel: 16-pixel tiles, the lane's pixel = its r16; conv_q_kernel: 64-pixel sets, one pixel per lane
            // conv_s_kernel: every wave holds the same 16 NT pixels
            const int r = p.cs ? nt * 16 + r16 : p.q4 ? wave * 64 * NT + nt * 64 + lane : wave * 16 * NT + nt * 16 + r16;
            const int il = r / a.ppi, pl = r % a.ppi;
            const int ly = pl / a.LW, lx = pl % a.LW;
            lc[(3 * nt + 0) * 256 + tid] = ((il * a.PR + ly * a.is) * a.PC + lx * a.is) * a.CP;
            lc[(3 * nt + 1) * 256 + tid] = ((il * a.Hout + ly * a.os) * a.Wout + lx * a.os) * a.Cout;
            lc[(3 * nt + 2) * 256 + tid] = il;
        }
    }
}

int conv_plan_finalize(ConvPlan* p, PlanArena* arena, hipStream_t s) {
    if (p->a.blob) return OCL_OK;
    if (!arena) {
        std::vector<int> t;
        conv_plan_tables(*p, &t);
        int* d = nullptr;
        OCL_HIP(hipMalloc((void**)&d, t.size() * sizeof(int)));
        OCL_HIP(hipMemcpy(d, t.data(), t.size() * sizeof(int), hipMemcpyHostToDevice));
        p->a.blob = d;
        return OCL_OK;
    }
    std::vector<int>* t = new std::vector<int>();
    arena->host_keep.push_back(t);
    conv_plan_tables(*p, t);
    const size_t bytes = (size_t)round_up((int64_t)t->size() * sizeof(int), 256);
    if (arena->chunks.empty() || arena->used + bytes > arena->cap) {
        const size_t cap = std::max<size_t>(8u << 20, bytes);
        void* c = nullptr;
        OCL_HIP(hipMalloc(&c, cap));
        {
            static const bool log_plans = [] { const char* e = getenv("OCL_LOG_PLANS"); return e && e[0] == '1'; }();
            if (log_plans) fprintf(stderr, "[ocl] plan-table arena: chunk %zu allocated (%zu bytes)\n", arena->chunks.size() + 1, cap);
        }
        arena->chunks.push_back(c);
        arena->used = 0;
        arena->cap = cap;
    }
    int* d = (int*)((char*)arena->chunks.back() + arena->used);
    arena->used += bytes;
    OCL_HIP(hipMemcpyAsync(d, t->data(), t->size() * sizeof(int), hipMemcpyHostToDevice, s));
    hipEvent_t e;
    OCL_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    arena->events.push_back(e);
    OCL_HIP(hipEventRecord(e, s));
    p->ready = e;
    p->ready_stream = s;
    p->a.blob = d;
    return OCL_OK;
}
void plan_arena_release(PlanArena* a) {
    for (hipEvent_t e : a->events) (void)hipEventDestroy(e);
    a->events.clear();
    for (void* c : a->chunks) (void)hipFree(c);
    for (auto* v : a->host_keep) delete v;
    a->chunks.clear();
    a->host_keep.clear();
    a->used = a->cap = 0;
}
void conv_plan_release(ConvPlan* p) {
    if (p->a.blob) (void)hipFree((void*)p->a.blob);
    p->a.blob = nullptr;
}

int plan_conv(const ConvGeomDesc& g, ConvPlan* p) {
    memset(p, 0, sizeof(*p));
    ConvArgs& a = p->a;
    OCL_REQUIRE(g.N > 0 && g.groups > 0 && g.N % g.groups == 0, "plan_conv: N=%d not divisible into %d groups", g.N, g.groups);
    OCL_REQUIRE(g.Cin % 4 == 0 && g.Cout % 4 == 0 && g.ntaps >= 1 && g.ntaps <= 9, "plan_conv: Cin=%d Cout=%d ntaps=%d", g.Cin,
                g.Cout, g.ntaps);
    OCL_REQUIRE(g.LH > 0 && g.LW > 0, "plan_conv: empty lattice");
    a.N = g.N; a.Hin = g.Hin; a.Win = g.Win; a.Cin = g.Cin;
    a.Hout = g.Hout; a.Wout = g.Wout; a.Cout = g.Cout;
    a.LH = g.LH; a.LW = g.LW; a.os = g.os; a.oy0 = g.oy0; a.ox0 = g.ox0; a.is = g.is;
    a.ntaps = g.ntaps;
    a.min_dy = a.min_dx = 1 << 20;
    a.max_dy = a.max_dx = -(1 << 20);
    for (int t = 0; t < g.ntaps; ++t) {
        a.tdy[t] = g.tdy[t]; a.tdx[t] = g.tdx[t]; a.tw[t] = g.tw[t];
        a.min_dy = std::min(a.min_dy, g.tdy[t]); a.max_dy = std::max(a.max_dy, g.tdy[t]);
        a.min_dx = std::min(a.min_dx, g.tdx[t]); a.max_dx = std::max(a.max_dx, g.tdx[t]);
    }
    if (plan_conv_t(g, p) != OCL_OK) {
        set_error("plan_conv: no tiling fits (Hin=%d Win=%d Cin=%d Cout=%d taps=%d classes=%d, forced MT=%d NT=%d)", g.Hin, g.Win, g.Cin, g.Cout,
                  g.ntaps, g.ncls, g.force_MT, g.force_NT);
        return OCL_ERR_ARG;
    }
    return OCL_OK;
}

int pack_width(int channels) { return (int)round_up(channels, 16); }

void geom_fwd(const ConvShape& c, int N, int groups, ConvGeomDesc* g) {
    memset(g, 0, sizeof(*g));
    g->N = N; g->groups = groups;
    g->Hin = c.Hin; g->Win = c.Win; g->Cin = c.CinT;
    g->Hout = c.Ho; g->Wout = c.Wo; g->Cout = c.Cout;
    g->LH = c.Ho; g->LW = c.Wo; g->os = 1; g->oy0 = 0; g->ox0 = 0; g->is = c.stride;
    g->WPT = c.CoutP;
    const int pad = c.k == 3 ? 1 : 0;
    g->ntaps = c.k * c.k;
    for (int t = 0; t < g->ntaps; ++t) {
        g->tdy[t] = t / c.k - pad;
        g->tdx[t] = t % c.k - pad;
        g->tw[t] = t;
    }
}

void geom_dgrad(const ConvShape& c, int N, std::vector<ConvGeomDesc>* out, bool merge_classes, int groups) {
    out->clear();
    ConvGeomDesc g;
    memset(&g, 0, sizeof(g));
    g.N = N; g.groups = groups;
    g.Hin = c.Ho; g.Win = c.Wo; g.Cin = c.Cout;
    g.Hout = c.Hin; g.Wout = c.Win; g.Cout = c.Cin;
    g.is = 1;
    g.WPT = c.CiP;
    if (c.stride == 1) {
        g.LH = c.Hin; g.LW = c.Win; g.os = 1;
        const int pad = c.k == 3 ? 1 : 0;
        g.ntaps = c.k * c.k;
        for (int t = 0; t < g.ntaps; ++t) {
            g.tdy[t] = pad - t / c.k;
            g.tdx[t] = pad - t % c.k;
            g.tw[t] = t;
        }
        out->push_back(g);
    } else if (c.k == 1) {  // 1x1 stride 2, pad 0: only even pixels receive gradient
        g.os = 2; g.oy0 = 0; g.ox0 = 0;
        g.LH = (c.Hin + 1) / 2; g.LW = (c.Win + 1) / 2;
        g.ntaps = 1;
        g.tdy[0] = 0; g.tdx[0] = 0; g.tw[0] = 0;
        out->push_back(g);
    } else if (merge_classes && c.Hin % 2 == 0 && c.Win % 2 == 0) {
        // the four parity classes as output classes of one launch: they read the same dy window (rows / columns +0, +1), use
        // disjoint taps (1, 2, 2 and 4 of the 9) and write the four interleaved lattices of dx
        g.os = 2; g.oy0 = 0; g.ox0 = 0;
        g.LH = c.Hin / 2; g.LW = c.Win / 2;
        g.ncls = 4;
        int nt = 0;
        for (int py = 0; py < 2; ++py)
            for (int px = 0; px < 2; ++px) {
                const int cls = py * 2 + px;
                g.cls_oy[cls] = py; g.cls_ox[cls] = px;
                const int first = nt;
                for (int ky = 0; ky < 3; ++ky) {
                    if (((py + 1 - ky) & 1) != 0) continue;
                    for (int kx = 0; kx < 3; ++kx) {
                        if (((px + 1 - kx) & 1) != 0) continue;
                        g.tdy[nt] = (py + 1 - ky) / 2;
                        g.tdx[nt] = (px + 1 - kx) / 2;
                        g.tw[nt] = ky * 3 + kx;
                        ++nt;
                    }
                }
                g.cls_ntaps[cls] = nt - first;
            }
        g.ntaps = nt;
        out->push_back(g);
    } else {  // 3x3 stride 2 pad 1: four dense parity classes of the dx lattice
        for (int py = 0; py < 2; ++py)
            for (int px = 0; px < 2; ++px) {
                ConvGeomDesc q = g;
                q.os = 2; q.oy0 = py; q.ox0 = px;
                q.LH = (c.Hin - py + 1) / 2; q.LW = (c.Win - px + 1) / 2;
                if (q.LH <= 0 || q.LW <= 0) continue;
                int nt = 0;
                for (int ky = 0; ky < 3; ++ky) {
                    if (((py + 1 - ky) & 1) != 0) continue;
                    for (int kx = 0; kx < 3; ++kx) {
                        if (((px + 1 - kx) & 1) != 0) continue;
                        q.tdy[nt] = (py + 1 - ky) / 2;  // exact: even numerator
                        q.tdx[nt] = (px + 1 - kx) / 2;
                        q.tw[nt] = ky * 3 + kx;
                        ++nt;
                    }
                }
                q.ntaps = nt;
                out->push_back(q);
            }
    }
}

static conv_fn_t convq_trace_fn(int ntq, int pf, int stats) {   // measurement builds: the 220-view plans of layer 1
    if (ntq != 2 || pf != 12) return nullptr;
    return stats == 2 ? conv_q_kernel<2, 12, 2, 1> : stats == 1 ? conv_q_kernel<2, 12, 1, 1> : conv_q_kernel<2, 12, 0, 1>;
}
static conv_fn_t convq_fn(int ntq, int pf, int stats) {   // stats: 0 none, 1 EPI_STATS, 2 EPI_BNB
#define OCL_CASE(N, P)                                                                                                         \
    if (ntq == N && pf == P) return stats == 2 ? conv_q_kernel<N, P, 2> : stats == 1 ? conv_q_kernel<N, P, 1> : conv_q_kernel<N, P, 0>;
    OCL_CASE(2, 4) OCL_CASE(2, 12) OCL_CASE(1, 12)
#undef OCL_CASE
    return nullptr;
}

int launch_conv(const ConvPlan& p, hipStream_t s) {
    if (p.cw) return launch_conv_w(p, s);
    if (p.cs) {
        if (!p.a.blob) {
            set_error("launch_conv: plan without device tables (conv_plan_finalize)");
            return OCL_ERR_STATE;
        }
        ProfScope ps(PROF_CONV, s);
        const int det = det_host();
        hipLaunchKernelGGL(convs_fn(p.NT, p.a.trace != nullptr && !det, (p.a.flags & EPI_BNB) != 0, det != 0), dim3(p.grid_x, p.grid_y), dim3(256),
                           p.lds_bytes, s, p.a);
        OCL_LAUNCH_CHECK();
        return OCL_OK;
    }
    if (p.q4) {
        conv_fn_t fq = convq_fn(p.q4, (p.a.off_loc - p.a.off_pu) / (3 * 256), (p.a.flags & EPI_BNB) ? 2 : (p.a.flags & EPI_STATS) ? 1 : 0);
        if (p.a.trace)
            if (conv_fn_t ft = convq_trace_fn(p.q4, (p.a.off_loc - p.a.off_pu) / (3 * 256), (p.a.flags & EPI_BNB) ? 2 : (p.a.flags & EPI_STATS) ? 1 : 0)) {
                fq = ft;
                OCL_HIP(hipFuncSetAttribute((const void*)fq, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit));
            }
        if (!fq || !p.a.blob) {
            set_error("launch_conv: no conv_q_kernel for q4=%d / plan without device tables", p.q4);
            return OCL_ERR_STATE;
        }
        ProfScope ps(PROF_CONV, s);
        hipLaunchKernelGGL(fq, dim3(p.grid_x, p.grid_y), dim3(256), p.lds_bytes, s, p.a);
        OCL_LAUNCH_CHECK();
        return OCL_OK;
    }
    conv_fn_t fn = convt_fn(p.MT, p.NT, convt_pf_for(p.a.imgs * p.a.PR * p.a.PC * (p.a.KC / 4)), p.a.wres, (p.a.cls_pack & 15) > 1, p.a.pipe,
                            (p.a.flags & EPI_BNB) ? 1 : 0);
    if (!fn) {
        set_error("launch_conv: no kernel for MT=%d NT=%d", p.MT, p.NT);
        return OCL_ERR_STATE;
    }
    if (!p.a.blob) {
        set_error("launch_conv: plan without device tables (conv_plan_finalize)");
        return OCL_ERR_STATE;
    }
    ProfScope ps(PROF_CONV, s);
    hipLaunchKernelGGL(fn, dim3(p.grid_x, p.grid_y), dim3(256), p.lds_bytes, s, p.a);
    OCL_LAUNCH_CHECK();
    return OCL_OK;
}


// =====================================================================================================
// weight packing (all conv layers in one launch)
// =====================================================================================================
__global__ void __launch_bounds__(256) pack_weights_kernel(const float* __restrict__ params, float* __restrict__ arena,
                                                           const PackDesc* __restrict__ descs, int mask, int n_layers, StatCell* __restrict__ zero_a,
                                                           int64_t zero_a_n, StatCell* __restrict__ zero_b, int64_t zero_b_n) {
    if ((int)blockIdx.y >= n_layers) {   // the last grid row clears the statistics arenas of the pass (saves two memset launches)
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < zero_a_n + zero_b_n; i += (int64_t)gridDim.x * blockDim.x) {
            StatCell z;
            z.lo = 0ull; z.hi = 0ll;
            if (i < zero_a_n) zero_a[i] = z;
            else zero_b[i - zero_a_n] = z;
        }
        return;
    }
    PackDesc d = descs[blockIdx.y];
    // a pass writes only the packs it reads (PACK_* bits).  The threads walk the PACKS in storage order -- rows of Cout x 4 (forward) /
    // Cin x 4 (data gradient) consecutive floats, coalesced stores -- and gather from the OIHW tensor (read-only, 36-byte strides: served by
    // L2); walking the tensor and scattering 4-byte stores into both packs was 13 us at the head of every step's chain.  Padding rows /
    // columns of a pack are never written (zero since the arena was created).
    if (!(mask & PACK_TF)) d.tf_off = -1;
    if (!(mask & PACK_TD)) d.td_off = -1;
    const int ci4n = (d.Cin + 3) >> 2;
    const int nF = d.tf_off >= 0 ? d.ntaps * ci4n * d.Cout * 4 : 0;
    const int nD = d.td_off >= 0 ? d.ntaps * d.Cout * d.Cin : 0;   // (Cout is a multiple of 4)
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < nF + nD; e += gridDim.x * blockDim.x) {
        if (e < nF) {   // [t][ci >> 2][co][ci & 3]
            const int k = e & 3, r = e >> 2;
            const int co = r % d.Cout, r2 = r / d.Cout;
            const int c4 = r2 % ci4n, t = r2 / ci4n;
            const int ci = c4 * 4 + k;
            if (ci < d.Cin)
                arena[d.tf_off + ((((int64_t)t * (d.CinP >> 2) + c4) * d.CoutP + co) << 2) + k] = params[d.w_off + ((int64_t)co * d.Cin + ci) * d.ntaps + t];
        } else {        // [t][co >> 2][ci][co & 3]
            const int f = e - nF;
            const int k = f & 3, r = f >> 2;
            const int ci = r % d.Cin, r2 = r / d.Cin;
            const int o4 = r2 % (d.Cout >> 2), t = r2 / (d.Cout >> 2);
            const int co = o4 * 4 + k;
            arena[d.td_off + ((((int64_t)t * (d.Cout >> 2) + o4) * d.CiP + ci) << 2) + k] = params[d.w_off + ((int64_t)co * d.Cin + ci) * d.ntaps + t];
        }
    }
}

int launch_pack_weights(const float* params, float* arena, const PackDesc* descs_dev, int n_layers, int max_elems, hipStream_t s,
                        int mask, StatCell* zero_a, int64_t zero_a_n, StatCell* zero_b, int64_t zero_b_n) {
    ProfScope ps(PROF_BN, s);
    const int extra = (zero_a_n + zero_b_n) > 0 ? 1 : 0;
    // (up to 256 workgroups per layer: layer 4's 230 k weights in 4 passes per thread instead of 14)
    hipLaunchKernelGGL(pack_weights_kernel, dim3(std::min(512, cdiv(2 * max_elems, 256)), n_layers + extra), dim3(256), 0, s, params, arena,
                       descs_dev, mask, n_layers, zero_a, zero_a_n, zero_b, zero_b_n);
    OCL_LAUNCH_CHECK();
    return OCL_OK;
}

// =====================================================================================================
// layout
// =====================================================================================================
__global__ void __launch_bounds__(256) nchw3_to_nhwc4_kernel(const float* __restrict__ x, float4* __restrict__ out, int HW,
                                                             int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = i / HW;
        const int p = (int)(i - n * HW);
        const float* b = x + n * 3 * HW + p;
        out[i] = make_float4(b[0], b[HW], b[2 * (int64_t)HW], 0.f);
    }
}
int launch_nchw3_to_nhwc4(const float* x, float* out, int N, int H, int W, hipStream_t s) {
    const int64_t total = (int64_t)N * H * W;
    ProfScope ps(PROF_BN, s);
    hipLaunchKernelGGL(nchw3_to_nhwc4_kernel, dim3((unsigned)std::min<int64_t>(2048, (total + 255) / 256)), dim3(256), 0, s, x,
                       (float4*)out, H * W, total);
    OCL_LAUNCH_CHECK();
    return OCL_OK;
}
// the same from up to kMaxInputSegments separate [n_i, 3, H, W] tensors that together form the batch (memory rows + stream batch +
// augmented views: the reference's torch.cat((mem_x, batch_x)) and the per-view forward calls, without materialising the concatenation)
__global__ void __launch_bounds__(256) nchw3_to_nhwc4_seg_kernel(const InputSegments sg, float4* __restrict__ out, int HW, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int n = (int)(i / HW);
        const int p = (int)(i - (int64_t)n * HW);
        int k = 0;
#pragma unroll
        for (int j = 1; j < kMaxInputSegments; ++j) k = (j < sg.n && n >= sg.first[j]) ? j : k;
        const float* xs = sg.x[0];
#pragma unroll
        for (int j = 1; j < kMaxInputSegments; ++j) xs = k == j ? sg.x[j] : xs;
        int f = sg.first[0];
#pragma unroll
        for (int j = 1; j < kMaxInputSegments; ++j) f = k == j ? sg.first[j] : f;
        const float* b = xs + (int64_t)(n - f) * 3 * HW + p;
        out[i] = make_float4(b[0], b[HW], b[2 * (int64_t)HW], 0.f);
    }
}
int launch_nchw3_to_nhwc4_segments(const InputSegments& sg, float* out, int N, int H, int W, hipStream_t s) {
    if (sg.n == 1) return launch_nchw3_to_nhwc4(sg.x[0], out, N, H, W, s);
    const int64_t total = (int64_t)N * H * W;
    ProfScope ps(PROF_BN, s);
    hipLaunchKernelGGL(nchw3_to_nhwc4_seg_kernel, dim3((unsigned)std::min<int64_t>(2048, (total + 255) / 256)), dim3(256), 0, s, sg,
                       (float4*)out, H * W, total);
    OCL_LAUNCH_CHECK();
    return OCL_OK;
}

// =====================================================================================================
// BatchNorm forward (train mode): normalise + optional residual + ReLU; block (0,0) updates running stats
// (nn.BatchNorm2d: biased variance to normalise, unbiased for the running update, momentum 0.1)
// =====================================================================================================
__global__ void __launch_bounds__(256) bn_fwd_kernel(const BnFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* sc = sm;
    float* sh = sm + a.C;
    const int g = blockIdx.y, tid = threadIdx.x;
    const double M = (double)a.m_per_group;
    for (int c = tid; c < a.C; c += 256) {
        double mean, var;
        bn_batch_moments(a.stats, a.stat_rep_stride, g, c, a.C, M, a.eps, mean, var);
        if (a.frozen_mean) {   // eval-mode BatchNorm on the tape: the running statistics, folded exactly as bn_fold_kernel does
            mean = (double)a.frozen_mean[c];
            var = (double)a.frozen_var[c];
        }
        const double invstd = 1.0 / sqrt(var + (double)a.eps);
        bn_scale_shift(a.gamma[c], a.beta[c], (float)mean, (float)invstd, sc[c], sh[c]);
        if (blockIdx.x == 0) {
            a.save_mean[(int64_t)g * a.C + c] = (float)mean;
            a.save_invstd[(int64_t)g * a.C + c] = (float)invstd;
        }
    }
    if (blockIdx.x == 0 && g == 0 && a.running_mean)
        bn_running_update(a.stats, a.stat_rep_stride, a.G, a.C, M, a.momentum, a.eps, a.running_mean, a.running_var, a.nbt, tid, 256);
    float* scb = sm + 2 * a.C;
    float* shb = sm + 3 * a.C;
    if (a.yb) {   // the projection shortcut's BatchNorm: the same table, statistics and running update from its own arena
        for (int c = tid; c < a.C; c += 256) {
            double mean, var;
            bn_batch_moments(a.stats_b, a.stat_rep_stride, g, c, a.C, M, a.eps, mean, var);
            if (a.frozen_mean_b) {
                mean = (double)a.frozen_mean_b[c];
                var = (double)a.frozen_var_b[c];
            }
            const double invstd = 1.0 / sqrt(var + (double)a.eps);
            bn_scale_shift(a.gamma_b[c], a.beta_b[c], (float)mean, (float)invstd, scb[c], shb[c]);
            if (blockIdx.x == 0) {
                a.save_mean_b[(int64_t)g * a.C + c] = (float)mean;
                a.save_invstd_b[(int64_t)g * a.C + c] = (float)invstd;
            }
        }
        if (blockIdx.x == 0 && g == 0 && a.running_mean_b)
            bn_running_update(a.stats_b, a.stat_rep_stride, a.G, a.C, M, a.momentum, a.eps, a.running_mean_b, a.running_var_b, a.nbt_b, tid, 256);
    }
    __syncthreads();
    const int C4 = a.C >> 2;
    const int64_t units = a.m_per_group * C4;
    const float4* y4 = (const float4*)a.y + (int64_t)g * units;
    const float4* b4 = a.yb ? (const float4*)a.yb + (int64_t)g * units : nullptr;
    const float4* r4 = a.res ? (const float4*)a.res + (int64_t)g * units : nullptr;
    float4* z4 = (float4*)a.z + (int64_t)g * units;
    for (int64_t u = (int64_t)blockIdx.x * 256 + tid; u < units; u += (int64_t)gridDim.x * 256) {
        const int c = (int)(u % C4) * 4;
        float4 v = y4[u];
        v.x = __fmaf_rn(v.x, sc[c], sh[c]);
        v.y = __fmaf_rn(v.y, sc[c + 1], sh[c + 1]);
        v.z = __fmaf_rn(v.z, sc[c + 2], sh[c + 2]);
        v.w = __fmaf_rn(v.w, sc[c + 3], sh[c + 3]);
        if (r4) {
            const float4 r = r4[u];
            v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
        }
        if (b4) {
            float4 r = b4[u];
            r.x = __fmaf_rn(r.x, scb[c], shb[c]);
            r.y = __fmaf_rn(r.y, scb[c + 1], shb[c + 1]);
            r.z = __fmaf_rn(r.z, scb[c + 2], shb[c + 2]);
            r.w = __fmaf_rn(r.w, scb[c + 3], shb[c + 3]);
            v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
        }
        if (a.relu) {
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        z4[u] = v;
    }
}

int launch_bn_fwd(const BnFwdArgs& a, hipStream_t s) {
    const int64_t units = a.m_per_group * (a.C / 4);
    const int bx = (int)std::max<int64_t>(1, std::min<int64_t>(1024, (units + 1023) / 1024));
    ProfScope ps(PROF_BN, s);
    hipLaunchKernelGGL(bn_fwd_kernel, dim3(bx, a.G), dim3(256), (size_t)a.C * (a.yb ? 16 : 8), s, a);
    OCL_LAUNCH_CHECK();
    return OCL_OK;
}

// z = relu(fma(y, scale, shift)) with scale / shift from SAVED statistics: materialises the activation a fused pass never wrote
__global__ void __launch_bounds__(256) bn_apply_saved_kernel(const float* __restrict__ y, const float* __restrict__ mean, const float* __restrict__ invstd,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ z,
                                                             int64_t m_per_group, int C) {
    const int g = blockIdx.y, C4 = C >> 2;
    const int64_t units = m_per_group * C4;
    for (int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x; u < units; u += (int64_t)gridDim.x * 256) {
        const int c = (int)(u % C4) * 4;
        float4 v = ((const float4*)y)[(int64_t)g * units + u];
        float o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float sc, sh;
            bn_scale_shift(gamma[c + e], beta[c + e], mean[(int64_t)g * C + c + e], invstd[(int64_t)g * C + c + e], sc, sh);
            o[e] = fmaxf(__fmaf_rn(o[e], sc, sh), 0.f);
        }
        ((float4*)z)[(int64_t)g * units + u] = make_float4(o[0], o[1], o[2], o[3]);
    }
}
int launch_bn_apply_saved(const float* y, const float* mean, const float* invstd, const float* gamma, const float* beta, float* z,
                          int64_t m_per_group, int G, int C, hipStream_t s) {
    const int64_t units = m_per_group * (C / 4);
    const int bx = (int)std::max<int64_t>(1, std::min<int64_t>(1024, (units + 1023) / 1024));
    hipLaunchKernelGGL(bn_apply_saved_kernel, dim3(bx, G), dim3(256), 0, s, y, mean, invstd, gamma, beta, z, m_per_group, C);
    OCL_LAUNCH_CHECK();
    return OCL_OK;
}

__global__ void __launch_bounds__(256) bn_fold_kernel(const float* __restrict__ params, const float* __restrict__ running,
                                                      float* __restrict__ out, const BnFoldDesc* __restrict__ descs, float eps) {
    const BnFoldDesc d = descs[blockIdx.x];
    for (int c = threadIdx.x; c < d.C; c += 256) {
        const float rm = running[d.stat_off + c], rv = running[d.stat_off + d.C + c];
        const float invstd = (float)(1.0 / sqrt((double)rv + (double)eps));
        const float scale = params[d.gamma_off + c] * invstd;
        out[d.out_off + c] = scale;
        out[d.out_off + d.C + c] = params[d.beta_off + c] - rm * scale;
    }
}
int launch_bn_fold(const float* params, const float* running, float* out, const BnFoldDesc* descs_dev, int n_bn, float eps,
                   hipStream_t s) {
    ProfScope ps(PROF_BN, s);
    hipLaunchKernelGGL(bn_fold_kernel, dim3(n_bn), dim3(256), 0, s, params, running, out, descs_dev, eps);
    OCL_LAUNCH_CHECK();
    return OCL_OK;
}

// =====================================================================================================
// BatchNorm backward (+ReLU mask), one or two BNs sharing the incoming gradient
// =====================================================================================================
// One block walks a contiguous pixel range: thread t < PT*C4 owns (pixel lane t / C4, channel quad t % C4), so one pass of the block
// reads PT*C4 consecutive float4s of each tensor.  U passes are loaded before any is consumed (3U 16-byte loads in flight per lane).
template <int U>
__global__ void __launch_bounds__(256) bn_bwd_reduce_kernel(const BnBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int C4 = a.C >> 2;
    const int PT = 256 / C4;  // pixel lanes
    const int tid = threadIdx.x;
    const int c4 = tid % C4, pl = tid / C4;
    const int g = blockIdx.y;
    const int64_t M = a.m_per_group;
    const int64_t per = (M + gridDim.x - 1) / gridDim.x;
    const int64_t pbeg = (int64_t)blockIdx.x * per, pend = min(M, pbeg + per);
    float4 sd[2], sx[2];
    float4 mean[2], istd[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        sd[k] = sx[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        mean[k] = istd[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (pl < PT) {
#pragma unroll
        for (int k = 0; k < 2; ++k)
            if (k < a.nsets) {
                mean[k] = *(const float4*)(a.mean[k] + (int64_t)g * a.C + c4 * 4);
                istd[k] = *(const float4*)(a.invstd[k] + (int64_t)g * a.C + c4 * 4);
            }
        const float4* dz4 = (const float4*)a.dz + (int64_t)g * M * C4;
        const float4* z4 = a.z ? (const float4*)a.z + (int64_t)g * M * C4 : nullptr;
        const float4* y40 = (const float4*)a.y[0] + (int64_t)g * M * C4;
        const float4* y41 = a.nsets > 1 ? (const float4*)a.y[1] + (int64_t)g * M * C4 : y40;
        const int64_t step = (int64_t)PT * C4;
        const int64_t eend = pend * C4;
        int64_t e = (pbeg + pl) * C4 + c4;
        float4 msc = make_float4(0.f, 0.f, 0.f, 0.f), msh = msc;   // mask_from_y: scale / shift of this thread's channel quad
        if (a.mask_from_y) {
            const float4 gm = *(const float4*)(a.gamma[0] + c4 * 4), bt = *(const float4*)(a.beta[0] + c4 * 4);
            bn_scale_shift(gm.x, bt.x, mean[0].x, istd[0].x, msc.x, msh.x); bn_scale_shift(gm.y, bt.y, mean[0].y, istd[0].y, msc.y, msh.y);
            bn_scale_shift(gm.z, bt.z, mean[0].z, istd[0].z, msc.z, msh.z); bn_scale_shift(gm.w, bt.w, mean[0].w, istd[0].w, msc.w, msh.w);
        }
        auto consume = [&](float4 d, float4 zz, const float4& ya, const float4& yb) __attribute__((always_inline)) {
            if (a.mask_from_y)
                zz = make_float4(__fmaf_rn(ya.x, msc.x, msh.x), __fmaf_rn(ya.y, msc.y, msh.y), __fmaf_rn(ya.z, msc.z, msh.z), __fmaf_rn(ya.w, msc.w, msh.w));
            if (z4 || a.mask_from_y) {
                d.x = zz.x > 0.f ? d.x : 0.f; d.y = zz.y > 0.f ? d.y : 0.f;
                d.z = zz.z > 0.f ? d.z : 0.f; d.w = zz.w > 0.f ? d.w : 0.f;
            }
#pragma unroll
            for (int k = 0; k < 2; ++k)
                if (k < a.nsets) {
                    const float4& y = k ? yb : ya;
                    sd[k].x += d.x; sd[k].y += d.y; sd[k].z += d.z; sd[k].w += d.w;
                    sx[k].x = fmaf(d.x, (y.x - mean[k].x) * istd[k].x, sx[k].x);
                    sx[k].y = fmaf(d.y, (y.y - mean[k].y) * istd[k].y, sx[k].y);
                    sx[k].z = fmaf(d.z, (y.z - mean[k].z) * istd[k].z, sx[k].z);
                    sx[k].w = fmaf(d.w, (y.w - mean[k].w) * istd[k].w, sx[k].w);
                }
        };
        const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
        for (; e + (U - 1) * step < eend; e += U * step) {
            float4 d[U], zz[U], ya[U], yb[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                d[u] = dz4[e + u * step];
                zz[u] = z4 ? z4[e + u * step] : zero4;
                ya[u] = y40[e + u * step];
                yb[u] = a.nsets > 1 ? y41[e + u * step] : zero4;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) consume(d[u], zz[u], ya[u], yb[u]);
        }
        for (; e < eend; e += step)
            consume(dz4[e], z4 ? z4[e] : zero4, y40[e], a.nsets > 1 ? y41[e] : zero4);
    }
    // LDS layout: [set][2][PT][C]
    float* base = sm;
    if (pl < PT) {
#pragma unroll
        for (int k = 0; k < 2; ++k)
            if (k < a.nsets) {
                *(float4*)(base + ((size_t)(k * 2 + 0) * PT + pl) * a.C + c4 * 4) = sd[k];
                *(float4*)(base + ((size_t)(k * 2 + 1) * PT + pl) * a.C + c4 * 4) = sx[k];
            }
    }
    __syncthreads();
    for (int j = tid; j < a.nsets * 2 * a.C; j += 256) {
        const int c = j % a.C, kk = j / a.C;  // kk = set*2 + which
        double t = 0.0;
        for (int r = 0; r < PT; ++r) t += (double)base[((size_t)kk * PT + r) * a.C + c];
        const int k = kk >> 1, which = kk & 1;
        fx_add(&a.sums[(((int64_t)k * a.G + g) * 2 + which) * a.C + c], t);
    }
}

__global__ void __launch_bounds__(256) bn_bwd_apply_kernel(const BnBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    // per set: k1[C] (mean dpre), k2[C] (mean dpre*xhat), scale[C], mean[C], invstd[C]
    const int g = blockIdx.y, tid = threadIdx.x;
    const double Md = (double)a.m_per_group;
    for (int j = tid; j < a.nsets * a.C; j += 256) {
        const int c = j % a.C, k = j / a.C;
        const StatCell cdy = a.sums[(((int64_t)k * a.G + g) * 2 + 0) * a.C + c], cdx = a.sums[(((int64_t)k * a.G + g) * 2 + 1) * a.C + c];
        const double sdy = fx_decode(cdy.hi, cdy.lo), sdx = fx_decode(cdx.hi, cdx.lo);
        float* s = sm + (size_t)k * 6 * a.C;
        const float istd = a.invstd[k][(int64_t)g * a.C + c];
        s[c] = a.frozen ? 0.f : (float)(sdy / Md);
        s[a.C + c] = a.frozen ? 0.f : (float)(sdx / Md);
        s[2 * a.C + c] = a.gamma[k][c] * istd;
        s[3 * a.C + c] = a.mean[k][(int64_t)g * a.C + c];
        s[4 * a.C + c] = istd;
        if (a.mask_from_y) {
            float sc_, sh_;
            bn_scale_shift(a.gamma[k][c], a.beta[k][c], a.mean[k][(int64_t)g * a.C + c], istd, sc_, sh_);
            s[5 * a.C + c] = sh_;   // (scale: s[2C + c] = gamma * invstd, the same product)
        }
        if (blockIdx.x == 0 && g == 0) {
            double dg = 0.0, db = 0.0;
            for (int gg = 0; gg < a.G; ++gg) {
                const StatCell cb = a.sums[(((int64_t)k * a.G + gg) * 2 + 0) * a.C + c], cg = a.sums[(((int64_t)k * a.G + gg) * 2 + 1) * a.C + c];
                db += fx_decode(cb.hi, cb.lo);
                dg += fx_decode(cg.hi, cg.lo);
            }
            if (a.accumulate) {
                a.dgamma[k][c] += (float)dg;
                a.dbeta[k][c] += (float)db;
            } else {
                a.dgamma[k][c] = (float)dg;
                a.dbeta[k][c] = (float)db;
            }
        }
    }
    __syncthreads();
    const int C4 = a.C >> 2;
    const int64_t units = a.m_per_group * C4;
    for (int64_t u = (int64_t)blockIdx.x * 256 + tid; u < units; u += (int64_t)gridDim.x * 256) {
        const int c = (int)(u % C4) * 4;
        const int64_t e = (int64_t)g * units + u;
        float4 d = ((const float4*)a.dz)[e];
        if (a.z) {
            const float4 zz = ((const float4*)a.z)[e];
            d.x = zz.x > 0.f ? d.x : 0.f; d.y = zz.y > 0.f ? d.y : 0.f;
            d.z = zz.z > 0.f ? d.z : 0.f; d.w = zz.w > 0.f ? d.w : 0.f;
        } else if (a.mask_from_y) {
            const float4 y = ((const float4*)a.y[0])[e];
            d.x = __fmaf_rn(y.x, sm[2 * a.C + c], sm[5 * a.C + c]) > 0.f ? d.x : 0.f;
            d.y = __fmaf_rn(y.y, sm[2 * a.C + c + 1], sm[5 * a.C + c + 1]) > 0.f ? d.y : 0.f;
            d.z = __fmaf_rn(y.z, sm[2 * a.C + c + 2], sm[5 * a.C + c + 2]) > 0.f ? d.z : 0.f;
            d.w = __fmaf_rn(y.w, sm[2 * a.C + c + 3], sm[5 * a.C + c + 3]) > 0.f ? d.w : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 2; ++k)
            if (k < a.nsets) {
                const float* s = sm + (size_t)k * 6 * a.C;
                const float4 y = ((const float4*)a.y[k])[e];
                float4 o;
                o.x = s[2 * a.C + c] * (d.x - s[c] - (y.x - s[3 * a.C + c]) * s[4 * a.C + c] * s[a.C + c]);
                o.y = s[2 * a.C + c + 1] * (d.y - s[c + 1] - (y.y - s[3 * a.C + c + 1]) * s[4 * a.C + c + 1] * s[a.C + c + 1]);
                o.z = s[2 * a.C + c + 2] * (d.z - s[c + 2] - (y.z - s[3 * a.C + c + 2]) * s[4 * a.C + c + 2] * s[a.C + c + 2]);
                o.w = s[2 * a.C + c + 3] * (d.w - s[c + 3] - (y.w - s[3 * a.C + c + 3]) * s[4 * a.C + c + 3] * s[a.C + c + 3]);
                ((float4*)a.dy[k])[e] = o;
            }
    }
}

// One-pass BatchNorm backward (one BatchNorm, <= 2 groups): every thread keeps its share of the masked gradient and of xhat in
// registers (<= E float4 each), the workgroups reduce, meet at a grid-wide arrival counter, and apply from registers: dz, z, y are
// read once and dy written once (4 tensor passes instead of the 7 of reduce + apply).  All workgroups must be resident at once:
// the grid is one 512-thread workgroup per CU (54-160 VGPRs, 17 KB LDS); workgroups that find their CU full of weight-gradient
// workgroups of the second stream start when one of those retires; the wait is bounded so that a scheduling surprise shows up as a parity failure, not as a hung GPU.
constexpr int kBnFusedThreads = 512;
// accumulator replicas (same-address returning atomics serialise at the coherence point: 256 workgroups on one address cost ~20 us)
constexpr int kBnFusedReps = 8;
#ifndef OCL_BN_FLAT
#define OCL_BN_FLAT 40
#endif
constexpr int kBnFusedFlat = OCL_BN_FLAT;   // grids up to this size arrive at one counter
// NS = 2: the two BatchNorms of a projection block (main path + shortcut) share the masked gradient dz; their outputs differ only
// in xhat.  One launch reads dz, z, y_a, y_b and writes dy_a, dy_b (6 tensor passes, one grid arrival) instead of reduce + apply
// (10 passes, 2 launches).  The sum of the masked gradient is the same for both; each BatchNorm's arena receives it with its own
// sum of d * xhat.
template <int E, int NS = 1>
__global__ void __launch_bounds__(kBnFusedThreads) bn_bwd_fused_kernel(const BnBwdArgs a) {
    __shared__ float4 red[1 + NS][kBnFusedThreads];
    __shared__ float kk[1 + NS][4 * 40];
    __shared__ bool timed_out;   // some workgroup never arrived (not all resident at once): the results are poisoned with NaN
    const int C4 = a.C >> 2;
    const int tid = threadIdx.x;
    if (tid == 0) timed_out = false;
    const int wpg = gridDim.x / a.G;                   // workgroups per group
    const int g = blockIdx.x / wpg;
    const int S = (wpg * kBnFusedThreads / C4) * C4;   // unit stride of a thread: a multiple of C4, so its channel quad is fixed
    const int gt = (blockIdx.x - g * wpg) * kBnFusedThreads + tid;
    const int c4 = gt % C4;
    const int64_t M = a.m_per_group;
    const int64_t units = M * C4;
    const float4* dz4 = (const float4*)a.dz + (int64_t)g * units;
    const float4* z4 = a.z ? (const float4*)a.z + (int64_t)g * units : nullptr;
    const bool live = g < a.G && gt < S;
    float4 d[E], xh[NS][E];
    float4 sd = make_float4(0.f, 0.f, 0.f, 0.f), sx[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) sx[k] = sd;
    if (live) {
        float4 zz[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int64_t u = (int64_t)gt + (int64_t)e * S;
            const bool in = u < units;
            const int64_t uu = in ? u : 0;
            d[e] = dz4[uu];
#pragma unroll
            for (int k = 0; k < NS; ++k) xh[k][e] = ((const float4*)a.y[k] + (int64_t)g * units)[uu];
            zz[e] = z4 ? z4[uu] : make_float4(1.f, 1.f, 1.f, 1.f);
            if (!in) d[e] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (a.mask_from_y) {   // the activation was never written: its sign from the raw output, with the staging kernels' arithmetic
            const float4 gm = *(const float4*)(a.gamma[0] + c4 * 4), bt = *(const float4*)(a.beta[0] + c4 * 4);
            const float4 mn = *(const float4*)(a.mean[0] + (int64_t)g * a.C + c4 * 4), is = *(const float4*)(a.invstd[0] + (int64_t)g * a.C + c4 * 4);
            float4 sc, sh;
            bn_scale_shift(gm.x, bt.x, mn.x, is.x, sc.x, sh.x); bn_scale_shift(gm.y, bt.y, mn.y, is.y, sc.y, sh.y);
            bn_scale_shift(gm.z, bt.z, mn.z, is.z, sc.z, sh.z); bn_scale_shift(gm.w, bt.w, mn.w, is.w, sc.w, sh.w);
#pragma unroll
            for (int e = 0; e < E; ++e)
                zz[e] = make_float4(__fmaf_rn(xh[0][e].x, sc.x, sh.x), __fmaf_rn(xh[0][e].y, sc.y, sh.y), __fmaf_rn(xh[0][e].z, sc.z, sh.z),
                                    __fmaf_rn(xh[0][e].w, sc.w, sh.w));
        }
#pragma unroll
        for (int e = 0; e < E; ++e) {
            d[e].x = zz[e].x > 0.f ? d[e].x : 0.f; d[e].y = zz[e].y > 0.f ? d[e].y : 0.f;
            d[e].z = zz[e].z > 0.f ? d[e].z : 0.f; d[e].w = zz[e].w > 0.f ? d[e].w : 0.f;
            sd.x += d[e].x; sd.y += d[e].y; sd.z += d[e].z; sd.w += d[e].w;
        }
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const float4 mean = *(const float4*)(a.mean[k] + (int64_t)g * a.C + c4 * 4);
            const float4 istd = *(const float4*)(a.invstd[k] + (int64_t)g * a.C + c4 * 4);
#pragma unroll
            for (int e = 0; e < E; ++e) {
                xh[k][e].x = (xh[k][e].x - mean.x) * istd.x; xh[k][e].y = (xh[k][e].y - mean.y) * istd.y;
                xh[k][e].z = (xh[k][e].z - mean.z) * istd.z; xh[k][e].w = (xh[k][e].w - mean.w) * istd.w;
                sx[k].x = fmaf(d[e].x, xh[k][e].x, sx[k].x); sx[k].y = fmaf(d[e].y, xh[k][e].y, sx[k].y);
                sx[k].z = fmaf(d[e].z, xh[k][e].z, sx[k].z); sx[k].w = fmaf(d[e].w, xh[k][e].w, sx[k].w);
            }
        }
    }
    red[0][tid] = sd;
#pragma unroll
    for (int k = 0; k < NS; ++k) red[1 + k][tid] = sx[k];
    __syncthreads();
    // threads of this workgroup with channel quad q: tid = first(q) + k*C4
    if (g < a.G && tid < (1 + NS) * C4) {
        const int which = tid / C4, q = tid - which * C4;   // 0: sum d; 1 + k: sum d * xhat of BatchNorm k
        const int base = (blockIdx.x - g * wpg) * kBnFusedThreads;
        int first = (q - base % C4 + C4) % C4;
        double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
        for (int t = first; t < kBnFusedThreads; t += C4) {
            const float4 v = red[which][t];
            t0 += (double)v.x; t1 += (double)v.y; t2 += (double)v.z; t3 += (double)v.w;
        }
        unsigned long long r = 0ull;
        // returning atomics: the wave waits for them to have executed (at the device-wide coherence point) before the barrier below
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            if (which != 0 && which != 1 + k) continue;   // the sum of d goes to both arenas, the sum of d * xhat_k to its own
            StatCell* arena = k == 0 ? a.fsums : a.fsums_b;
            StatCell* dst = arena + ((int64_t)(blockIdx.x % kBnFusedReps) * a.G * 2 + (int64_t)g * 2 + (which ? 1 : 0)) * a.C + q * 4;
            r += fx_fetch_add(dst + 0, t0) + fx_fetch_add(dst + 1, t1) + fx_fetch_add(dst + 2, t2) + fx_fetch_add(dst + 3, t3);
        }
        if (r == 0x123456789abcdef1ull) red[0][0].x = 0.f;   // keeps the returns (practically never true)
    }
    // ---- grid-wide arrival ---------------------------------------------------------------------------
    // Relaxed device-scope atomics only: they execute at the coherence point and bypass the per-XCD L2, so no release / acquire
    // fence (an L2 write-back + invalidate per fence on this part: ~50 us per launch when the spin loop carried an acquire).
    __syncthreads();
    if (tid == 0) {
        // two-level arrival: 8 sub-counters (same-address atomics serialise: 256 arrivals on one counter cost ~15 us), the last
        // arrival of each sub-counter reports to the master counter a.barrier[0]
        const unsigned sub = blockIdx.x % kBnFusedReps;
        const unsigned members = (gridDim.x - sub + kBnFusedReps - 1) / kBnFusedReps;
        // (replay-sized passes run 8 - 33 workgroups: they arrive at the master counter directly -- one device-scope round trip less in a
        // kernel that is nothing but such round trips there, profiles/r6_bn_flat_arrival_ab.txt)
        const bool flat = gridDim.x <= (unsigned)kBnFusedFlat;
        const unsigned groups_total = flat ? gridDim.x : (unsigned)kBnFusedReps;
        if (flat || __hip_atomic_fetch_add(a.barrier + 1 + sub, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == members - 1)
            __hip_atomic_fetch_add(a.barrier, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(a.barrier, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < groups_total && ++spins < (1 << 22))
            __builtin_amdgcn_s_sleep(1);
        timed_out = spins >= (1 << 22);
        if (timed_out && a.err) __hip_atomic_fetch_or(a.err, (unsigned)ASYNC_ERR_BN_BARRIER, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __syncthreads();
    const double Md = (double)M;
    if (g < a.G && tid < (1 + NS) * a.C) {
        const int which = tid / a.C, c = tid - which * a.C;
        const StatCell* arena = which <= 1 ? a.fsums : a.fsums_b;
        const double v = fx_total_atomic(arena, (int64_t)a.G * 2 * a.C, ((int64_t)g * 2 + (which ? 1 : 0)) * a.C + c);
        kk[which][c] = timed_out ? __builtin_nanf("") : (float)(v / Md);
    }
    if (blockIdx.x == 0 && tid < NS * a.C) {   // dgamma / dbeta over all groups
        const int k = tid / a.C, c = tid - k * a.C;
        const StatCell* arena = k == 0 ? a.fsums : a.fsums_b;
        double db = 0.0, dg = 0.0;
        for (int gg = 0; gg < a.G; ++gg) {   // (each group's total is exact; the groups are added in order)
            db += fx_total_atomic(arena, (int64_t)a.G * 2 * a.C, ((int64_t)gg * 2 + 0) * a.C + c);
            dg += fx_total_atomic(arena, (int64_t)a.G * 2 * a.C, ((int64_t)gg * 2 + 1) * a.C + c);
        }
        if (a.accumulate) {
            a.dgamma[k][c] += (float)dg;
            a.dbeta[k][c] += (float)db;
        } else {
            a.dgamma[k][c] = (float)dg;
            a.dbeta[k][c] = (float)db;
        }
    }
    __syncthreads();
    if (live) {
        const float4 k1 = *(const float4*)&kk[0][c4 * 4];
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const float4 k2 = *(const float4*)&kk[1 + k][c4 * 4];
            const float4 gm = *(const float4*)(a.gamma[k] + c4 * 4);
            const float4 istd = *(const float4*)(a.invstd[k] + (int64_t)g * a.C + c4 * 4);
            const float4 sc = make_float4(gm.x * istd.x, gm.y * istd.y, gm.z * istd.z, gm.w * istd.w);
            float4* o4 = (float4*)a.dy[k] + (int64_t)g * units;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int64_t u = (int64_t)gt + (int64_t)e * S;
                if (u < units) {
                    float4 o;
                    o.x = sc.x * (d[e].x - k1.x - xh[k][e].x * k2.x);
                    o.y = sc.y * (d[e].y - k1.y - xh[k][e].y * k2.y);
                    o.z = sc.z * (d[e].z - k1.z - xh[k][e].z * k2.z);
                    o.w = sc.w * (d[e].w - k1.w - xh[k][e].w * k2.w);
                    o4[u] = o;
                }
            }
        }
    }
}

// BatchNorm backward of a SMALL map (layer 4 of a replay-sized pass), partitioned by CHANNEL: one workgroup
// owns one channel quad for every pixel of every group, so its batch sums need nobody else -- no atomics, no grid-wide arrival, no replicas
// to read back.  bn_bwd_fused_kernel on such a map is five dependent device-scope round trips (10.8 - 12.5 us for a few hundred KB); here:
// one strided read of dz, z, y (16 bytes per lane at a stride of C floats), a wave butterfly + the wave totals in fp64 in a fixed order
// (deterministic), the apply from registers: 6.7 - 8.4 us at one unit per thread (profiles/r6_bn_chan_ab.txt; at four units per thread --
// layer 3 of a 20-image pass on 20 workgroups -- it LOSES to the one-pass kernel, hence the size gate in launch_bn_bwd).  Two groups split
// the workgroup's threads (the two-group replay pass of ER: 10 + 10 images).  NS as bn_bwd_fused_kernel.
constexpr int kBnChanThreads = 512;
template <int E, int NS>
__global__ void __launch_bounds__(kBnChanThreads) bn_bwd_chan_kernel(const BnBwdArgs a) {
    constexpr int NW = kBnChanThreads / 64;
    __shared__ double wred[1 + NS][4][NW];
    __shared__ double tot[2][1 + NS][4];
    __shared__ float kk[2][1 + NS][4];
    const int c4 = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int C4 = a.C >> 2;
    const int T = kBnChanThreads / a.G;           // threads per group (G = 1 or 2)
    const int g = tid / T, pt = tid - g * T;
    const int64_t M = a.m_per_group;
    const float4 f4z = make_float4(0.f, 0.f, 0.f, 0.f);
    const int64_t base = (int64_t)g * M * C4 + c4;
    float4 d[E], xh[NS][E], zz[E];
    float4 sd = f4z, sx[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) sx[k] = f4z;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int64_t pix = pt + (int64_t)e * T;
        const bool in = pix < M;
        const int64_t u = base + (in ? pix : 0) * C4;
        d[e] = ((const float4*)a.dz)[u];
#pragma unroll
        for (int k = 0; k < NS; ++k) xh[k][e] = ((const float4*)a.y[k])[u];
        zz[e] = a.z ? ((const float4*)a.z)[u] : make_float4(1.f, 1.f, 1.f, 1.f);
        if (!in) d[e] = f4z;
    }
    if (a.mask_from_y) {   // (the arithmetic of the staging kernels, as in bn_bwd_fused_kernel)
        const float4 gm = *(const float4*)(a.gamma[0] + c4 * 4), bt = *(const float4*)(a.beta[0] + c4 * 4);
        const float4 mn = *(const float4*)(a.mean[0] + (int64_t)g * a.C + c4 * 4), is = *(const float4*)(a.invstd[0] + (int64_t)g * a.C + c4 * 4);
        float4 sc, sh;
        bn_scale_shift(gm.x, bt.x, mn.x, is.x, sc.x, sh.x); bn_scale_shift(gm.y, bt.y, mn.y, is.y, sc.y, sh.y);
        bn_scale_shift(gm.z, bt.z, mn.z, is.z, sc.z, sh.z); bn_scale_shift(gm.w, bt.w, mn.w, is.w, sc.w, sh.w);
#pragma unroll
        for (int e = 0; e < E; ++e)
            zz[e] = make_float4(__fmaf_rn(xh[0][e].x, sc.x, sh.x), __fmaf_rn(xh[0][e].y, sc.y, sh.y), __fmaf_rn(xh[0][e].z, sc.z, sh.z),
                                __fmaf_rn(xh[0][e].w, sc.w, sh.w));
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
        d[e].x = zz[e].x > 0.f ? d[e].x : 0.f; d[e].y = zz[e].y > 0.f ? d[e].y : 0.f;
        d[e].z = zz[e].z > 0.f ? d[e].z : 0.f; d[e].w = zz[e].w > 0.f ? d[e].w : 0.f;
        sd.x += d[e].x; sd.y += d[e].y; sd.z += d[e].z; sd.w += d[e].w;
    }
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const float4 mean = *(const float4*)(a.mean[k] + (int64_t)g * a.C + c4 * 4);
        const float4 istd = *(const float4*)(a.invstd[k] + (int64_t)g * a.C + c4 * 4);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            xh[k][e].x = (xh[k][e].x - mean.x) * istd.x; xh[k][e].y = (xh[k][e].y - mean.y) * istd.y;
            xh[k][e].z = (xh[k][e].z - mean.z) * istd.z; xh[k][e].w = (xh[k][e].w - mean.w) * istd.w;
            sx[k].x = fmaf(d[e].x, xh[k][e].x, sx[k].x); sx[k].y = fmaf(d[e].y, xh[k][e].y, sx[k].y);
            sx[k].z = fmaf(d[e].z, xh[k][e].z, sx[k].z); sx[k].w = fmaf(d[e].w, xh[k][e].w, sx[k].w);
        }
    }
    // wave butterfly in fp64 (a lane's partial covers <= E values; a wave lies inside one group), wave totals to LDS
    auto wsum = [&](float v) __attribute__((always_inline)) -> double {
        double t = (double)v;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
        return t;
    };
    {
        const double t0 = wsum(sd.x), t1 = wsum(sd.y), t2 = wsum(sd.z), t3 = wsum(sd.w);
        if (lane == 0) { wred[0][0][wave] = t0; wred[0][1][wave] = t1; wred[0][2][wave] = t2; wred[0][3][wave] = t3; }
    }
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const double t0 = wsum(sx[k].x), t1 = wsum(sx[k].y), t2 = wsum(sx[k].z), t3 = wsum(sx[k].w);
        if (lane == 0) { wred[1 + k][0][wave] = t0; wred[1 + k][1][wave] = t1; wred[1 + k][2][wave] = t2; wred[1 + k][3][wave] = t3; }
    }
    __syncthreads();
    if (tid < (1 + NS) * 4 * a.G) {
        const int gg = tid / ((1 + NS) * 4), r = tid - gg * (1 + NS) * 4;
        const int which = r >> 2, c = r & 3;
        const int wpg = NW / a.G;
        double t = 0.0;
        for (int w = gg * wpg; w < (gg + 1) * wpg; ++w) t += wred[which][c][w];
        tot[gg][which][c] = t;
        kk[gg][which][c] = (float)(t / (double)M);
    }
    __syncthreads();
    if (tid < (1 + NS) * 4) {   // dbeta = sum(d), dgamma_k = sum(d * xhat_k), over the groups in order
        const int which = tid >> 2, c = tid & 3, ch = c4 * 4 + c;
        double t = tot[0][which][c];
        if (a.G == 2) t += tot[1][which][c];
        if (which == 0) {
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                if (a.accumulate) a.dbeta[k][ch] += (float)t;
                else a.dbeta[k][ch] = (float)t;
            }
        } else {
            if (a.accumulate) a.dgamma[which - 1][ch] += (float)t;
            else a.dgamma[which - 1][ch] = (float)t;
        }
    }
    const float4 k1 = *(const float4*)&kk[g][0][0];
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const float4 k2 = *(const float4*)&kk[g][1 + k][0];
        const float4 gm = *(const float4*)(a.gamma[k] + c4 * 4);
        const float4 istd = *(const float4*)(a.invstd[k] + (int64_t)g * a.C + c4 * 4);
        const float4 sc = make_float4(gm.x * istd.x, gm.y * istd.y, gm.z * istd.z, gm.w * istd.w);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int64_t pix = pt + (int64_t)e * T;
            if (pix < M) {
                float4 o;
                o.x = sc.x * (d[e].x - k1.x - xh[k][e].x * k2.x);
                o.y = sc.y * (d[e].y - k1.y - xh[k][e].y * k2.y);
                o.z = sc.z * (d[e].z - k1.z - xh[k][e].z * k2.z);
                o.w = sc.w * (d[e].w - k1.w - xh[k][e].w * k2.w);
                ((float4*)a.dy[k])[base + pix * C4] = o;
            }
        }
    }
}

static int g_bn_bwd_cap = 0, g_bn_bwd_unroll = 0, g_bn_bwd_phase = 0;   // micro-benchmark overrides (kbench)
void bn_bwd_tune(int cap, int unroll, int phase) { g_bn_bwd_cap = cap; g_bn_bwd_unroll = unroll; g_bn_bwd_phase = phase; }

static int g_bn_fused = -1;   // -1: environment (OCL_BN_FUSED, default on; 0 = the reduce + apply pair that passes of > 2 groups use anyway: the way out when a
                              // shared GPU cannot hold the one-pass kernel's grid-wide arrival, see check_async_error; tests/test_gpu_ring.py)
static int g_num_cus = 0;
void bn_bwd_fused_enable(int on) { g_bn_fused = on; }

int launch_bn_bwd(const BnBwdArgs& a, hipStream_t s) {
    OCL_REQUIRE(a.nsets == 1 || a.nsets == 2, "bn_bwd: nsets=%d", a.nsets);
    const int C4 = a.C / 4, PT = 256 / C4;
    if (g_bn_fused < 0) {
        const char* e = getenv("OCL_BN_FUSED");
        g_bn_fused = e ? atoi(e) : 1;
    }
    // small maps: one workgroup per channel quad, no cross-workgroup reduction (bn_bwd_chan_kernel; OCL_BN_CHAN=0: off).  Gate: ONE unit per
    // thread (all groups' pixels <= 512: layer 4 up to 32 images) and >= 16 channel quads; at two units per thread it is neutral (6 x 84x84) or
    // loses (64 images in two groups: +7.5 us per pass), at four (layer 3 of a 20-image pass) it loses -- profiles/r6_bn_chan_ab.txt
    static const bool bn_chan = [] { const char* e = getenv("OCL_BN_CHAN"); return !(e && e[0] == '0'); }();
    if (bn_chan && g_bn_fused && !a.frozen && g_bn_bwd_phase == 0 && a.G <= 2 && C4 >= 16 && a.m_per_group * a.G <= kBnChanThreads) {
        ProfScope ps(PROF_BN, s);
        if (a.nsets == 2) hipLaunchKernelGGL((bn_bwd_chan_kernel<1, 2>), dim3(C4), dim3(kBnChanThreads), 0, s, a);
        else hipLaunchKernelGGL((bn_bwd_chan_kernel<1, 1>), dim3(C4), dim3(kBnChanThreads), 0, s, a);
        OCL_LAUNCH_CHECK();
        return OCL_OK;
    }
    if (g_bn_fused && a.barrier && a.fsums && (a.nsets == 1 || a.fsums_b) && a.G <= 2 && a.C <= 160 && g_bn_bwd_phase == 0 && !a.frozen) {
        if (!g_num_cus) {
            int dev = 0;
            hipDeviceProp_t prop;
            OCL_HIP(hipGetDevice(&dev));
            OCL_HIP(hipGetDeviceProperties(&prop, dev));
            // residency: the grid never exceeds one workgroup per CU, and every instantiation must be admissible at that rate
            // (checked once against the occupancy query); a time-out at run time is reported through the asynchronous error word
            int b3 = 0, b6 = 0, b12 = 0, c3 = 0, c6 = 0;
            OCL_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&b3, bn_bwd_fused_kernel<3>, kBnFusedThreads, 0));
            OCL_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&b6, bn_bwd_fused_kernel<6>, kBnFusedThreads, 0));
            OCL_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&b12, bn_bwd_fused_kernel<12>, kBnFusedThreads, 0));
            OCL_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&c3, (bn_bwd_fused_kernel<3, 2>), kBnFusedThreads, 0));
            OCL_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&c6, (bn_bwd_fused_kernel<6, 2>), kBnFusedThreads, 0));
            if (std::min(std::min(b3, std::min(b6, b12)), std::min(c3, c6)) < 1) g_bn_fused = 0;   // cannot be co-resident: two-kernel path
            g_num_cus = std::max(2, prop.multiProcessorCount);
        }
        // about 6 float4 per thread and tensor; never more workgroups than CUs (all must be resident), fewer for the small maps
        // (the arrival costs grow with the workgroup count, the small maps are latency-bound anyway)
        const int64_t total_units = a.m_per_group * C4 * a.G;
        const int per_thread = a.nsets == 2 ? 5 : 6;   // (two sets keep one more register array per unit: at most 6 units per thread)
        int grid = (int)std::min<int64_t>(g_num_cus, std::max<int64_t>(8, (total_units + kBnFusedThreads * per_thread - 1) / (kBnFusedThreads * per_thread)));
        grid = std::max(a.G, grid / a.G * a.G);
        const int wpg = grid / a.G;
        const int64_t S = (int64_t)(wpg * kBnFusedThreads / C4) * C4;
        const int64_t need = (a.m_per_group * C4 + S - 1) / S;
        if (a.nsets == 2 && need <= 6 && g_bn_fused) {   // two BatchNorms sharing dz (projection blocks)
            ProfScope ps(PROF_BN, s);
            BnBwdArgs af = a;
            af.err = async_error_word_device();
            if (need <= 3) hipLaunchKernelGGL((bn_bwd_fused_kernel<3, 2>), dim3(grid), dim3(kBnFusedThreads), 0, s, af);
            else hipLaunchKernelGGL((bn_bwd_fused_kernel<6, 2>), dim3(grid), dim3(kBnFusedThreads), 0, s, af);
            OCL_LAUNCH_CHECK();
            return OCL_OK;
        }
        if (a.nsets == 1 && need <= 12 && g_bn_fused) {
            ProfScope ps(PROF_BN, s);
            BnBwdArgs af = a;
            af.err = async_error_word_device();
            if (need <= 3) hipLaunchKernelGGL(bn_bwd_fused_kernel<3>, dim3(grid), dim3(kBnFusedThreads), 0, s, af);
            else if (need <= 6) hipLaunchKernelGGL(bn_bwd_fused_kernel<6>, dim3(grid), dim3(kBnFusedThreads), 0, s, af);
            else hipLaunchKernelGGL(bn_bwd_fused_kernel<12>, dim3(grid), dim3(kBnFusedThreads), 0, s, af);
            OCL_LAUNCH_CHECK();
            return OCL_OK;
        }
    }
    const int U = g_bn_bwd_unroll ? g_bn_bwd_unroll : 4;
    // passes per block: 8 for the large maps, 4 once a group has fewer than 1024 passes in total (more, shorter blocks: the
    // small layers are latency-bound) -- profiles/r1_kbench_bn_sweep.txt
    const int passes = g_bn_bwd_cap > 1024 ? (g_bn_bwd_cap > 2048 ? 2 : 4) : (g_bn_bwd_cap == 0 && a.m_per_group / PT < 1024 ? 4 : 8);
    const int cap = g_bn_bwd_cap ? g_bn_bwd_cap : 1024;
    const int64_t per_block_pixels = (int64_t)PT * passes;
    const int bx = (int)std::max<int64_t>(1, std::min<int64_t>(std::max(1, cap / a.G), (a.m_per_group + per_block_pixels - 1) / per_block_pixels));
    ProfScope ps(PROF_BN, s);
    const size_t sm1 = (size_t)a.nsets * 2 * PT * a.C * 4;
    if (g_bn_bwd_phase != 2) {
        if (U == 1) hipLaunchKernelGGL(bn_bwd_reduce_kernel<1>, dim3(bx, a.G), dim3(256), sm1, s, a);
        else if (U == 2) hipLaunchKernelGGL(bn_bwd_reduce_kernel<2>, dim3(bx, a.G), dim3(256), sm1, s, a);
        else hipLaunchKernelGGL(bn_bwd_reduce_kernel<4>, dim3(bx, a.G), dim3(256), sm1, s, a);
        OCL_LAUNCH_CHECK();
    }
    if (g_bn_bwd_phase == 1) return OCL_OK;
    const int64_t units = a.m_per_group * C4;
    const int bx2 = (int)std::max<int64_t>(1, std::min<int64_t>(1024, (units + 1023) / 1024));
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(bx2, a.G), dim3(256), (size_t)a.nsets * 6 * a.C * 4, s, a);
    OCL_LAUNCH_CHECK();
    return OCL_OK;
}

// ---- apply half of a BatchNorm backward whose two batch sums came out of the producing data gradient's epilogue (EPI_BNB) ------------
// d is the ReLU-masked gradient; per (group, channel): k1 = sum(d) / M, k2 = invstd * sum(d * (y - mean)) / M (= mean of d * xhat);
// dy = gamma * invstd * (d - k1 - xhat * k2), the statement of bn_bwd_apply_kernel.  The replicas are summed in a fixed order.
__global__ void __launch_bounds__(256) bn_bwd_apply_e_kernel(const BnApplyEArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];   // k1[C], k2[C], scale[C], mean[C], invstd[C]
    const int g = blockIdx.y, tid = threadIdx.x;
    const double Md = (double)a.m_per_group;
    for (int c = tid; c < a.C; c += 256) {
        const float istd = a.invstd[(int64_t)g * a.C + c];
        double s1, s2;
        fx_total2(a.esums, a.esums_rep_stride, ((int64_t)g * 2 + 0) * a.C + c, ((int64_t)g * 2 + 1) * a.C + c, s1, s2);
        sm[c] = (float)(s1 / Md);
        sm[a.C + c] = (float)(s2 * (double)istd / Md);
        sm[2 * a.C + c] = a.gamma[c] * istd;
        sm[3 * a.C + c] = a.mean[(int64_t)g * a.C + c];
        sm[4 * a.C + c] = istd;
        if (blockIdx.x == 0 && g == 0) {   // dgamma = sum over the groups of sum(d * xhat), dbeta = sum(d)
            double dg = 0.0, db = 0.0;
            for (int gg = 0; gg < a.G; ++gg) {
                double t1, t2;
                fx_total2(a.esums, a.esums_rep_stride, ((int64_t)gg * 2 + 0) * a.C + c, ((int64_t)gg * 2 + 1) * a.C + c, t1, t2);
                db += t1;
                dg += t2 * (double)a.invstd[(int64_t)gg * a.C + c];
            }
            if (a.accumulate) {
                a.dgamma[c] += (float)dg;
                a.dbeta[c] += (float)db;
            } else {
                a.dgamma[c] = (float)dg;
                a.dbeta[c] = (float)db;
            }
        }
    }
    __syncthreads();
    const int C4 = a.C >> 2;
    const int64_t units = a.m_per_group * C4;
    const float4* d4 = (const float4*)a.d + (int64_t)g * units;
    const float4* y4 = (const float4*)a.y + (int64_t)g * units;
    float4* o4 = (float4*)a.dy + (int64_t)g * units;
    const int64_t stride = (int64_t)gridDim.x * 256;
    auto one = [&](int64_t u, const float4 d, const float4 y) __attribute__((always_inline)) {
        const int c = (int)(u % C4) * 4;
        float4 o;
        o.x = sm[2 * a.C + c] * (d.x - sm[c] - (y.x - sm[3 * a.C + c]) * sm[4 * a.C + c] * sm[a.C + c]);
        o.y = sm[2 * a.C + c + 1] * (d.y - sm[c + 1] - (y.y - sm[3 * a.C + c + 1]) * sm[4 * a.C + c + 1] * sm[a.C + c + 1]);
        o.z = sm[2 * a.C + c + 2] * (d.z - sm[c + 2] - (y.z - sm[3 * a.C + c + 2]) * sm[4 * a.C + c + 2] * sm[a.C + c + 2]);
        o.w = sm[2 * a.C + c + 3] * (d.w - sm[c + 3] - (y.w - sm[3 * a.C + c + 3]) * sm[4 * a.C + c + 3] * sm[a.C + c + 3]);
        o4[u] = o;
    };
    int64_t u = (int64_t)blockIdx.x * 256 + tid;
    for (; u + 3 * stride < units; u += 4 * stride) {   // four units in flight per thread
        const float4 d0 = d4[u], d1 = d4[u + stride], d2 = d4[u + 2 * stride], d3 = d4[u + 3 * stride];
        const float4 y0 = y4[u], y1 = y4[u + stride], y2 = y4[u + 2 * stride], y3 = y4[u + 3 * stride];
        one(u, d0, y0); one(u + stride, d1, y1); one(u + 2 * stride, d2, y2); one(u + 3 * stride, d3, y3);
    }
    for (; u < units; u += stride) one(u, d4[u], y4[u]);
}

int launch_bn_apply_e(const BnApplyEArgs& a, hipStream_t s) {
    OCL_REQUIRE(a.C % 4 == 0 && a.G >= 1 && a.m_per_group > 0, "bn_apply_e: C=%d G=%d", a.C, a.G);
    const int64_t units = a.m_per_group * (a.C / 4);
    const int bx = (int)std::max<int64_t>(1, std::min<int64_t>(1024 / a.G, (units + 1023) / 1024));
    ProfScope ps(PROF_BN, s);
    hipLaunchKernelGGL(bn_bwd_apply_e_kernel, dim3(bx, a.G), dim3(256), (size_t)5 * a.C * 4, s, a);
    OCL_LAUNCH_CHECK();
    return OCL_OK;
}

// =====================================================================================================
// avg_pool2d(4) + flatten (C,ph,pw order), l2-normalise, misc
// =====================================================================================================
__global__ void __launch_bounds__(256) avgpool_fwd_kernel(const float* __restrict__ z, float* __restrict__ feat, int H, int W, int C,
                                                          int PH, int PW) {
    const int n = blockIdx.x;
    const int D = C * PH * PW;
    for (int o = threadIdx.x; o < D; o += blockDim.x) {
        const int c = o / (PH * PW), r = o - c * PH * PW;
        const int ph = r / PW, pw = r - ph * PW;
        float s = 0.f;
        for (int dy = 0; dy < 4; ++dy)
            for (int dx = 0; dx < 4; ++dx) s += z[(((int64_t)n * H + ph * 4 + dy) * W + pw * 4 + dx) * C + c];
        feat[(int64_t)n * D + o] = s * (1.0f / 16.0f);
    }
}
__global__ void __launch_bounds__(256) avgpool_bwd_kernel(const float* __restrict__ dfeat, float* __restrict__ dz, int H, int W, int C,
                                                          int PH, int PW) {
    const int n = blockIdx.x;
    const int D = C * PH * PW;
    const int total = H * W * C;
    for (int e = threadIdx.x + blockIdx.y * blockDim.x; e < total; e += blockDim.x * gridDim.y) {
        const int c = e % C, p = e / C;
        const int y = p / W, x = p - y * W;
        float v = 0.f;
        if (y < PH * 4 && x < PW * 4) v = dfeat[(int64_t)n * D + c * PH * PW + (y >> 2) * PW + (x >> 2)] * (1.0f / 16.0f);
        dz[(int64_t)n * total + e] = v;
    }
}
int launch_avgpool_fwd(const float* z, float* feat, int N, int H, int W, int C, hipStream_t s) {
    ProfScope ps(PROF_HEAD, s);
    hipLaunchKernelGGL(avgpool_fwd_kernel, dim3(N), dim3(256), 0, s, z, feat, H, W, C, H / 4, W / 4);
    OCL_LAUNCH_CHECK();
    return OCL_OK;
}
int launch_avgpool_bwd(const float* dfeat, float* dz, int N, int H, int W, int C, hipStream_t s) {
    ProfScope ps(PROF_HEAD, s);
    hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(N, std::max(1, std::min(8, cdiv(H * W * C, 2048)))), dim3(256), 0, s, dfeat, dz, H,
                       W, C, H / 4, W / 4);
    OCL_LAUNCH_CHECK();
    return OCL_OK;
}

__global__ void __launch_bounds__(64) l2norm_fwd_kernel(const float* __restrict__ v, float* __restrict__ out, float* __restrict__ norms,
                                                        int d, float* __restrict__ out2) {
    const int n = blockIdx.x, lane = threadIdx.x;
    const float* p = v + (int64_t)n * d;
    float ss = 0.f;
    for (int j = lane; j < d; j += 64) ss = fmaf(p[j], p[j], ss);
    ss = wave_sum(ss);
    const float nrm = fmaxf(sqrtf(ss), 1e-12f);  // F.normalize eps
    if (lane == 0) norms[n] = nrm;
    for (int j = lane; j < d; j += 64) {
        const float q = p[j] / nrm;
        out[(int64_t)n * d + j] = q;
        if (out2) out2[(int64_t)n * d + j] = q;   // the caller's tensor (saves a device-to-device copy launch)
    }
}
__global__ void __launch_bounds__(64) l2norm_bwd_kernel(const float* __restrict__ out, const float* __restrict__ norms,
                                                        const float* __restrict__ dout, float* __restrict__ dv, int d) {
    const int n = blockIdx.x, lane = threadIdx.x;
    const float* o = out + (int64_t)n * d;
    const float* g = dout + (int64_t)n * d;
    float dot = 0.f;
    for (int j = lane; j < d; j += 64) dot = fmaf(o[j], g[j], dot);
    dot = wave_sum(dot);
    const float inv = 1.0f / norms[n];
    for (int j = lane; j < d; j += 64) dv[(int64_t)n * d + j] = (g[j] - o[j] * dot) * inv;
}
int launch_l2norm_fwd(const float* v, float* out, float* norms, int n, int d, hipStream_t s, float* out2) {
    ProfScope ps(PROF_HEAD, s);
    hipLaunchKernelGGL(l2norm_fwd_kernel, dim3(n), dim3(64), 0, s, v, out, norms, d, out2);
    OCL_LAUNCH_CHECK();
    return OCL_OK;
}
int launch_l2norm_bwd(const float* out, const float* norms, const float* dout, float* dv, int n, int d, hipStream_t s) {
    ProfScope ps(PROF_HEAD, s);
    hipLaunchKernelGGL(l2norm_bwd_kernel, dim3(n), dim3(64), 0, s, out, norms, dout, dv, d);
    OCL_LAUNCH_CHECK();
    return OCL_OK;
}

__global__ void __launch_bounds__(256) relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ a, float* __restrict__ dx,
                                                       int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        dx[i] = a[i] > 0.f ? dy[i] : 0.f;
}
int launch_relu_bwd(const float* dy, const float* a, float* dx, int64_t n, hipStream_t s) {
    ProfScope ps(PROF_HEAD, s);
    hipLaunchKernelGGL(relu_bwd_kernel, dim3((unsigned)std::min<int64_t>(1024, (n + 255) / 256)), dim3(256), 0, s, dy, a, dx, n);
    OCL_LAUNCH_CHECK();
    return OCL_OK;
}

// out[c] (+)= sum_r m[r][c]: 32 columns x 8 row lanes per workgroup, lane sums combined through LDS in a fixed order
__global__ void __launch_bounds__(256) colsum_kernel(const float* __restrict__ m, int rows, int cols, float* __restrict__ out,
                                                     int accumulate) {
    __shared__ float red[8][33];
    const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    float s = 0.f;
    if (c < cols)
        for (int r = rl; r < rows; r += 8) s += m[(int64_t)r * cols + c];
    red[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && c < cols) {
        const float v = ((red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl])) + ((red[4][cl] + red[5][cl]) + (red[6][cl] + red[7][cl]));
        out[c] = accumulate ? out[c] + v : v;
    }
}
int launch_colsum(const float* m, int rows, int cols, float* out, int accumulate, hipStream_t s) {
    ProfScope ps(PROF_HEAD, s);
    hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(cols, 32)), dim3(256), 0, s, m, rows, cols, out, accumulate);
    OCL_LAUNCH_CHECK();
    return OCL_OK;
}

__global__ void __launch_bounds__(256) fill_kernel(float* __restrict__ p, int64_t n, float v) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = v;
}
int launch_fill(float* p, int64_t n, float v, hipStream_t s) {
    if (n <= 0) return OCL_OK;
    ProfScope ps(PROF_HEAD, s);
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)std::min<int64_t>(1024, (n + 255) / 256)), dim3(256), 0, s, p, n, v);
    OCL_LAUNCH_CHECK();
    return OCL_OK;
}

// Allow every instantiation to use the full 160 KiB of dynamic LDS.
int conv_kernels_init() {
    static bool done_dev[kMaxDevices] = {false};   // function attributes and the mode symbol are per device
    int dev = 0;
    OCL_HIP(hipGetDevice(&dev));
    OCL_REQUIRE(dev >= 0 && dev < kMaxDevices, "conv_kernels_init: device %d", dev);
    bool& done = done_dev[dev];
    if (done) return OCL_OK;
    {
        const char* e = getenv("OCL_DETERMINISTIC");
        if (e && e[0] == '1') {
            int rc = set_deterministic_sums(1);
            if (rc != OCL_OK) return rc;
        }
    }
    if (int rc = wgrad_kernels_init()) return rc;
    if (int rc = convw_kernels_init()) return rc;
    for (int m = 1; m <= 5; ++m)
        for (int n = 1; n <= 2; ++n)
            for (int pf = 4; pf <= 8; pf += 4)
                for (int res = 0; res < 2; ++res)
                    OCL_HIP(hipFuncSetAttribute((const void*)convt_fn(m, n, pf, res), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit));
    for (int m = 1; m <= 5; ++m)
        for (int pf = 4; pf <= 8; pf += 4)
            for (int res = 0; res < 2; ++res)
                OCL_HIP(hipFuncSetAttribute((const void*)convt_fn(m, 1, pf, res, 1), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit));
    for (int m = 1; m <= 5; ++m)
        for (int pf = 4; pf <= 8; pf += 4)
            for (int cls = 0; cls < 2; ++cls)
                OCL_HIP(hipFuncSetAttribute((const void*)convt_fn(m, 1, pf, 0, cls, 1), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit));
    for (int nt = 1; nt <= 2; ++nt)
        for (int tr = 0; tr < 2; ++tr)
            OCL_HIP(hipFuncSetAttribute((const void*)convs_fn(nt, tr != 0), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit));
    // the EPI_BNB instantiations
    for (int m = 1; m <= 5; ++m)
        for (int pf = 4; pf <= 8; pf += 4) {
            for (int n = 1; n <= 2; ++n)
                for (int res = 0; res < 2; ++res)
                    OCL_HIP(hipFuncSetAttribute((const void*)convt_fn(m, n, pf, res, 0, 0, 1), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit));
            OCL_HIP(hipFuncSetAttribute((const void*)convt_fn(m, 1, pf, 0, 0, 1, 1), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit));
        }
    for (int nt = 1; nt <= 2; ++nt)
        for (int det = 0; det < 2; ++det)
            for (int bnb = det ? 0 : 1; bnb < 2; ++bnb)
                OCL_HIP(hipFuncSetAttribute((const void*)convs_fn(nt, false, bnb != 0, det != 0), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit));
    for (int st = 0; st < 3; ++st) {
        OCL_HIP(hipFuncSetAttribute((const void*)convq_fn(2, 4, st), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit));
        OCL_HIP(hipFuncSetAttribute((const void*)convq_fn(2, 12, st), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit));
        OCL_HIP(hipFuncSetAttribute((const void*)convq_fn(1, 12, st), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit));
    }
    done = true;
    return OCL_OK;
}

}  // namespace ocl
