// K1-K4: Reduced-ResNet18 convolution / batch-norm kernels for gfx950.
//
//  conv_t_kernel      implicit-GEMM 3x3 / 1x1 convolution on exact-fp32 MFMA (v_mfma_f32_16x16x4_f32), D[channel][pixel] tiles with
//                     K-grouped operands.  One generic "tap list + output lattice" geometry covers forward (stride 1/2), data
//                     gradient (stride 1; stride 2 as four parity classes, in one launch where the lattices coincide) and the
//                     1x1 shortcut.  The input patch (with halo) of a 64/128-pixel tile is staged ONCE in LDS and reused by
//                     all taps; weights are resident in LDS or stream through a double-buffered stage.  Epilogues from
//                     registers: BN batch statistics (fp64 atomics), folded eval-mode BN, residual, ReLU, masked residual.
//  conv_wgrad_kernel  weight gradient as a (tap,ci) x co GEMM reduced over pixels, split-K over pixel tiles,
//                     partials summed by wgrad_reduce_kernel straight into PyTorch's OIHW gradient.
//  bn_*               train-mode BatchNorm forward (normalise+residual+ReLU, running-stat update) and backward.
//
// Replaces the ATen sequences behind models/resnet.py:10-12,32-37,90-99 and their autograd.
#include "conv.h"
#include <string.h>
#include <algorithm>
#include <type_traits>
#include <cmath>

namespace ocl {

static const size_t kLdsLimit = 160 * 1024;      // hardware: 160 KiB per workgroup
static const size_t kLdsTarget = 72 * 1024;      // weight-gradient planner target (2 workgroups per CU)

// =====================================================================================================
// helpers shared by the convolution and the weight-gradient kernels
// =====================================================================================================
// exact u / d for 0 <= u < 2^22 with a precomputed float reciprocal (one correction step either way)
__device__ __forceinline__ int fdiv(int u, int d, float inv, int& rem) {
    int q = (int)((float)u * inv);
    int r = u - q * d;
    if (r < 0) { --q; r += d; }
    else if (r >= d) { ++q; r -= d; }
    rem = r;
    return q;
}

// 16-byte buffer load with a 32-bit byte offset; an offset of kOob (>= num_records of every descriptor made by
// make_rsrc) returns zeros in hardware: no exec-mask branch, no 64-bit address arithmetic, no select on the result.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int kOob = 0x7fffffff;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7ffffff0, 0x00020000);
}
__device__ __forceinline__ float4 buf_load16(__amdgpu_buffer_rsrc_t r, int byte_off) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// value of a small per-tap table at a block-uniform index, without dynamic indexing of the kernel-argument struct
// (which would spill it to scratch): a 9-way select chain on scalars.
__device__ __forceinline__ int tap_sel(const int (&tab)[9], int t) {
    int v = tab[0];
#pragma unroll
    for (int i = 1; i < 9; ++i) v = (t == i) ? tab[i] : v;
    return v;
}

constexpr int kPatchPF = 8;      // max float4 patch-prefetch registers per thread of the wgrad kernels
constexpr int kConvPatchPF = 8;  // ... of the conv kernel (planner: patch units <= 256*kConvPatchPF)


// =====================================================================================================
// BatchNorm arithmetic shared by every kernel that applies or differentiates a train-mode BatchNorm: one statement of the
// scale / shift (so that an activation recomputed from the raw convolution output -- consuming convolution, weight gradient,
// ReLU mask of the backward -- has the bits the BatchNorm kernel would have written)
// =====================================================================================================
__device__ __forceinline__ void bn_scale_shift(float gamma, float beta, float mean, float invstd, float& sc, float& sh) {
    sc = gamma * invstd;
    sh = __fmaf_rn(-mean, sc, beta);
}
// ---- batch sums (StatCell, conv.h) ---------------------------------------------------------------------------------------------
// Two ways to accumulate a cell, chosen at run time (ocl_set_deterministic / OCL_DETERMINISTIC=1, a __constant__ flag):
//  * default: the cell's first word holds a double and takes fp64 atomics (rounds 1 - 3): totals depend on the workgroups' arrival
//    order in the last bit;
//  * deterministic: 2^-40 fixed point added as two 64-bit INTEGERS (associative): bit-identical totals whatever the order.  Costs
//    two atomics per partial sum instead of one: +12 % on the SCR step, +13 % on ER (profiles/r4_batch_sums_ab.txt) -- which is why
//    it is a mode and not the default.
__constant__ int g_det_sums = 0;
// MODE -1: read the flag at run time; 0 / 1: compiled for the default / deterministic mode only (conv_s_kernel: a 96-register kernel
// that cannot carry both paths without spilling -- its two instantiations are chosen by the host's copy of the flag)
template <int MODE>
__device__ __forceinline__ bool fx_det() { return MODE < 0 ? g_det_sums != 0 : MODE == 1; }

__device__ __forceinline__ void fx_split(double v, long long& hi, unsigned long long& lo) {
    if (fabs(v) < 7.0e13) {                                   // (false for NaN / Inf as well)
        const double q = v * 1099511627776.0;                 // v * 2^40: exact
        const double h = floor(q * (1.0 / 4294967296.0));     // floor(q / 2^32)
        hi = (long long)h;
        lo = (unsigned long long)(q - h * 4294967296.0);      // [0, 2^32): truncating it to an integer is the only rounding (< 2^-40)
    } else {
        hi = 1ll << 56;                                       // poison: reads back as NaN
        lo = 0ull;
    }
}
template <int MODE = -1>
__device__ __forceinline__ void fx_add(StatCell* cell, double v) {
    if (!fx_det<MODE>()) {
        atomicAdd((double*)&cell->lo, v);
        return;
    }
    long long hi;
    unsigned long long lo;
    fx_split(v, hi, lo);
    atomicAdd(&cell->lo, lo);
    atomicAdd((unsigned long long*)&cell->hi, (unsigned long long)hi);
}
template <int MODE = -1>
__device__ __forceinline__ double fx_decode(long long hi, unsigned long long lo) {
    if (!fx_det<MODE>()) return __longlong_as_double((long long)lo);
    if (hi >= (1ll << 55) || hi <= -(1ll << 55)) return __builtin_nan("");
    return (double)hi * (1.0 / 256.0) + (double)lo * (1.0 / 1099511627776.0);
}
typedef unsigned long long u64x2_t __attribute__((ext_vector_type(2)));
// the total of a cell's kStatReps replicas (deterministic mode: integer sums, exact in any order; default: the replicas in a fixed
// order).  All replicas are requested before any is consumed: left to itself the compiler waited for each 16-byte load before issuing
// the next -- eight dependent L2 round trips in the prologue of every kernel that reads a statistic.
// B: replicas in flight at once (4 registers each): 8 by default, 4 in conv_s_kernel's prologue (a 96-register kernel: with all
// sixteen loads of a (sum, sum of squares) pair in flight it spilled 50 VGPRs to scratch)
template <int B = kStatReps, int MODE = -1>
__device__ __forceinline__ double fx_total(const StatCell* __restrict__ cells, int64_t rep_stride, int64_t idx) {
    static_assert(kStatReps % B == 0, "batch divides the replica count");
    if (!fx_det<MODE>()) {   // the replicas in a fixed order
        double t = 0.0;
#pragma unroll
        for (int r0 = 0; r0 < kStatReps; r0 += B) {
            double c[B];
#pragma unroll
            for (int r = 0; r < B; ++r) c[r] = *(const double*)&cells[(r0 + r) * rep_stride + idx].lo;
#pragma unroll
            for (int r = 0; r < B; ++r) t += c[r];
        }
        return t;
    }
    long long hi = 0;
    unsigned long long lo = 0;
    bool bad = false;
#pragma unroll
    for (int r0 = 0; r0 < kStatReps; r0 += B) {
        u64x2_t c[B];
#pragma unroll
        for (int r = 0; r < B; ++r) c[r] = *(const u64x2_t*)(cells + (r0 + r) * rep_stride + idx);
#pragma unroll
        for (int r = 0; r < B; ++r) {
            const long long h = (long long)c[r].y;
            bad |= h >= (1ll << 55) || h <= -(1ll << 55);
            hi += h;
            lo += c[r].x;
        }
    }
    return bad ? __builtin_nan("") : fx_decode<MODE>(hi, lo);
}
// two totals at once: all 2 * kStatReps loads in flight together
__device__ __forceinline__ void fx_total2(const StatCell* __restrict__ cells, int64_t rep_stride, int64_t idx1, int64_t idx2, double& t1, double& t2) {
    u64x2_t a[kStatReps], b[kStatReps];
#pragma unroll
    for (int r = 0; r < kStatReps; ++r) {
        a[r] = *(const u64x2_t*)(cells + r * rep_stride + idx1);
        b[r] = *(const u64x2_t*)(cells + r * rep_stride + idx2);
    }
    if (!g_det_sums) {
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int r = 0; r < kStatReps; ++r) {
            s1 += __longlong_as_double((long long)a[r].x);
            s2 += __longlong_as_double((long long)b[r].x);
        }
        t1 = s1; t2 = s2;
        return;
    }
    long long h1 = 0, h2 = 0;
    unsigned long long l1 = 0, l2 = 0;
    bool bad1 = false, bad2 = false;
#pragma unroll
    for (int r = 0; r < kStatReps; ++r) {
        const long long x = (long long)a[r].y, y = (long long)b[r].y;
        bad1 |= x >= (1ll << 55) || x <= -(1ll << 55);
        bad2 |= y >= (1ll << 55) || y <= -(1ll << 55);
        h1 += x; l1 += a[r].x;
        h2 += y; l2 += b[r].x;
    }
    t1 = bad1 ? __builtin_nan("") : fx_decode(h1, l1);
    t2 = bad2 ? __builtin_nan("") : fx_decode(h2, l2);
}
// the same with returning device-scope atomics / device-scope atomic loads (bn_bwd_fused_kernel: the adds must have executed at the
// coherence point before the wave signals its arrival; the totals are read while other workgroups may still be spinning)
__device__ __forceinline__ unsigned long long fx_fetch_add(StatCell* cell, double v) {
    if (!g_det_sums)
        return (unsigned long long)__double_as_longlong(__hip_atomic_fetch_add((double*)&cell->lo, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    long long hi;
    unsigned long long lo;
    fx_split(v, hi, lo);
    return __hip_atomic_fetch_add(&cell->lo, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) +
           __hip_atomic_fetch_add((unsigned long long*)&cell->hi, (unsigned long long)hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double fx_total_atomic(const StatCell* cells, int64_t rep_stride, int64_t idx) {
    if (!g_det_sums) {
        double t = 0.0;
        for (int r = 0; r < kStatReps; ++r)
            t += __hip_atomic_load((const double*)&cells[r * rep_stride + idx].lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return t;
    }
    long long hi = 0;
    unsigned long long lo = 0;
    bool bad = false;
    for (int r = 0; r < kStatReps; ++r) {
        const StatCell* c = cells + r * rep_stride + idx;
        const long long h = (long long)__hip_atomic_load((const unsigned long long*)&c->hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bad |= h >= (1ll << 55) || h <= -(1ll << 55);
        hi += h;
        lo += __hip_atomic_load(&c->lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return bad ? __builtin_nan("") : fx_decode(hi, lo);
}
static int g_det_host = 0;   // the host's copy of g_det_sums (conv_s_kernel's instantiation is chosen by it)
int set_deterministic_sums(int on) {
    const int v = on ? 1 : 0;
    g_det_host = v;
    OCL_HIP(hipDeviceSynchronize());   // (no launch may straddle the switch: the cells are interpreted by the flag)
    OCL_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_det_sums), &v, sizeof(int)));
    return OCL_OK;
}

// mean / invstd of (group g, channel c) from the replicated batch sums (biased variance, nn.BatchNorm2d's normalisation)
template <int B = 2 * kStatReps, int MODE = -1>   // replica loads in flight (see fx_total)
__device__ __forceinline__ void bn_batch_moments(const StatCell* __restrict__ stats, int64_t rep_stride, int g, int c, int C, double M, float eps,
                                                 double& mean, double& var) {
    double s1, s2;
    if constexpr (B >= 2 * kStatReps) {
        fx_total2(stats, rep_stride, ((int64_t)g * 2 + 0) * C + c, ((int64_t)g * 2 + 1) * C + c, s1, s2);
    } else {
        s1 = fx_total<B, MODE>(stats, rep_stride, ((int64_t)g * 2 + 0) * C + c);
        s2 = fx_total<B, MODE>(stats, rep_stride, ((int64_t)g * 2 + 1) * C + c);
    }
    mean = s1 / M;
    var = s2 / M - mean * mean;
    if (var < 0.0) var = 0.0;
    (void)eps;
}
// running statistics: one update per group, in order (= the reference's separate forward calls), unbiased variance, momentum
template <int MODE = -1>
__device__ __forceinline__ void bn_running_update(const StatCell* __restrict__ stats, int64_t rep_stride, int G, int C, double M, float momentum,
                                                  float eps, float* __restrict__ running_mean, float* __restrict__ running_var,
                                                  int64_t* __restrict__ nbt, int tid, int nthreads) {
    for (int c = tid; c < C; c += nthreads) {
        float rm = running_mean[c], rv = running_var[c];
        for (int gg = 0; gg < G; ++gg) {
            double mean, var;
            bn_batch_moments<1, MODE>(stats, rep_stride, gg, c, C, M, eps, mean, var);   // (one workgroup per launch runs this: few loads in flight, few registers)
            const double unb = M > 1.0 ? var * M / (M - 1.0) : var;
            rm = momentum * (float)mean + (1.f - momentum) * rm;
            rv = momentum * (float)unb + (1.f - momentum) * rv;
        }
        running_mean[c] = rm;
        running_var[c] = rv;
    }
    if (tid == 0 && nbt) *nbt += G;
}

// ---- EPI_BNB: the reduction half of a BatchNorm backward in the epilogue of the data gradient that produces its input gradient ------
// table [groups][Cout/4][3][4]: scale quad, shift quad (the forward's bn_scale_shift: the recomputed ReLU mask has the forward's bits),
// mean quad
__device__ __forceinline__ void bnb_table(const ConvArgs& a, float* tab, int tid, int nthreads) {
    const int C = a.Cout;
    for (int j = tid; j < a.groups * C; j += nthreads) {
        const int gq = j / C, c = j - gq * C;
        const float mean = a.bnb_mean[j];
        float sc, sh;
        bn_scale_shift(a.bnb_gamma[c], a.bnb_beta[c], mean, a.bnb_invstd[j], sc, sh);
        float* t = tab + (size_t)(gq * (C >> 2) + (c >> 2)) * 12 + (c & 3);
        t[0] = sc;
        t[4] = sh;
        t[8] = mean;
    }
}
// one channel quad of one pixel: v = gradient w.r.t. the ReLU'd BatchNorm output (complete); masks it and adds to the lane's partial sums
__device__ __forceinline__ void bnb_apply(const ConvArgs& a, const float4 sc, const float4 sh, const float4 mu, int64_t eo, float4& v,
                                          float (&s1)[4], float (&s2)[4]) {
    const float4 y = *(const float4*)(a.bnb_y + eo);
    float4 zz;
    if (a.bnb_z) zz = *(const float4*)(a.bnb_z + eo);
    else zz = make_float4(__fmaf_rn(y.x, sc.x, sh.x), __fmaf_rn(y.y, sc.y, sh.y), __fmaf_rn(y.z, sc.z, sh.z), __fmaf_rn(y.w, sc.w, sh.w));
    v.x = zz.x > 0.f ? v.x : 0.f; v.y = zz.y > 0.f ? v.y : 0.f; v.z = zz.z > 0.f ? v.z : 0.f; v.w = zz.w > 0.f ? v.w : 0.f;
    s1[0] += v.x; s1[1] += v.y; s1[2] += v.z; s1[3] += v.w;
    s2[0] = fmaf(v.x, y.x - mu.x, s2[0]); s2[1] = fmaf(v.y, y.y - mu.y, s2[1]);
    s2[2] = fmaf(v.z, y.z - mu.z, s2[2]); s2[3] = fmaf(v.w, y.w - mu.w, s2[3]);
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
// x / d for a plan constant d through its precomputed M = ceil(2^32 / d): exact for x * d < 2^32 (checked by the planner)
__device__ __forceinline__ int mdiv(int x, unsigned M, int d, int& rem) {
    const int q = d == 1 ? x : (int)__umulhi((unsigned)x, M);
    rem = x - q * d;
    return q;
}
constexpr int kWPF = 4;                        // float4 weight-prefetch registers per thread (staged weights)
constexpr size_t kResidentBytes = 80 * 1024;   // weights of one channel split kept in LDS for the workgroup's lifetime up to this

// sum over the 16 lanes of a DPP row (lanes 16k .. 16k+15), result in every lane: four v_add_f32 with DPP operands, no LDS traffic
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));   // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));   // row_mirror
    return v;
}
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
constexpr int kQBlocks = 5;   // blocks of four output channels (Cout <= 20)
template <int NTQ, int PF, int STATS>   // 0: no sums; 1: forward batch statistics (EPI_STATS); 2: BatchNorm-backward sums (EPI_BNB)
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
        store_patch(d0.z, d1.y);
        if (k + 1 < nwt) load_patch_d(*(const int4*)(tdesc + (k + 1) * 8));
        __syncthreads();   // patch (and, the first time, the weights and the transform table) visible
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
    }
    if ((flags & (EPI_STATS | EPI_BNB)) && run_grp >= 0) flush_stats();
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
        for (int j = tid; j < a.groups * C; j += 256) {
            const int gq = j / C, c = j - gq * C;
            double mean, var;
            bn_batch_moments<1, FXM>(a.xf_stats, a.xf_rep_stride, gq, c, C, M, a.xf_eps, mean, var);
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
            static const size_t res_limit = [] { const char* e = getenv("OCL_RES_KB"); return e ? (size_t)atoi(e) * 1024 : kResidentBytes; }();   // (measurement knob)
            a.wres = (KC == g.Cin && w_all <= res_limit) ? 1 : 0;
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
    static const int env_units = [] { const char* e = getenv("OCL_CONV_S_UNITS"); return e ? atoi(e) : 1000; }();   // measurement knob
    if (g.force_cs <= 0 && ((LP > 16 && tiles64 * cdiv(g.Cout, 16) >= env_units) || a.Qc <= 36)) return OCL_ERR_ARG;
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

static conv_fn_t convq_fn(int ntq, int pf, int stats) {   // stats: 0 none, 1 EPI_STATS, 2 EPI_BNB
#define OCL_CASE(N, P)                                                                                                         \
    if (ntq == N && pf == P) return stats == 2 ? conv_q_kernel<N, P, 2> : stats == 1 ? conv_q_kernel<N, P, 1> : conv_q_kernel<N, P, 0>;
    OCL_CASE(2, 4) OCL_CASE(2, 12) OCL_CASE(1, 12)
#undef OCL_CASE
    return nullptr;
}

int launch_conv(const ConvPlan& p, hipStream_t s) {
    if (p.cs) {
        if (!p.a.blob) {
            set_error("launch_conv: plan without device tables (conv_plan_finalize)");
            return OCL_ERR_STATE;
        }
        ProfScope ps(PROF_CONV, s);
        hipLaunchKernelGGL(convs_fn(p.NT, p.a.trace != nullptr && !g_det_host, (p.a.flags & EPI_BNB) != 0, g_det_host != 0), dim3(p.grid_x, p.grid_y), dim3(256),
                           p.lds_bytes, s, p.a);
        OCL_LAUNCH_CHECK();
        return OCL_OK;
    }
    if (p.q4) {
        conv_fn_t fq = convq_fn(p.q4, (p.a.off_loc - p.a.off_pu) / (3 * 256), (p.a.flags & EPI_BNB) ? 2 : (p.a.flags & EPI_STATS) ? 1 : 0);
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
// weight gradient
// =====================================================================================================
// Weight gradient: (tap, ci) x co GEMM reduced over pixels.  A workgroup owns one (channel chunk, row block, column
// block) output tile and a strided subset of the pixel tiles; like the forward kernel it is software-pipelined: the dy
// tile and the input patch of the NEXT pixel tile are fetched into registers while the MFMAs of the current one run.
struct WTile {
    int img0, p0, oy0, nrows;
};

//
// RGW > 0 selects the 4x4x1 form of the product for layers with at most 20 output channels and one channel chunk (stem, layer 1;
// EXPERIMENTAL, OCL_WGRAD_Q=1, see plan_wgrad).  The 16x16x4 tiles pad layer 1's 180 x 20 gradient to 192 x 32: 41 % of the issued
// MFMAs multiply zeros, and by the probes of round 3 the kernel is bound by the MFMAs it issues.  With v_mfma_f32_4x4x1_16b_f32 the
// sixteen blocks of an instruction are sixteen PIXELS (the reduction dimension), and nothing is padded beyond quads:
//   block b = pixel s0 + b of the tile;   A: lane 4b + i holds the 16-byte unit u = 4 * rowgroup + i = (tap, channel quad) of that
//   pixel's patch (one ds_read_b128 feeds four MFMAs, k = channel inside the quad);   B: lane 4b + j holds dy[pixel][4s + j];
//   D: register e of lane 4b + j accumulates  x[unit 4 * rowgroup + e][k] * dy[4s + j]  summed over the pixels b, b + 16, ...
// so a wave owns RGW row groups (16 rows each) x 5 column quads x 4 channels = 20 * RGW accumulators, fed by RGW + 5 operand reads
// per 16 pixels, and the sixteen per-block partial sums are combined once, at the end (two DPP row shifts, two cross-row shuffles),
// in a fixed order.  The slab format is the 16x16x4 form's: the reduction kernels do not know which form wrote it.
//
// TAB = 1: the staging of a pixel tile with its tile-invariant half precomputed.  Which pixel slot / channel quad / patch position a
// thread's units are does not change from tile to tile; only the tile's base addresses and its validity limits do.  The TAB = 0 form
// re-derives everything per tile from packed positions (~1100 instructions per tile and wave around ~600 of the K loop); here the
// thread keeps per unit a constant byte offset, an LDS address and a packed (row, patch row, image) word, and a tile costs an add,
// two compares and a select per load.  Units that lie outside the tile or the patch for good store into a 16-byte dummy slot in
// front of the pixel table instead of branching around the store.  Same values into the same LDS cells: bit-identical results.
// TRACE = 1 (measurement build, kbench `wgradtrace`): s_memtime stamps of thread 0 at the phase boundaries into WgradArgs::trace,
// 64 slots per workgroup: start | prologue done | per tile: passed barrier 1, tile stored, passed barrier 2, next tile's loads issued,
// K loop done | ... | slab written.
template <int MTW, int NTW, int PF, int RGW = 0, int TAB = 0, int TRACE = 0>
__global__ void __launch_bounds__(256) conv_wgrad_kernel(const WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    int* pixoff = (int*)lds_raw + 4;                // [KP]   (in front of it: the dummy slot of the TAB form)
    float* dyt = (float*)(pixoff + a.KP);           // [KP][DP]
    float* patch = dyt + (size_t)a.KP * a.DP;       // [imgs][PR][PC][CP]
    float* xft = patch + (((size_t)a.imgs * a.PR * a.PC * a.CP + 3) & ~(size_t)3);   // input transform (WgradArgs::xf): [groups][Cin/4][2][4] scale / shift quads
    constexpr int BNW = RGW > 0 ? 4 * kQBlocks : 16 * NTW;
    constexpr int Q = BNW / 4;
    constexpr int DPF = (128 * Q + 255) / 256;      // dy prefetch registers (KP <= 128)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, g = lane >> 4;
    // (pixel split bx, output block by) of this workgroup.  xcd_by > 0 (OCL_WGRAD_XCD=1, a one-dimensional launch of S * by workgroups):
    // the `by` workgroups that read the SAME pixel tiles get linear ids that agree modulo 8 and lie within 8 * by of each other --
    // workgroup b is observed to run on XCD b % 8, so they share one L2 (4 MB per XCD, not coherent across XCDs) at about the same
    // time, instead of fetching every tile once per output block from memory.  The last S % 8 splits keep the plain order.
    int bx = blockIdx.x, by = blockIdx.y;
    if (a.xcd_by > 0) {
        const int id = blockIdx.x, per = 8 * a.xcd_by, full = (a.S >> 3) * per;
        if (id < full) {
            const int grp = id / per, rem = id - grp * per;
            by = rem >> 3;
            bx = grp * 8 + (rem & 7);
        } else {
            const int r = a.S & 7, t = id - full;
            by = t / r;
            bx = (a.S & ~7) + (t - by * r);
        }
    }
    const int nb = by % a.nblocks;
    const int t1 = by / a.nblocks;
    const int mb = t1 % a.mblocks_per_chunk;
    const int chunk = t1 / a.mblocks_per_chunk;
    const int c0 = chunk * a.KC;
    const int n0 = nb * BNW;
    const int m0 = mb * 64 * MTW;
    const int LP = a.Ho * a.Wo;

    int aoff[MTW];
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) {
        const int m = min(m0 + wave * 16 * MTW + mt * 16 + r16, a.Mchunk - 1);
        const int t = m / a.KC, cc = m - t * a.KC;
        aoff[mt] = ((a.tdy[t] - a.min_dy) * a.PC + (a.tdx[t] - a.min_dx)) * a.CP + cc;
    }
    f32x4 acc[MTW][NTW];
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // 4x4x1 form: LDS offset of the lane's unit inside a pixel's patch (0 for the units past the last one: their rows are not written)
    constexpr int RG = RGW > 0 ? RGW : 1;
    int qoff[RG];
    f32x4 qacc[RG][kQBlocks][4];
    if constexpr (RGW > 0) {
        const int qk4 = a.KC >> 2, units = a.ntaps * qk4;
#pragma unroll
        for (int r = 0; r < RG; ++r) {
            const int u = ((by * 4 + wave) * RG + r) * 4 + (lane & 3);
            int c4;
            const int t = fdiv(min(u, units - 1), qk4, 1.0f / (float)qk4, c4);
            qoff[r] = u < units ? ((tap_sel(a.tdy, t) - a.min_dy) * a.PC + (tap_sel(a.tdx, t) - a.min_dx)) * a.CP + c4 * 4 : 0;
#pragma unroll
            for (int s = 0; s < kQBlocks; ++s)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    qacc[r][s][k] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    asm volatile("" : "+a"(qacc[r][s][k]));
                }
        }
    }

    const float inv_tpi = 1.0f / (float)a.tiles_per_img, inv_wo0 = 1.0f / (float)a.Wo;
    auto geom = [&](int tile) __attribute__((always_inline)) -> WTile {
        WTile t;
        int tp, rem;
        const int ti = fdiv(tile, a.tiles_per_img, inv_tpi, tp);
        t.img0 = ti * a.imgs;
        t.p0 = tp * a.ppi;
        t.oy0 = fdiv(t.p0, a.Wo, inv_wo0, rem);
        const int pend = min(t.p0 + a.ppi, LP);
        const int oy1 = fdiv(pend - 1, a.Wo, inv_wo0, rem);
        t.nrows = a.imgs > 1 ? min(a.imgs, a.N - t.img0) * a.PR : (oy1 - t.oy0) * a.stride + (a.max_dy - a.min_dy) + 1;
        return t;
    };

    // ---- per-thread unit bookkeeping (identical for every tile) ----------------------------------------------------
    // dy units: u = tid + i*256 -> (pixel slot q, float4 column c4); packed q << 8 | c4, -1 past the tile
    const int kc4 = a.KC >> 2;
    const float inv_ppi = 1.0f / (float)a.ppi, inv_wo = 1.0f / (float)a.Wo;
    int du_pos[DPF];
#pragma unroll
    for (int i = 0; i < DPF; ++i) {
        const int u = tid + i * 256;
        int c4;
        const int q = fdiv(u, Q, 1.0f / (float)Q, c4);
        du_pos[i] = q < a.KP ? (q << 8) | c4 : -1;
    }
    // patch units: flat [row][pc][c4], packed il << 24 | pr << 16 | pc << 8 | c4
    int pu_pos[PF];
    {
        int c4, pc;
        const int pix = fdiv(tid, kc4, 1.0f / (float)kc4, c4);
        int row = fdiv(pix, a.PC, 1.0f / (float)a.PC, pc);
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            int il = 0, pr = row;
            if (a.imgs > 1) il = fdiv(row, a.PR, a.inv_PR, pr);
            pu_pos[i] = (il << 24) | (pr << 16) | (pc << 8) | c4;
            if (il >= 128 || pr >= 256) pu_pos[i] = 0x7fff0000;   // past any tile's last row
            c4 += a.d_c4;
            pc += a.d_pc;
            if (c4 >= kc4) { c4 -= kc4; pc += 1; }
            row += a.d_row;
            if (pc >= a.PC) { pc -= a.PC; row += 1; }
        }
    }
    float4 dv[DPF], pv[PF];
    unsigned okm = 0;   // bit i: patch unit i of the tile in flight lies inside the image (input transform: the others stay zero)
    if (a.xf) {   // x is a raw convolution output: its BatchNorm + ReLU is applied while the patch is staged (first barrier of the tile loop publishes the table)
        const int C = a.Cin;
        for (int j = threadIdx.x; j < a.xf_groups * C; j += 256) {
            const int gq = j / C, c = j - gq * C;
            float sc, sh;
            bn_scale_shift(a.xf_gamma[c], a.xf_beta[c], a.xf_mean[j], a.xf_invstd[j], sc, sh);
            float* t = xft + (size_t)(gq * (C >> 2) + (c >> 2)) * 8 + (c & 3);
            t[0] = sc;
            t[4] = sh;
        }
    }
    const float inv_gs = a.xf ? 1.0f / (float)a.xf_group_size : 0.f;
    int dpo[DPF];   // LDS patch offset of the pixel (units with c4 == 0 publish it), -1: unit not in this tile
    const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(a.x), rs_dy = make_rsrc(a.dy);
    auto load_tile = [&](const WTile& t) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < DPF; ++i) {
            const int q = du_pos[i] >> 8, c4 = du_pos[i] & 255;
            int pl, ox;
            const int il = fdiv(q, a.ppi, inv_ppi, pl);
            const int p = t.p0 + pl, n = t.img0 + il;
            const int oy = fdiv(p, a.Wo, inv_wo, ox);
            const bool in_tile = du_pos[i] >= 0;
            const bool v = in_tile & (il < a.imgs) & (n < a.N) & (p < LP);
            const int co = n0 + c4 * 4;
            const bool ok = v & (co < a.Cout);
            dv[i] = buf_load16(rs_dy, ok ? (((n * a.Ho + oy) * a.Wo + ox) * a.Cout + co) * 4 : kOob);   // zeros when masked
            dpo[i] = in_tile ? (v ? ((il * a.PR + (oy - t.oy0) * a.stride) * a.PC + ox * a.stride) * a.CP : 0) : -1;
        }
        const int iy0 = t.oy0 * a.stride + a.min_dy;
        const int base = (((t.img0 * a.Hin + iy0) * a.Win + a.min_dx) * a.Cin + c0) * 4;   // bytes; may be negative (halo)
        okm = 0;
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            const int il = pu_pos[i] >> 24, pr = (pu_pos[i] >> 16) & 255, pc = (pu_pos[i] >> 8) & 255, c4 = pu_pos[i] & 255;
            const int iy = iy0 + pr, ix = a.min_dx + pc;
            const bool ok = (il * a.PR + pr < t.nrows) & (iy >= 0) & (iy < a.Hin) & (ix >= 0) & (ix < a.Win);
            pv[i] = buf_load16(rs_x, ok ? base + (((il * a.Hin + pr) * a.Win + pc) * a.Cin + c4 * 4) * 4 : kOob);
            okm |= ok ? (1u << i) : 0u;
        }
    };
    auto store_tile = [&](const WTile& t) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < DPF; ++i)
            if (dpo[i] >= 0) {
                const int q = du_pos[i] >> 8, c4 = du_pos[i] & 255;
                if (c4 == 0) pixoff[q] = dpo[i];
                *(float4*)(dyt + (size_t)q * a.DP + c4 * 4) = dv[i];
            }
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            const int il = pu_pos[i] >> 24, pr = (pu_pos[i] >> 16) & 255, pc = (pu_pos[i] >> 8) & 255, c4 = pu_pos[i] & 255;
            const int row = il * a.PR + pr;
            if (row < t.nrows) {
                float4 v = pv[i];
                if (a.xf) {   // block-uniform
                    int rem;
                    const int gq = min(fdiv(t.img0 + il, a.xf_group_size, inv_gs, rem), a.xf_groups - 1);
                    const float* tb = xft + (size_t)(gq * (a.Cin >> 2) + (c0 >> 2) + c4) * 8;
                    const float4 sc = *(const float4*)tb, sh = *(const float4*)(tb + 4);
                    v.x = fmaxf(__fmaf_rn(v.x, sc.x, sh.x), 0.f); v.y = fmaxf(__fmaf_rn(v.y, sc.y, sh.y), 0.f);
                    v.z = fmaxf(__fmaf_rn(v.z, sc.z, sh.z), 0.f); v.w = fmaxf(__fmaf_rn(v.w, sc.w, sh.w), 0.f);
                    if (!((okm >> i) & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f);
                }
                float* d = patch + (row * a.PC + pc) * a.CP + c4 * 4;
                *(float2*)d = make_float2(v.x, v.y);
                *(float2*)(d + 2) = make_float2(v.z, v.w);
            }
        }
    };

    // ---- TAB form of the same two steps -------------------------------------------------------------------------------
    constexpr int TDPF = TAB ? DPF : 1, TPF = TAB ? PF : 1;
    int d_pl[TDPF], d_il[TDPF], d_goff[TDPF], d_lds[TDPF];   // pixel inside its image (huge: never valid), image inside the tile, byte offset from the tile's dy base, LDS float offset of the quad (dummy slot when the unit is outside the tile)
    int px_pl = 0, px_il = 0, px_idx = -4, pxo = 0;          // the pixel-table entry of pixel slot `tid` (threads past the tile write the dummy slot)
    int p_word[TPF], p_goff[TPF], p_lds[TPF];   // row | patch row << 16 | image << 24 (row 0xffff: never loaded);  byte offset from the tile's x base;  LDS byte offset | channel quad << 24
    if constexpr (TAB) {
        const int dummy_f = (int)((float*)lds_raw - dyt);   // float offset of the dummy slot relative to dyt (negative)
#pragma unroll
        for (int i = 0; i < DPF; ++i) {
            const int u = tid + i * 256;
            int c4, pl;
            const int q = fdiv(u, Q, 1.0f / (float)Q, c4);
            const int il = fdiv(q, a.ppi, inv_ppi, pl);
            const bool in_tile = q < a.KP;
            d_pl[i] = (in_tile && il < a.imgs && n0 + c4 * 4 < a.Cout) ? pl : 0x20000000;
            d_il[i] = il;
            d_goff[i] = ((il * LP + pl) * a.Cout + n0 + c4 * 4) * 4;
            d_lds[i] = in_tile ? q * a.DP + c4 * 4 : dummy_f;
        }
        {
            int pl;
            const int il = fdiv(tid, a.ppi, inv_ppi, pl);
            px_pl = (tid < a.KP && il < a.imgs) ? pl : 0x20000000;
            px_il = il;
            px_idx = tid < a.KP ? tid : -4;
        }
        const int n_units = a.imgs * a.PR * a.PC * kc4;
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            const int u = tid + i * 256;
            int c4, pc, pr;
            const int pix = fdiv(u, kc4, 1.0f / (float)kc4, c4);
            const int row = fdiv(pix, a.PC, 1.0f / (float)a.PC, pc);
            const int il = fdiv(row, a.PR, a.inv_PR, pr);
            const int ix = a.min_dx + pc;
            const bool inside = u < n_units;
            const bool x_ok = inside && ix >= 0 && ix < a.Win;
            p_word[i] = (x_ok ? row : 0xffff) | ((pr & 255) << 16) | ((il & 127) << 24);
            p_goff[i] = (((il * a.Hin + pr) * a.Win + pc) * a.Cin + c4 * 4) * 4;
            p_lds[i] = (inside ? (int)((patch - (float*)lds_raw) + (row * a.PC + pc) * a.CP + c4 * 4) * 4 : 0) | (c4 << 24);
        }
    }
    auto load_tile_t = [&](const WTile& t) __attribute__((always_inline)) {
        const int dbase = (t.img0 * LP + t.p0) * a.Cout * 4;
        const int ox0 = t.p0 - t.oy0 * a.Wo;
#pragma unroll
        for (int i = 0; i < TDPF; ++i) {
            const int p = t.p0 + d_pl[i], n = t.img0 + d_il[i];
            const bool v = (p < LP) & (n < a.N);
            dv[i] = buf_load16(rs_dy, v ? dbase + d_goff[i] : kOob);   // zeros when masked
        }
        {   // LDS patch offset of pixel slot `tid`: (row, column) of the pixel relative to the tile's first output row (branch-free division)
            const int p = t.p0 + px_pl, n = t.img0 + px_il;
            const bool v = (p < LP) & (n < a.N);
            const int r = (ox0 + px_pl) & 0x3fffff;
            int dr = (int)((float)r * inv_wo), ox = r - dr * a.Wo;
            const int lo = ox < 0 ? 1 : 0, hi = ox >= a.Wo ? 1 : 0;
            dr += hi - lo;
            ox += (lo - hi) * a.Wo;
            pxo = v ? ((px_il * a.PR + dr * a.stride) * a.PC + ox * a.stride) * a.CP : 0;
        }
        const int iy0 = t.oy0 * a.stride + a.min_dy;
        const int base = (((t.img0 * a.Hin + iy0) * a.Win + a.min_dx) * a.Cin + c0) * 4;   // bytes; may be negative (halo)
        okm = 0;
#pragma unroll
        for (int i = 0; i < TPF; ++i) {
            const int row = p_word[i] & 0xffff, iy = iy0 + ((p_word[i] >> 16) & 255);
            const bool ok = (row < t.nrows) & (iy >= 0) & (iy < a.Hin);
            pv[i] = buf_load16(rs_x, ok ? base + p_goff[i] : kOob);
            okm |= ok ? (1u << i) : 0u;
        }
    };
    auto store_tile_t = [&](const WTile& t) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < TDPF; ++i) *(float4*)(dyt + d_lds[i]) = dv[i];
        pixoff[px_idx] = pxo;
#pragma unroll
        for (int i = 0; i < TPF; ++i) {
            float4 v = pv[i];
            if (a.xf) {   // block-uniform
                int rem;
                const int gq = min(fdiv(t.img0 + (p_word[i] >> 24), a.xf_group_size, inv_gs, rem), a.xf_groups - 1);
                const float* tb = xft + (size_t)(gq * (a.Cin >> 2) + (c0 >> 2) + ((unsigned)p_lds[i] >> 24)) * 8;
                const float4 sc = *(const float4*)tb, sh = *(const float4*)(tb + 4);
                v.x = fmaxf(__fmaf_rn(v.x, sc.x, sh.x), 0.f); v.y = fmaxf(__fmaf_rn(v.y, sc.y, sh.y), 0.f);
                v.z = fmaxf(__fmaf_rn(v.z, sc.z, sh.z), 0.f); v.w = fmaxf(__fmaf_rn(v.w, sc.w, sh.w), 0.f);
                if (!((okm >> i) & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            float* d = (float*)(lds_raw + (p_lds[i] & 0xffffff));
            *(float2*)d = make_float2(v.x, v.y);
            *(float2*)(d + 2) = make_float2(v.z, v.w);
        }
    };

    int tr_n = 0;
    auto stamp = [&]() __attribute__((always_inline)) {
        if constexpr (TRACE) {
            if (tid == 0 && tr_n < 64) a.trace[(size_t)(by * a.S + bx) * 64 + tr_n++] = __builtin_amdgcn_s_memtime();
        }
    };
    stamp();
    int tile = bx;
    WTile cur = geom(tile);
    if (tile < a.total_tiles) {
        if constexpr (TAB) load_tile_t(cur);
        else load_tile(cur);
    }
    stamp();
    for (; tile < a.total_tiles; tile += a.S) {
        __syncthreads();  // previous tile consumed
        stamp();
        if constexpr (TAB) store_tile_t(cur);
        else store_tile(cur);
        stamp();
        __syncthreads();
        stamp();
        const int next = tile + a.S;
        if (next < a.total_tiles) {
            cur = geom(next);
            if constexpr (TAB) load_tile_t(cur);
            else load_tile(cur);
        }
        stamp();
        if constexpr (RGW > 0) {   // 16 pixels per step (KP is a multiple of 16 in this form); operands of step g + 1 are read while the MFMAs of step g issue
            const float* dq = dyt + (size_t)(lane >> 2) * a.DP + (lane & 3);
            const int* pq = pixoff + (lane >> 2);
            const int ng = a.KP >> 4;
            float bv[2][kQBlocks];
            float4 av[2][RG];
            int po_n = pq[0];   // (the pixel's patch offset is read one step ahead of the operand reads that depend on it)
            auto fetch = [&](int set, int g) __attribute__((always_inline)) {
                const int po = po_n;
                po_n = pq[min(g + 1, ng - 1) * 16];
#pragma unroll
                for (int r = 0; r < RG; ++r) av[set][r] = *(const float4*)(patch + po + qoff[r]);
#pragma unroll
                for (int s = 0; s < kQBlocks; ++s) bv[set][s] = dq[(size_t)g * 16 * a.DP + 4 * s];
            };
            auto fma = [&](int set) __attribute__((always_inline)) {
#pragma unroll
                for (int r = 0; r < RG; ++r)
#pragma unroll
                    for (int s = 0; s < kQBlocks; ++s) {
                        qacc[r][s][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[set][r].x, bv[set][s], qacc[r][s][0], 0, 0, 0);
                        qacc[r][s][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[set][r].y, bv[set][s], qacc[r][s][1], 0, 0, 0);
                        qacc[r][s][2] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[set][r].z, bv[set][s], qacc[r][s][2], 0, 0, 0);
                        qacc[r][s][3] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[set][r].w, bv[set][s], qacc[r][s][3], 0, 0, 0);
                    }
            };
            fetch(0, 0);
            int g = 0;
            for (; g + 2 <= ng; g += 2) {
                fetch(1, g + 1);
                fma(0);
                if (g + 2 < ng) fetch(0, g + 2);
                fma(1);
            }
            if (g < ng) fma(0);
            // (the accumulators are pinned to AccVGPRs across the tile loop: left alone, the register allocator keeps them in ArchVGPRs
            // outside the pixel loop and copies all 80 * RGW of them in and out per tile -- spilling at RGW = 3)
#pragma unroll
            for (int r = 0; r < RG; ++r)
#pragma unroll
                for (int s = 0; s < kQBlocks; ++s)
#pragma unroll
                    for (int k = 0; k < 4; ++k) asm volatile("" : "+a"(qacc[r][s][k]));
            stamp();
            continue;
        }
        const float* pb = dyt + (size_t)g * a.DP + r16;
        // (A hand-pipelined form of this loop -- two operand register sets, the reads of iteration i + 1 issued in front of the MFMAs of
        // iteration i, table entries one iteration further ahead, pinned with sched_barriers: the conv kernel's recipe -- measured 5 - 8 %
        // SLOWER on every layer, profiles/r3_kbench_wgrad_pipelined_ab.txt: with two or three waves per SIMD the other waves already cover
        // the two LDS round trips of an iteration, and the second register set costs occupancy.  What bounds this loop is the number of
        // LDS instructions, one 4-byte read per MFMA; the remedy is K-grouped 16-byte operands, i.e. channel-major tiles.)
        auto ksteps = [&](int s, auto UC) __attribute__((always_inline)) {
            constexpr int U = decltype(UC)::value;
            int po[U];
            float av[U][MTW], bv[U][NTW];
#pragma unroll
            for (int u = 0; u < U; ++u) po[u] = pixoff[s + 4 * u + g];
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt) av[u][mt] = patch[po[u] + aoff[mt]];
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) bv[u][nt] = pb[(size_t)(s + 4 * u) * a.DP + nt * 16];
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][mt], bv[u][nt], acc[mt][nt], 0, 0, 0);
        };
        int s = 0;
        for (; s + 16 <= a.KP; s += 16) ksteps(s, std::integral_constant<int, 4>());
        for (; s < a.KP; s += 4) ksteps(s, std::integral_constant<int, 1>());
        stamp();
    }
    // partial tile out: rows (chunk, mblock, m), cols co
    const int mrows_chunk = a.mblocks_per_chunk * 64 * MTW;
    float* dst = a.partial + (int64_t)bx * a.Mrows_total * a.CoutP;
    if constexpr (RGW > 0) {
        // Sum over the sixteen blocks (lanes 4b + j, b = 0..15), in a fixed order.  The four registers e of an accumulator are four rows
        // of the gradient, and the wave has four DPP rows: two v_permlane16_swap + one v_permlane32_swap (gfx950: exchanges of whole
        // 16- / 32-lane groups between two registers, no LDS) add the rows' partial sums so that DPP row R is left with register R's --
        // a reduce-scatter, 3 exchanges + 3 adds per accumulator instead of 8 shuffles + 8 adds -- then two row shifts add the four
        // blocks of the row, and lanes 12..15 of every row store (16 lanes, one instruction per accumulator).
        const int qk4 = a.KC >> 2, units = a.ntaps * qk4;
        int rowoff[RG];   // slab offset of (row of unit 4 * rowgroup + DPP row, channel 0) + column j, -1: past the last unit / not a storing lane
#pragma unroll
        for (int r = 0; r < RG; ++r) {
            const int u = ((by * 4 + wave) * RG + r) * 4 + (lane >> 4);
            int c4;
            const int t = fdiv(min(u, units - 1), qk4, 1.0f / (float)qk4, c4);
            rowoff[r] = (u < units && (lane & 12) == 12) ? (t * a.KC + c4 * 4) * a.CoutP + (lane & 3) : -1;
        }
        auto sw16 = [](float x, float y) __attribute__((always_inline)) -> float {   // rows: [x0 + x1, y0 + y1, x2 + x3, y2 + y3]
            const auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, y), false, false);
            return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
        };
        auto sw32 = [](float x, float y) __attribute__((always_inline)) -> float {   // halves: [x.lo + x.hi, y.lo + y.hi]
            const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, y), false, false);
            return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
        };
#pragma unroll
        for (int r = 0; r < RG; ++r)
#pragma unroll
            for (int s = 0; s < kQBlocks; ++s)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const f32x4 v = qacc[r][s][k];
                    float x = sw32(sw16(v[0], v[1]), sw16(v[2], v[3]));   // DPP row R: register R summed over the four rows
                    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x114, 0xf, 0xf, true));   // row_shr:4
                    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x118, 0xf, 0xf, true));   // row_shr:8
                    if (rowoff[r] >= 0 && 4 * s + (lane & 3) < a.Cout) dst[rowoff[r] + k * a.CoutP + 4 * s] = x;
                }
        stamp();
        return;
    }
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int ml = m0 + wave * 16 * MTW + mt * 16 + g * 4 + reg;   // row inside the chunk: (tap, channel)
            const int row = chunk * mrows_chunk + ml;
            if (ml >= a.Mchunk) continue;   // padding rows / columns are never read by the reduction: not written either
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
                if (n0 + nt * 16 + r16 < a.Cout) dst[(int64_t)row * a.CoutP + n0 + nt * 16 + r16] = acc[mt][nt][reg];
        }
    stamp();
}

typedef void (*wgrad_fn_t)(const WgradArgs);
static wgrad_fn_t wgrad_fn(int M, int N, int PF, int tab) {
#define OCL_CASE(A, B)                                                                                          \
    if (M == A && N == B) {                                                                                     \
        if (PF == 4) return tab ? conv_wgrad_kernel<A, B, 4, 0, 1> : conv_wgrad_kernel<A, B, 4, 0, 0>;          \
        if (PF == 8) return tab ? conv_wgrad_kernel<A, B, 8, 0, 1> : conv_wgrad_kernel<A, B, 8, 0, 0>;          \
    }
    OCL_CASE(1, 1) OCL_CASE(1, 2) OCL_CASE(1, 3) OCL_CASE(1, 4) OCL_CASE(1, 5)
    OCL_CASE(2, 1) OCL_CASE(2, 2) OCL_CASE(2, 3) OCL_CASE(2, 4) OCL_CASE(2, 5)
    OCL_CASE(3, 1) OCL_CASE(3, 2) OCL_CASE(3, 3) OCL_CASE(3, 4) OCL_CASE(3, 5)
    OCL_CASE(4, 1) OCL_CASE(4, 2) OCL_CASE(4, 3) OCL_CASE(4, 4) OCL_CASE(4, 5)
#undef OCL_CASE
    return nullptr;
}
static wgrad_fn_t wgrad_q_fn(int rgw, int PF, int tab) {
#define OCL_CASE(R)                                                                                                     \
    if (rgw == R)                                                                                                       \
        return tab ? (PF == 4 ? conv_wgrad_kernel<1, 1, 4, R, 1> : conv_wgrad_kernel<1, 1, 8, R, 1>)                    \
                   : (PF == 4 ? conv_wgrad_kernel<1, 1, 4, R, 0> : conv_wgrad_kernel<1, 1, 8, R, 0>);
    OCL_CASE(1) OCL_CASE(2) OCL_CASE(3)
#undef OCL_CASE
    return nullptr;
}
// measurement builds (TRACE) of the forms the SCR pass runs most
static wgrad_fn_t wgrad_trace_fn(int M, int N, int PF, int rgw, int tab) {
    if (!tab || PF != 8) return nullptr;
    if (rgw == 3) return conv_wgrad_kernel<1, 1, 8, 3, 1, 1>;
    if (rgw) return nullptr;
    if (M == 2 && N == 3) return conv_wgrad_kernel<2, 3, 8, 0, 1, 1>;
    if (M == 3 && N == 2) return conv_wgrad_kernel<3, 2, 8, 0, 1, 1>;
    if (M == 1 && N == 3) return conv_wgrad_kernel<1, 3, 8, 0, 1, 1>;
    return nullptr;
}
static int wgrad_pf_for(int units) { return units <= 1024 ? 4 : 8; }

// sums the split-K partials into the OIHW gradient: grad[co][ci][t] (+)= sum_s partial[s][(chunk,t,cc)][co].
// 32 consecutive outputs (co fastest: coalesced partial reads) x 8 split lanes per block; the 8 lane sums are combined
// through LDS in a fixed order, so the result does not depend on scheduling.
__device__ __forceinline__ void wgrad_reduce_body(const float* __restrict__ partial, int S, int Mrows_total, int CoutP, int mrows_chunk,
                                                  int KC, int ntaps, int CinReal, int Cout, float* __restrict__ grad, int accumulate,
                                                  int block, float (*red)[33]) {
    const int o = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int idx = block * 32 + o;  // (t, ci, co) with co fastest
    const int total = ntaps * CinReal * Cout;
    const bool valid = idx < total;
    const int co = idx % Cout;
    const int r = idx / Cout;
    const int ci = r % CinReal, t = r / CinReal;
    const int chunk = ci / KC, cc = ci - chunk * KC;
    const int row = chunk * mrows_chunk + t * KC + cc;
    const int64_t stride = (int64_t)Mrows_total * CoutP;
    const float* p = partial + (int64_t)row * CoutP + co;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (valid) {
        int s = sl;
        for (; s + 24 < S; s += 32) {
            s0 += p[(int64_t)s * stride];
            s1 += p[(int64_t)(s + 8) * stride];
            s2 += p[(int64_t)(s + 16) * stride];
            s3 += p[(int64_t)(s + 24) * stride];
        }
        for (; s < S; s += 8) s0 += p[(int64_t)s * stride];
    }
    red[sl][o] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (sl == 0 && valid) {
        float v = ((red[0][o] + red[1][o]) + (red[2][o] + red[3][o])) + ((red[4][o] + red[5][o]) + (red[6][o] + red[7][o]));
        float* gp = grad + ((int64_t)co * CinReal + ci) * ntaps + t;
        if (accumulate) v += *gp;
        *gp = v;
    }
}

__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float* __restrict__ partial, int S, int Mrows_total, int CoutP,
                                                           int mrows_chunk, int KC, int ntaps, int CinReal, int Cout,
                                                           float* __restrict__ grad, int accumulate) {
    __shared__ float red[8][33];
    wgrad_reduce_body(partial, S, Mrows_total, CoutP, mrows_chunk, KC, ntaps, CinReal, Cout, grad, accumulate, blockIdx.x, red);
}

// the reductions of ALL layers of a backward pass in one launch (replay-sized batches run the whole backward on one stream and are
// bound by the number of dependent launches: 21 reductions -> 1); every layer keeps its own slab region until then
__global__ void __launch_bounds__(256) wgrad_reduce_multi_kernel(const WgradReduceMulti m) {
    __shared__ float red[8][33];
    int l = 0;
#pragma unroll 1
    while (l + 1 < m.n && (int)blockIdx.x >= m.L[l + 1].block0) ++l;
    const WgradReduceLayer& d = m.L[l];
    wgrad_reduce_body(m.partial + d.partial_off, d.S, d.Mrows_total, d.CoutP, d.mrows_chunk, d.KC, d.ntaps, d.CinReal, d.Cout,
                      m.grads + d.grad_off, m.accumulate, (int)blockIdx.x - d.block0, red);
}

// LDS pixel stride of the wgrad input patch.  A reads (ds_read_b32, 32 banks, lanes 0-31 = 2 pixels x 16 channels)
// are conflict-free when stride*CP = 16 (mod 32); take the smallest even CP >= KC within 4 banks of that.
static int wg_cp(int kc, int stride) {
    for (int cp = kc;; cp += 2) {
        const int x = (stride * cp) & 31;
        const int d = std::min(x, 32 - x);
        if (16 - d <= 4) return cp;
    }
}

int plan_wgrad(int N, int Hin, int Win, int Cin, int Ho, int Wo, int Cout, int ksize, int stride, WgradPlan* p, int xf_groups) {
    memset(p, 0, sizeof(*p));
    WgradArgs& a = p->a;
    OCL_REQUIRE(Cin % 4 == 0 && Cout % 4 == 0 && (ksize == 1 || ksize == 3), "plan_wgrad: Cin=%d Cout=%d k=%d", Cin, Cout, ksize);
    a.N = N; a.Hin = Hin; a.Win = Win; a.Cin = Cin; a.Ho = Ho; a.Wo = Wo; a.Cout = Cout;
    a.stride = stride;
    const int pad = ksize == 3 ? 1 : 0;
    a.ntaps = ksize * ksize;
    for (int t = 0; t < a.ntaps; ++t) {
        a.tdy[t] = t / ksize - pad;
        a.tdx[t] = t % ksize - pad;
    }
    a.min_dy = a.min_dx = -pad;
    a.max_dy = a.max_dx = ksize - 1 - pad;
    const int ntile = cdiv(Cout, 16);
    int NTW = ntile <= 3 ? ntile : 3;   // 48 output channels per workgroup (register budget of the dy prefetch)
    a.nblocks = cdiv(ntile, NTW);
    if (a.nblocks > 1) NTW = cdiv(ntile, a.nblocks);
    a.CoutP = a.nblocks * NTW * 16;
    int dp = NTW * 16;
    while ((dp & 31) != 16) dp += 16;
    a.DP = dp;
    const int LP = Ho * Wo;
    // pixel tile (KP output pixels, 128 / 64 / 32) and channel chunk KC: the largest tile whose patch + dy fit the LDS
    // target and the prefetch registers with a chunk of at least min(20, Cin) channels; else the best that fits at all.
    bool found = false;
    // measurement knobs (kbench sweeps): largest pixel tile, workgroup target of the pixel split, smallest grid that stops the search
    static const int env_kp = [] { const char* e = getenv("OCL_WGRAD_KP"); return e ? atoi(e) : 128; }();
    static const int env_target = [] { const char* e = getenv("OCL_WGRAD_TARGET"); return e ? atoi(e) : 512; }();
    static const int env_enough = [] { const char* e = getenv("OCL_WGRAD_ENOUGH"); return e ? atoi(e) : 384; }();
    for (int pass = 0; pass < 2 && !found; ++pass) {
        for (int KPmax = env_kp; KPmax >= 32 && !found; KPmax /= 2) {
            if (LP >= KPmax) {
                a.imgs = 1; a.ppi = KPmax; a.tiles_per_img = cdiv(LP, KPmax); a.KP = KPmax;
            } else {
                a.imgs = std::min(KPmax / LP, N); a.ppi = LP; a.tiles_per_img = 1; a.KP = (int)round_up((int64_t)a.imgs * LP, 4);
            }
            a.PC = (Wo - 1) * stride + (a.max_dx - a.min_dx) + 1;
            const int rows_l = (a.imgs == 1 && LP >= KPmax) ? std::min(Ho, (KPmax + Wo - 2) / Wo + 1) : Ho;
            a.PR = (rows_l - 1) * stride + (a.max_dy - a.min_dy) + 1;
            if (a.imgs > 127 || a.PR >= 256 || a.PC >= 256) continue;
            for (int KC = Cin; KC >= 4; KC -= 4) {
                if (Cin % KC) continue;
                if (pass == 0 && KC < std::min(20, Cin)) break;
                a.KC = KC; a.CP = wg_cp(KC, stride);
                const size_t bytes = 16 + (size_t)a.KP * 4 + (size_t)a.KP * a.DP * 4 + (size_t)a.imgs * a.PR * a.PC * a.CP * 4 +
                                     (size_t)xf_groups * Cin * 8 + (xf_groups ? 16 : 0);   // (dummy slot + ... + the input-transform table)
                const bool fits = (pass == 0 ? bytes <= kLdsTarget : bytes <= kLdsLimit - 1024) &&
                                  a.imgs * a.PR * a.PC * (KC / 4) <= 256 * kPatchPF;
                if (fits) { p->lds_bytes = bytes; found = true; break; }
            }
        }
    }
    if (!found) {
        set_error("plan_wgrad: no pixel tile fits the LDS (Hin=%d Win=%d Cin=%d Cout=%d)", Hin, Win, Cin, Cout);
        return OCL_ERR_ARG;
    }
    a.nchunks = Cin / a.KC;
    a.Mchunk = a.ntaps * a.KC;
    {   // patch prefetch walk: 256 units = d_row rows + d_pc pixels + d_c4 float4s
        const int kc4 = a.KC / 4;
        a.d_c4 = 256 % kc4;
        const int d_pix = 256 / kc4;
        a.d_pc = d_pix % a.PC;
        a.d_row = d_pix / a.PC;
        a.inv_PR = 1.0f / (float)a.PR;
    }
    const int mtiles = cdiv(a.Mchunk, 16);
    a.total_tiles = cdiv(N, a.imgs) * a.tiles_per_img;
    // Block tile (64*MTW rows) and pixel split S: aim at >= 384 workgroups (1.5 per CU) with the largest tile that
    // gets there, cap the split so that the fp32 partial slabs stay <= 12 MB (they are written and read once), and
    // balance the pixel tiles over the S slices.
    int MTW = 1, bestS = 1;
    int64_t best_blocks = -1;
    for (int m = std::min(4, cdiv(mtiles, 4)); m >= 1; --m) {
        if (m * NTW > 20) continue;
        const int mb = cdiv(mtiles, 4 * m);
        const int by = a.nchunks * mb * a.nblocks;
        const int64_t slab = (int64_t)a.nchunks * mb * 64 * m * a.CoutP * 4;
        const int s_cap = (int)std::max<int64_t>(1, (12ll << 20) / slab);
        int S = std::max(1, std::min(std::min(a.total_tiles, s_cap), cdiv(env_target, by)));
        const int tpb = cdiv(a.total_tiles, S);
        S = cdiv(a.total_tiles, tpb);
        const int64_t blocks = (int64_t)by * S;
        if (blocks > best_blocks) { best_blocks = blocks; MTW = m; bestS = S; }
        if (blocks >= env_enough) break;
    }
    a.mblocks_per_chunk = cdiv(mtiles, 4 * MTW);
    a.Mrows_total = a.nchunks * a.mblocks_per_chunk * 64 * MTW;
    const int by = a.nchunks * a.mblocks_per_chunk * a.nblocks;
    a.S = bestS;
    p->MTW = MTW; p->NTW = NTW;
    p->grid_x = a.S; p->grid_y = by;
    p->partial_floats = (size_t)a.S * a.Mrows_total * a.CoutP;
    // staging with precomputed unit tables (conv_wgrad_kernel, TAB): the default; OCL_WGRAD_TAB=0 selects the form that re-derives the
    // units per tile (bit-identical results: scripts/gpu_r4z3.sh, profiles/r4_wgrad_tab_ab.txt)
    static const int env_tab = [] { const char* e = getenv("OCL_WGRAD_TAB"); return e ? atoi(e) : 1; }();
    p->tab = env_tab ? 1 : 0;
    // EXPERIMENTAL (OCL_WGRAD_Q=1, default off: written at the end of round 4 without GPU time left to validate it): the 4x4x1 form for
    // <= 20 output channels and a single channel chunk -- stem and layer 1, the two largest pixel counts of the network.
    //   OCL_WGRAD_Q_RGW     row groups (16 gradient rows) per wave, 1..3 (default: the smallest count that covers the rows with one
    //                       row block, i.e. the patch is staged once per pixel tile)
    //   OCL_WGRAD_Q_TARGET  workgroups aimed at by the pixel split (default 256: the block sums of the epilogue cost about one pixel
    //                       tile's MFMAs, so fewer, longer workgroups than the 16x16x4 form)
    static const int env_q = [] { const char* e = getenv("OCL_WGRAD_Q"); return e ? atoi(e) : 0; }();
    // (the stem's 9 units fill 3 of 4 waves: measured slower; the block sums of the epilogue cost about 1.7 pixel tiles, and the form
    // runs one workgroup per CU: it pays from ~6 tiles of 128 pixels per workgroup at 256 workgroups -- SCR's 220 views yes (-33 us per
    // pass), 20 images of 84 x 84 no (+40 us); OCL_WGRAD_Q=2 lifts that limit)
    if (env_q && Cout <= 4 * kQBlocks && Cin >= 8 && a.nchunks == 1 && a.CP % 4 == 0 && a.KP % 16 == 0 &&
        ((int64_t)a.total_tiles * a.KP >= 6 * 128 * 256 || env_q >= 2)) {
        static const int env_rgw = [] { const char* e = getenv("OCL_WGRAD_Q_RGW"); return e ? atoi(e) : 0; }();
        static const int env_qtarget = [] { const char* e = getenv("OCL_WGRAD_Q_TARGET"); return e ? atoi(e) : 256; }();
        const int rg = cdiv(a.ntaps * (a.KC / 4), 4);
        int rgw = rg <= 4 ? 1 : rg <= 8 ? 2 : 3;
        if (env_rgw >= 1 && env_rgw <= 3) rgw = env_rgw;
        a.nblocks = 1;
        a.DP = a.CoutP = 4 * kQBlocks;
        const size_t bytes = 16 + (size_t)a.KP * 4 + (size_t)a.KP * a.DP * 4 + (((size_t)a.imgs * a.PR * a.PC * a.CP + 3) & ~(size_t)3) * 4 +
                             (size_t)xf_groups * Cin * 8 + (xf_groups ? 16 : 0);
        p->lds_bytes = bytes;
        a.mblocks_per_chunk = cdiv(a.Mchunk, 64);   // slab rows as the 16x16x4 form with MTW = 1 (the reduction reads this format)
        a.Mrows_total = a.mblocks_per_chunk * 64;
        const int qby = cdiv(rg, 4 * rgw);
        const int64_t slab = (int64_t)a.Mrows_total * a.CoutP * 4;
        const int s_cap = (int)std::max<int64_t>(1, (12ll << 20) / slab);
        int S = std::max(1, std::min(std::min(a.total_tiles, s_cap), cdiv(env_qtarget, qby)));
        S = cdiv(a.total_tiles, cdiv(a.total_tiles, S));
        a.S = S;
        p->MTW = 1; p->NTW = 1; p->q_rgw = rgw;
        p->grid_x = S; p->grid_y = qby;
        p->partial_floats = (size_t)S * a.Mrows_total * a.CoutP;
    }
    // XCD-aware order of the workgroups (conv_wgrad_kernel: bx / by); written at the end of round 4, not yet measured: default off
    static const int env_xcd = [] { const char* e = getenv("OCL_WGRAD_XCD"); return e ? atoi(e) : 0; }();
    a.xcd_by = (env_xcd && p->grid_y > 1 && a.S >= 8) ? p->grid_y : 0;
    return OCL_OK;
}

int launch_wgrad(const WgradPlan& p, hipStream_t s) {
    const int pf = wgrad_pf_for(p.a.imgs * p.a.PR * p.a.PC * (p.a.KC / 4));
    wgrad_fn_t fn = p.q_rgw ? wgrad_q_fn(p.q_rgw, pf, p.tab) : wgrad_fn(p.MTW, p.NTW, pf, p.tab);
    if (p.a.trace) {
        fn = wgrad_trace_fn(p.MTW, p.NTW, pf, p.q_rgw, p.tab);
        if (!fn) {
            set_error("launch_wgrad: no trace build for MTW=%d NTW=%d PF=%d rgw=%d tab=%d", p.MTW, p.NTW, pf, p.q_rgw, p.tab);
            return OCL_ERR_UNSUPPORTED;
        }
        OCL_HIP(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit));
    }
    if (!fn) {
        set_error("launch_wgrad: no kernel for MTW=%d NTW=%d", p.MTW, p.NTW);
        return OCL_ERR_STATE;
    }
    ProfScope ps(PROF_WGRAD, s);
    hipLaunchKernelGGL(fn, p.a.xcd_by > 0 ? dim3(p.grid_x * p.grid_y, 1) : dim3(p.grid_x, p.grid_y), dim3(256), p.lds_bytes, s, p.a);
    OCL_LAUNCH_CHECK();
    return OCL_OK;
}

int launch_wgrad_reduce(const WgradPlan& p, float* grad_oihw, int accumulate, hipStream_t s) {
    const WgradArgs& a = p.a;
    const int cin_real = a.Cin == 4 ? 3 : a.Cin;  // the stem's NHWC4 input carries a zero 4th channel
    const int total = a.ntaps * cin_real * a.Cout;
    ProfScope ps(PROF_WGRAD, s);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cdiv(total, 32)), dim3(256), 0, s, a.partial, a.S, a.Mrows_total, a.CoutP,
                       a.mblocks_per_chunk * 64 * p.MTW, a.KC, a.ntaps, cin_real, a.Cout, grad_oihw, accumulate);
    OCL_LAUNCH_CHECK();
    return OCL_OK;
}

void wgrad_reduce_layer(const WgradPlan& p, int64_t partial_off, int64_t grad_off, WgradReduceLayer* d) {
    const WgradArgs& a = p.a;
    d->partial_off = partial_off;
    d->grad_off = grad_off;
    d->S = a.S; d->Mrows_total = a.Mrows_total; d->CoutP = a.CoutP;
    d->mrows_chunk = a.mblocks_per_chunk * 64 * p.MTW;
    d->KC = a.KC; d->ntaps = a.ntaps;
    d->CinReal = a.Cin == 4 ? 3 : a.Cin;
    d->Cout = a.Cout;
    d->block0 = 0;
}

int launch_wgrad_reduce_multi(WgradReduceMulti m, hipStream_t s) {
    OCL_REQUIRE(m.n >= 1 && m.n <= kMaxReduceLayers, "wgrad_reduce_multi: %d layers", m.n);
    int blocks = 0;
    for (int i = 0; i < m.n; ++i) {
        m.L[i].block0 = blocks;
        blocks += cdiv(m.L[i].ntaps * m.L[i].CinReal * m.L[i].Cout, 32);
    }
    ProfScope ps(PROF_WGRAD, s);
    hipLaunchKernelGGL(wgrad_reduce_multi_kernel, dim3(blocks), dim3(256), 0, s, m);
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
    // the scattered stores are the cost of this kernel: a pass writes only the packs it reads (PACK_* bits)
    if (!(mask & PACK_TF)) d.tf_off = -1;
    if (!(mask & PACK_TD)) d.td_off = -1;
    const int total = d.Cout * d.Cin * d.ntaps;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int co = e / (d.Cin * d.ntaps);
        const int rem = e - co * d.Cin * d.ntaps;
        const int ci = rem / d.ntaps, t = rem - ci * d.ntaps;
        const float v = params[d.w_off + e];
        if (d.tf_off >= 0) arena[d.tf_off + ((((int64_t)t * (d.CinP >> 2) + (ci >> 2)) * d.CoutP + co) << 2) + (ci & 3)] = v;
        if (d.td_off >= 0) arena[d.td_off + ((((int64_t)t * (d.Cout >> 2) + (co >> 2)) * d.CiP + ci) << 2) + (co & 3)] = v;
    }
}

int launch_pack_weights(const float* params, float* arena, const PackDesc* descs_dev, int n_layers, int max_elems, hipStream_t s,
                        int mask, StatCell* zero_a, int64_t zero_a_n, StatCell* zero_b, int64_t zero_b_n) {
    ProfScope ps(PROF_BN, s);
    const int extra = (zero_a_n + zero_b_n) > 0 ? 1 : 0;
    // (up to 256 workgroups per layer: layer 4's 230 k weights in 4 passes per thread instead of 14)
    hipLaunchKernelGGL(pack_weights_kernel, dim3(std::min(256, cdiv(max_elems, 256)), n_layers + extra), dim3(256), 0, s, params, arena,
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
    __syncthreads();
    const int C4 = a.C >> 2;
    const int64_t units = a.m_per_group * C4;
    const float4* y4 = (const float4*)a.y + (int64_t)g * units;
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
    hipLaunchKernelGGL(bn_fwd_kernel, dim3(bx, a.G), dim3(256), (size_t)a.C * 8, s, a);
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
        const unsigned groups_total = gridDim.x < (unsigned)kBnFusedReps ? gridDim.x : (unsigned)kBnFusedReps;
        if (__hip_atomic_fetch_add(a.barrier + 1 + sub, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == members - 1)
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

static int g_bn_bwd_cap = 0, g_bn_bwd_unroll = 0, g_bn_bwd_phase = 0;   // micro-benchmark overrides (kbench)
void bn_bwd_tune(int cap, int unroll, int phase) { g_bn_bwd_cap = cap; g_bn_bwd_unroll = unroll; g_bn_bwd_phase = phase; }

static int g_bn_fused = -1;   // -1: environment (OCL_BN_FUSED, default on)
static int g_num_cus = 0;
void bn_bwd_fused_enable(int on) { g_bn_fused = on; }

int launch_bn_bwd(const BnBwdArgs& a, hipStream_t s) {
    OCL_REQUIRE(a.nsets == 1 || a.nsets == 2, "bn_bwd: nsets=%d", a.nsets);
    const int C4 = a.C / 4, PT = 256 / C4;
    if (g_bn_fused < 0) {
        const char* e = getenv("OCL_BN_FUSED");
        g_bn_fused = e ? atoi(e) : 1;
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
    static bool done = false;
    if (done) return OCL_OK;
    {
        const char* e = getenv("OCL_DETERMINISTIC");
        if (e && e[0] == '1') {
            int rc = set_deterministic_sums(1);
            if (rc != OCL_OK) return rc;
        }
    }
    for (int m = 1; m <= 4; ++m)
        for (int n = 1; n <= 5; ++n)
            for (int pf = 4; pf <= 8; pf += 4)
                for (int tab = 0; tab < 2; ++tab)
                    OCL_HIP(hipFuncSetAttribute((const void*)wgrad_fn(m, n, pf, tab), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit));
    for (int r = 1; r <= 3; ++r)
        for (int pf = 4; pf <= 8; pf += 4)
            for (int tab = 0; tab < 2; ++tab)
                OCL_HIP(hipFuncSetAttribute((const void*)wgrad_q_fn(r, pf, tab), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit));
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
