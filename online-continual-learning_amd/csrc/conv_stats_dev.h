// Device-side batch-sum / BatchNorm helpers shared by the translation units that accumulate or read BatchNorm statistics (conv.hip:
// conv_t / conv_q / conv_s kernels and the BatchNorm kernels; convw.hip: conv_w_kernel).  Header-only.  Without relocatable device code every
// translation unit that includes this file owns a copy of the __constant__ mode flag g_det_sums: set_deterministic_sums (conv.hip) sets its own
// copy and calls the other units' setters (convw_set_det).
#pragma once
#include "conv_dev.h"

namespace ocl {

// ---- batch sums (StatCell, conv.h) ---------------------------------------------------------------------------------------------
// Two ways to accumulate a cell, chosen at run time (ocl_set_deterministic / OCL_DETERMINISTIC=1, a __constant__ flag):
//  * default: the cell's first word holds a double and takes fp64 atomics (rounds 1 - 3): totals depend on the workgroups' arrival
//    order in the last bit;
//  * deterministic: 2^-40 fixed point added as two 64-bit INTEGERS (associative): bit-identical totals whatever the order.  Costs
//    two atomics per partial sum instead of one: +12 % on the SCR step, +13 % on ER (profiles/r4_batch_sums_ab.txt) -- which is why
//    it is a mode and not the default.
static __constant__ int g_det_sums = 0;
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

// x / d for a plan constant d through its precomputed M = ceil(2^32 / d): exact for x * d < 2^32 (checked by the planner)
__device__ __forceinline__ int mdiv(int x, unsigned M, int d, int& rem) {
    const int q = d == 1 ? x : (int)__umulhi((unsigned)x, M);
    rem = x - q * d;
    return q;
}

// sum over the 16 lanes of a DPP row (lanes 16k .. 16k+15), result in every lane: four v_add_f32 with DPP operands, no LDS traffic
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));   // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));   // row_mirror
    return v;
}

}  // namespace ocl
