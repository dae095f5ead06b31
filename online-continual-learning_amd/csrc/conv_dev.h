// Device-side helpers and constants shared by the translation units of the convolution engine (conv.hip: forward / data gradient /
// BatchNorm kernels; wgrad.hip: weight gradient).  Header-only: everything here is inline / constexpr.
#pragma once
#include "conv.h"

namespace ocl {

static const size_t kLdsLimit = 160 * 1024;      // hardware: 160 KiB per workgroup
static const size_t kLdsTarget = 72 * 1024;      // weight-gradient planner target (2 workgroups per CU)
constexpr int kQBlocks = 5;   // blocks of four output channels (Cout <= 20)

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

}  // namespace ocl
