// Shared host/device helpers for libocl_hip.so (gfx950 only; no compatibility paths).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/ocl_hip.h"

namespace ocl {

// ---- error plumbing ------------------------------------------------------------------------------
void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what, const char* file, int line);

#define OCL_HIP(call)                                                              \
    do {                                                                           \
        hipError_t _e = (call);                                                    \
        if (_e != hipSuccess) return ocl::hip_fail(_e, #call, __FILE__, __LINE__); \
    } while (0)

#define OCL_REQUIRE(cond, ...)          \
    do {                                \
        if (!(cond)) {                  \
            ocl::set_error(__VA_ARGS__); \
            return OCL_ERR_ARG;         \
        }                               \
    } while (0)

// launch check: catches bad configurations immediately (cheap, no sync)
#define OCL_LAUNCH_CHECK()                                                          \
    do {                                                                            \
        hipError_t _e = hipGetLastError();                                          \
        if (_e != hipSuccess) return ocl::hip_fail(_e, "kernel launch", __FILE__, __LINE__); \
    } while (0)

// ---- per-class HIP-event profiling (ocl_prof_*) ---------------------------------------------------
enum ProfClass { PROF_CONV = 0, PROF_WGRAD = 1, PROF_BN = 2, PROF_HEAD = 3, PROF_KNN = 4, PROF_NCLS = 5 };
bool prof_on();
void prof_begin(int cls, hipStream_t s);
void prof_end(int cls, hipStream_t s);

struct ProfScope {
    int cls;
    hipStream_t s;
    bool on;
    ProfScope(int c, hipStream_t st) : cls(c), s(st), on(prof_on()) {
        if (on) prof_begin(cls, s);
    }
    ~ProfScope() {
        if (on) prof_end(cls, s);
    }
};

// ---- asynchronous device-side errors ------------------------------------------------------------------
// One host-mapped word per process (hipHostMalloc, mapped): a kernel that cannot complete correctly (the one-pass BatchNorm
// backward whose grid-wide arrival timed out) stores a non-zero code there; the next engine entry point reads the word from the
// host WITHOUT synchronising and fails with OCL_ERR_STATE -- the reporting model of a sticky asynchronous error.
enum AsyncErr : unsigned { ASYNC_ERR_BN_BARRIER = 1u };
unsigned* async_error_word_device();        // device-visible address (nullptr if the allocation failed)
int check_async_error(const char* where);   // OCL_OK, or OCL_ERR_STATE with the message set and the word cleared
// small host -> device upload through the pinned staging ring (runtime.hip): asynchronous on `s`, `host` reusable on return
int upload_small(const void* host, size_t nbytes, void* dev, hipStream_t s);

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
static inline int64_t round_up(int64_t a, int64_t b) { return (a + b - 1) / b * b; }

}  // namespace ocl

// ---- device helpers ------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Block-wide sum for blockDim.x <= 1024 (multiple of 64); `red` is >= 16 floats of LDS.
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) red[wid] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}
__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) red[wid] = v;
    __syncthreads();
    float t = red[0];
    for (int i = 1; i < nw; ++i) t = fmaxf(t, red[i]);
    return t;
}
