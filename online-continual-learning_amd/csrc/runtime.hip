// Library runtime: error strings, device check, HIP-event profiling accumulators.
#include "common.h"
#include <string.h>
#include <vector>
#include <mutex>

namespace ocl {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int hip_fail(hipError_t e, const char* what, const char* file, int line) {
    set_error("HIP error %d (%s) in `%s` at %s:%d", (int)e, hipGetErrorString(e), what, file, line);
    return OCL_ERR_HIP;
}

// ---- asynchronous device-side errors ----------------------------------------------------------------
static unsigned* g_async_host = nullptr;
static unsigned* g_async_dev = nullptr;
static std::once_flag g_async_once;

unsigned* async_error_word_device() {
    std::call_once(g_async_once, [] {
        void* h = nullptr;
        if (hipHostMalloc(&h, 64, hipHostMallocMapped) != hipSuccess) { (void)hipGetLastError(); return; }
        memset(h, 0, 64);
        void* d = nullptr;
        if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) { (void)hipGetLastError(); (void)hipHostFree(h); return; }
        g_async_host = (unsigned*)h;
        g_async_dev = (unsigned*)d;
    });
    return g_async_dev;
}

int check_async_error(const char* where) {
    if (!g_async_host) return OCL_OK;
    const unsigned code = __atomic_load_n(g_async_host, __ATOMIC_RELAXED);
    if (code == 0) return OCL_OK;
    __atomic_store_n(g_async_host, 0u, __ATOMIC_RELAXED);
    if (code & ASYNC_ERR_BN_BARRIER)
        set_error("%s: an earlier one-pass BatchNorm backward could not gather all of its workgroups (grid-wide arrival timed out: the "
                  "GPU is shared with work that holds its compute units); the gradients of that backward are poisoned with NaN. "
                  "Set OCL_BN_FUSED=0 to use the two-kernel BatchNorm backward.", where);
    else
        set_error("%s: asynchronous device error code %u", where, code);
    return OCL_ERR_STATE;
}

// ---- profiling -------------------------------------------------------------------------------------
struct EvPair {
    hipEvent_t a, b;
};
static bool g_prof = false;
static std::mutex g_prof_mu;
static std::vector<EvPair> g_pairs[PROF_NCLS];
static std::vector<EvPair> g_free;
static EvPair g_open[PROF_NCLS];

bool prof_on() { return g_prof; }

void prof_begin(int cls, hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    EvPair p;
    if (!g_free.empty()) {
        p = g_free.back();
        g_free.pop_back();
    } else {
        if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) return;
    }
    (void)hipEventRecord(p.a, s);
    g_open[cls] = p;
}

void prof_end(int cls, hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    EvPair p = g_open[cls];
    (void)hipEventRecord(p.b, s);
    g_pairs[cls].push_back(p);
}

// ---- small host -> device uploads ---------------------------------------------------------------------
// Index vectors and parameter rows (<= 64 KB) go through a ring of pinned staging slots: memcpy into the slot, asynchronous copy
// on the caller's stream, an event behind it; a slot is reused only after its event has completed.  One ring per device.
static constexpr int kStageSlots = 128;
static constexpr size_t kStageBytes = 64 * 1024;
struct StageRing {
    char* base = nullptr;
    hipEvent_t ev[kStageSlots];
    bool used[kStageSlots];
    int next = 0;
};
static StageRing g_rings[16];
static std::mutex g_stage_mu;

int upload_small(const void* host, size_t nbytes, void* dev, hipStream_t s) {
    if (nbytes == 0) return OCL_OK;
    if (nbytes > kStageBytes) {   // large payloads: plain copy (blocks the host until staged)
        OCL_HIP(hipMemcpyAsync(dev, host, nbytes, hipMemcpyHostToDevice, s));
        return OCL_OK;
    }
    int d = 0;
    OCL_HIP(hipGetDevice(&d));
    OCL_REQUIRE(d >= 0 && d < 16, "ocl_upload: device index %d not supported", d);
    std::lock_guard<std::mutex> lk(g_stage_mu);
    StageRing& r = g_rings[d];
    if (!r.base) {
        void* h = nullptr;
        OCL_HIP(hipHostMalloc(&h, kStageSlots * kStageBytes, hipHostMallocDefault));
        for (int i = 0; i < kStageSlots; ++i) {
            if (hipEventCreateWithFlags(&r.ev[i], hipEventDisableTiming) != hipSuccess) {   // the ring exists only once it is complete
                for (int j = 0; j < i; ++j) (void)hipEventDestroy(r.ev[j]);
                (void)hipHostFree(h);
                set_error("ocl_upload: cannot create the staging ring's events");
                return OCL_ERR_HIP;
            }
            r.used[i] = false;
        }
        r.base = (char*)h;
    }
    const int k = r.next;
    r.next = (k + 1) % kStageSlots;
    if (r.used[k]) OCL_HIP(hipEventSynchronize(r.ev[k]));
    char* slot = r.base + (size_t)k * kStageBytes;
    memcpy(slot, host, nbytes);
    OCL_HIP(hipMemcpyAsync(dev, slot, nbytes, hipMemcpyHostToDevice, s));
    OCL_HIP(hipEventRecord(r.ev[k], s));
    r.used[k] = true;
    return OCL_OK;
}

// ---- calibration: the fp32 MFMA rate of this box, registers only (see ocl_mfma_calibrate) -------------------------------
// __launch_bounds__(256, 2) keeps the accumulators in ArchVGPRs (no AccVGPR copies on the loop's back edge).
__global__ void __launch_bounds__(256, 2) mfma_calibrate_kernel(float* out, int iters) {
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float a = (float)(threadIdx.x & 3) * 0.25f, b = 1.0f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

}  // namespace ocl

using namespace ocl;

extern "C" {

int ocl_version(void) { return 100; }

const char* ocl_last_error(void) { return g_err; }

int ocl_init(int device) {
    int n = 0;
    OCL_HIP(hipGetDeviceCount(&n));
    OCL_REQUIRE(device >= 0 && device < n, "ocl_init: device %d out of range (have %d)", device, n);
    OCL_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    OCL_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("ocl_init: device %d is %s; this library is built for gfx950 (MI355X) only", device,
                  prop.gcnArchName);
        return OCL_ERR_UNSUPPORTED;
    }
    return OCL_OK;
}

int ocl_upload(const void* host, int64_t nbytes, void* dev, void* stream) {
    OCL_REQUIRE(nbytes >= 0 && (nbytes == 0 || (host && dev)), "ocl_upload: null pointer");
    return upload_small(host, (size_t)nbytes, dev, (hipStream_t)stream);
}

int ocl_mfma_calibrate(int iters, float* scratch, double* tflops, double* us, void* stream) {
    OCL_REQUIRE(iters > 0 && scratch, "mfma_calibrate: iters > 0 and a scratch buffer of 256 * n_cu floats");
    hipStream_t s = (hipStream_t)stream;
    int dev = 0, n_cu = 0;
    OCL_HIP(hipGetDevice(&dev));
    OCL_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    OCL_REQUIRE(n_cu > 0 && n_cu <= 1024, "mfma_calibrate: %d compute units", n_cu);
    hipEvent_t e0, e1;
    OCL_HIP(hipEventCreate(&e0));
    OCL_HIP(hipEventCreate(&e1));
    hipLaunchKernelGGL(mfma_calibrate_kernel, dim3(n_cu), dim3(256), 0, s, scratch, 16);   // (code object load, clocks)
    OCL_HIP(hipEventRecord(e0, s));
    hipLaunchKernelGGL(mfma_calibrate_kernel, dim3(n_cu), dim3(256), 0, s, scratch, iters);
    OCL_HIP(hipEventRecord(e1, s));
    OCL_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    OCL_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    const double flops = (double)n_cu * 4 * iters * 4 * 2.0 * 16 * 16 * 4;   // 4 waves x iters x 4 MFMAs x 2 * 16 * 16 * 4
    if (us) *us = ms * 1e3;
    if (tflops) *tflops = ms > 0.f ? flops / (ms * 1e-3) / 1e12 : 0.0;
    return OCL_OK;
}

int ocl_prof_enable(int on) {
    g_prof = on != 0;
    return OCL_OK;
}

int ocl_prof_reset(void) {
    OCL_HIP(hipDeviceSynchronize());
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (int c = 0; c < PROF_NCLS; ++c) {
        for (auto& p : g_pairs[c]) g_free.push_back(p);
        g_pairs[c].clear();
    }
    return OCL_OK;
}

int ocl_prof_query(int cls, double* ms, int64_t* launches) {
    OCL_REQUIRE(cls >= 0 && cls < PROF_NCLS, "ocl_prof_query: bad class %d", cls);
    OCL_HIP(hipDeviceSynchronize());
    std::lock_guard<std::mutex> lk(g_prof_mu);
    double t = 0.0;
    for (auto& p : g_pairs[cls]) {
        float e = 0.f;
        if (hipEventElapsedTime(&e, p.a, p.b) == hipSuccess) t += e;
    }
    if (ms) *ms = t;
    if (launches) *launches = (int64_t)g_pairs[cls].size();
    return OCL_OK;
}

}  // extern "C"
