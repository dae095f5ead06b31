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
