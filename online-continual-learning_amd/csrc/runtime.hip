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
