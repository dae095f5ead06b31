// Kernel-level C-ABI entry points (single layers), used by the per-kernel parity tests and micro-benchmarks.
#include "conv.h"
#include <string.h>

using namespace ocl;

extern "C" {

int ocl_bn_bwd_nhwc(const float* dz, const float* zmask, const float* y, const float* mean, const float* invstd, const float* gamma,
                    int64_t m_per_group, int groups, int c, float* dy, float* dgamma, float* dbeta, int accumulate, double* scratch,
                    void* stream) {
    OCL_REQUIRE(dz && y && mean && invstd && gamma && dy && dgamma && dbeta && scratch, "bn_bwd: null pointer");
    OCL_REQUIRE(m_per_group > 0 && groups > 0 && c > 0 && c % 4 == 0 && c <= 1024, "bn_bwd: bad sizes");
    hipStream_t s = (hipStream_t)stream;
    OCL_HIP(hipMemsetAsync(scratch, 0, (size_t)groups * 2 * c * sizeof(StatCell), s));   // (one 16-byte accumulator cell per sum)
    BnBwdArgs a;
    memset(&a, 0, sizeof(a));
    a.dz = dz; a.z = zmask; a.m_per_group = m_per_group; a.G = groups; a.C = c; a.nsets = 1;
    a.y[0] = y; a.mean[0] = mean; a.invstd[0] = invstd; a.gamma[0] = gamma; a.dy[0] = dy; a.dgamma[0] = dgamma; a.dbeta[0] = dbeta;
    a.sums = (StatCell*)scratch;
    a.accumulate = accumulate;
    return launch_bn_bwd(a, s);
}

int ocl_set_deterministic(int on) { return set_deterministic_sums(on); }

}  // extern "C"
