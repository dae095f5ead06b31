// netcheck: a consumer of the C-ABI and nothing else (include/ocl_hip.h + libocl_hip.so; no torch, starts in well under a second).
// One training forward + backward of the engine on seeded inputs; the flat gradient, the outputs and the BatchNorm running statistics
// are written to a file, or compared with a file written by an earlier run -- the A/B of two run-time modes of the library
// (OCL_WGRAD_Q, OCL_BNB_EPI, OCL_GRAPH, OCL_DETERMINISTIC, ...) through the whole pass, tensor by tensor:
//   netcheck <n> <groups> <hw> <head> write  ref.bin
//   OCL_WGRAD_Q=1 netcheck <n> <groups> <hw> <head> compare ref.bin      -> per-tensor max |a - b| / max |b|, exit 1 above 1e-4
// It is a measurement / validation tool like kbench; the product path is the Python host side.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/ocl_hip.h"

#define OK(x)                                                                           \
    do {                                                                                \
        int rc_ = (x);                                                                  \
        if (rc_ != 0) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, ocl_last_error()); return 2; } \
    } while (0)
#define CK(x)                                                                           \
    do {                                                                                \
        hipError_t e_ = (x);                                                            \
        if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } \
    } while (0)

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static float urand() {   // splitmix64 -> U[0,1)
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (float)((z >> 40) * (1.0 / 16777216.0));
}

int main(int argc, char** argv) {
    if (argc < 7) {
        fprintf(stderr, "usage: netcheck <n> <groups> <hw> <head 0..3> write|compare <file>\n");
        return 2;
    }
    const int n = atoi(argv[1]), groups = atoi(argv[2]), hw = atoi(argv[3]), head = atoi(argv[4]);
    const bool write = !strcmp(argv[5], "write");
    const char* path = argv[6];
    OK(ocl_init(0));
    ocl_net_desc d;
    memset(&d, 0, sizeof(d));
    d.in_h = d.in_w = hw; d.nf = 20; d.n_classes = 100; d.head = head; d.feat_dim = 128; d.max_batch = n; d.n_slots = 1;
    ocl_net* net = nullptr;
    OK(ocl_net_create(&d, &net));
    const int64_t np = ocl_net_param_count(net), nr = ocl_net_bn_stat_count(net), wsb = ocl_net_workspace_bytes(net);
    const int nbn = ocl_net_num_bn(net), od = ocl_net_out_dim(net), nt = ocl_net_num_tensors(net);
    // parameters: convolutions / linears U(-a, a) with a = sqrt(3 / fan_in); one-dimensional tensors named *.weight around 1, the others around 0
    std::vector<float> hp(np);
    std::vector<std::string> names(nt);
    std::vector<int64_t> offs(nt + 1, np);
    for (int i = 0; i < nt; ++i) {
        char nm[64];
        int64_t off, shape[4];
        int32_t nd;
        OK(ocl_net_tensor_info(net, i, nm, &off, &nd, shape));
        names[i] = nm;
        offs[i] = off;
        int64_t cnt = 1, fan = 1;
        for (int k = 0; k < nd; ++k) cnt *= shape[k];
        for (int k = 1; k < nd; ++k) fan *= shape[k];
        const bool is_w = names[i].size() >= 6 && names[i].compare(names[i].size() - 6, 6, "weight") == 0;
        const float a = nd > 1 ? sqrtf(3.0f / (float)fan) : 0.1f;
        for (int64_t j = 0; j < cnt; ++j) hp[off + j] = (nd == 1 && is_w ? 1.0f : 0.0f) + a * (2.0f * urand() - 1.0f);
    }
    std::vector<float> hx((size_t)n * 3 * hw * hw), hd((size_t)n * od), hr(nr);
    for (auto& v : hx) v = 2.0f * urand() - 1.0f;
    for (auto& v : hd) v = (2.0f * urand() - 1.0f) / (float)n;
    for (int i = 0; i < nbn; ++i) {
        char nm[64];
        int64_t off;
        int32_t c;
        OK(ocl_net_bn_info(net, i, nm, &off, &c));
        for (int j = 0; j < c; ++j) { hr[off + j] = 0.f; hr[off + c + j] = 1.f; }
    }
    float *p, *g, *r, *x, *dout, *out, *feat;
    int64_t* nbt;
    void* ws;
    const int fd = ocl_net_feature_dim(net);
    CK(hipMalloc(&p, np * 4)); CK(hipMalloc(&g, np * 4)); CK(hipMalloc(&r, nr * 4)); CK(hipMalloc(&nbt, nbn * 8));
    CK(hipMalloc(&ws, wsb)); CK(hipMalloc(&x, hx.size() * 4)); CK(hipMalloc(&dout, hd.size() * 4)); CK(hipMalloc(&out, hd.size() * 4));
    CK(hipMalloc(&feat, (size_t)n * fd * 4));
    CK(hipMemcpy(p, hp.data(), np * 4, hipMemcpyHostToDevice));
    CK(hipMemset(g, 0xff, np * 4));   // NaN: every gradient entry has to be written by the pass
    CK(hipMemcpy(r, hr.data(), nr * 4, hipMemcpyHostToDevice));
    CK(hipMemset(nbt, 0, nbn * 8));
    CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dout, hd.data(), hd.size() * 4, hipMemcpyHostToDevice));
    OK(ocl_net_bind(net, p, g, r, nbt, ws, wsb));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    const int passes = getenv("NETCHECK_PASSES") ? atoi(getenv("NETCHECK_PASSES")) : 3;   // (the third pass runs replayed with OCL_GRAPH=1)
    for (int it = 0; it < passes; ++it) {
        OK(ocl_net_forward(net, x, n, groups, OCL_FWD_TRAIN | OCL_FWD_SAVE_TAPE | OCL_FWD_UPDATE_RUNNING, nullptr, feat, out, 0, s));
        OK(ocl_net_backward(net, 0, dout, 0, s));
    }
    CK(hipStreamSynchronize(s));
    // timing of the same pair (events on the stream of the launches)
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 20;
    CK(hipEventRecord(e0, s));
    for (int it = 0; it < reps; ++it) {
        OK(ocl_net_forward(net, x, n, groups, OCL_FWD_TRAIN | OCL_FWD_SAVE_TAPE, nullptr, feat, out, 0, s));
        OK(ocl_net_backward(net, 0, dout, 0, s));
    }
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<float> res((size_t)np + hd.size() + nr);
    CK(hipMemcpy(res.data(), g, np * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(res.data() + np, out, hd.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(res.data() + np + hd.size(), r, nr * 4, hipMemcpyDeviceToHost));
    printf("netcheck n=%d groups=%d hw=%d head=%d: forward + backward %.1f us per pass (%d passes)\n", n, groups, hw, head, ms * 1e3 / reps, reps);
    if (write) {
        FILE* f = fopen(path, "wb");
        if (!f || fwrite(res.data(), 4, res.size(), f) != res.size()) { fprintf(stderr, "cannot write %s\n", path); return 2; }
        fclose(f);
        printf("wrote %zu floats to %s\n", res.size(), path);
        return 0;
    }
    std::vector<float> ref(res.size());
    FILE* f = fopen(path, "rb");
    if (!f || fread(ref.data(), 4, ref.size(), f) != ref.size()) { fprintf(stderr, "cannot read %zu floats from %s\n", ref.size(), path); return 2; }
    fclose(f);
    int bad = 0, differ = 0;
    auto cmp = [&](const char* nm, size_t a, size_t b) {
        double dmax = 0.0, rmax = 0.0;
        bool nan = false, same = true;
        for (size_t i = a; i < b; ++i) {
            if (std::isnan(res[i]) || std::isnan(ref[i])) nan = true;
            if (memcmp(&res[i], &ref[i], 4)) same = false;
            dmax = fmax(dmax, fabs((double)res[i] - (double)ref[i]));
            rmax = fmax(rmax, fabs((double)ref[i]));
        }
        const double rel = dmax / (rmax + 1e-30);
        if (!same || nan) {
            ++differ;
            printf("  %-40s %8zu floats  reldiff %.2e%s\n", nm, b - a, rel, nan ? "  NaN" : rel > 1e-4 ? "  <-- MISMATCH" : "");
        }
        if (nan || rel > 1e-4) ++bad;
    };
    for (int i = 0; i < nt; ++i) {
        size_t end = np;
        for (int k = 0; k < nt; ++k)
            if (offs[k] > offs[i] && (size_t)offs[k] < end) end = offs[k];
        cmp(names[i].c_str(), offs[i], end);
    }
    cmp("(output)", np, np + hd.size());
    cmp("(running statistics)", np + hd.size(), res.size());
    printf("%d of %d tensors differ in some bit, %d beyond 1e-4 of the tensor's largest entry\n", differ, nt + 2, bad);
    return bad ? 1 : 0;
}
