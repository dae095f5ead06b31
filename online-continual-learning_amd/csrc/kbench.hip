// kbench — per-layer micro-benchmark of the Reduced-ResNet18 kernels (measurement tool, not part of libocl_hip.so).
//   kbench [n_images=220] [groups=2] [hw=32] [mode=all|conv|wgrad|bn] [sweep=1]
// Times every convolution layer's forward / data-gradient geometry for the planner's tile choice and (sweep=1) for
// every admissible forced (MT,NT), the weight-gradient kernels and the BatchNorm kernels, with HIP events.
// Each forced configuration is also compared against the planner's output (max |diff|): all tilings must agree.
#include "conv.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <time.h>
#include <string>
#include <vector>

using namespace ocl;

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(2);                                                                       \
        }                                                                                  \
    } while (0)
#define OK(x)                                                                  \
    do {                                                                       \
        int r_ = (x);                                                          \
        if (r_ != OCL_OK) {                                                    \
            fprintf(stderr, "ocl error %d: %s (%s:%d)\n", r_, ocl_last_error(), __FILE__, __LINE__); \
            exit(3);                                                           \
        }                                                                      \
    } while (0)

struct Layer {
    std::string name;
    ConvShape s;
};

static std::vector<Layer> make_layers(int hw, int nf) {
    std::vector<Layer> v;
    auto add = [&](const std::string& name, int Cin, int Cout, int k, int stride, int H, int W) {
        Layer l;
        l.name = name;
        memset(&l.s, 0, sizeof(l.s));
        l.s.Cin = Cin; l.s.CinT = Cin == 3 ? 4 : Cin; l.s.Cout = Cout; l.s.k = k; l.s.stride = stride;
        l.s.Hin = H; l.s.Win = W;
        const int pad = k == 3 ? 1 : 0;
        l.s.Ho = (H + 2 * pad - k) / stride + 1;
        l.s.Wo = (W + 2 * pad - k) / stride + 1;
        l.s.CoutP = pack_width(Cout);
        l.s.CiP = Cin == 3 ? 0 : pack_width(Cin);
        v.push_back(l);
        return l.s;
    };
    int H = hw, W = hw;
    add("conv1", 3, nf, 3, 1, H, W);
    int in_planes = nf;
    for (int layer = 0; layer < 4; ++layer) {
        const int planes = nf << layer;
        for (int b = 0; b < 2; ++b) {
            const int stride = (b == 0 && layer > 0) ? 2 : 1;
            char nm[64];
            snprintf(nm, sizeof(nm), "layer%d.%d", layer + 1, b);
            ConvShape c1 = add(std::string(nm) + ".conv1", in_planes, planes, 3, stride, H, W);
            add(std::string(nm) + ".conv2", planes, planes, 3, 1, c1.Ho, c1.Wo);
            if (stride != 1 || in_planes != planes) add(std::string(nm) + ".shortcut", in_planes, planes, 1, stride, H, W);
            in_planes = planes;
            H = c1.Ho;
            W = c1.Wo;
        }
    }
    return v;
}

static float* dev_rand(size_t n, unsigned seed, float scale = 1.f) {
    std::vector<float> h(n);
    unsigned s = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < n; ++i) {
        s = s * 1664525u + 1013904223u;
        h[i] = ((float)((s >> 8) & 0xffff) / 65535.0f - 0.5f) * scale;
    }
    float* d;
    CK(hipMalloc(&d, n * 4));
    CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
    return d;
}

template <class F>
static double time_us(F&& fn, int iters = 10, int warm = 2) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int i = 0; i < warm; ++i) fn();
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < iters; ++i) fn();
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, a, b));
    CK(hipEventDestroy(a));
    CK(hipEventDestroy(b));
    return (double)ms * 1e3 / iters;
}

static double max_diff(const float* a, const float* b, size_t n) {
    std::vector<float> ha(n), hb(n);
    CK(hipMemcpy(ha.data(), a, n * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hb.data(), b, n * 4, hipMemcpyDeviceToHost));
    double m = 0.0;
    for (size_t i = 0; i < n; ++i) m = fmax(m, fabs((double)ha[i] - (double)hb[i]));
    return m;
}

// K-grouped pack [tap][Cin/4][WP][4] of the same weights as the row pack [tap][Cin][WP] (9 taps)
static float* make_packT(const float* w_dev, int Cin, int WP) {
    const size_t n = (size_t)9 * Cin * WP;
    std::vector<float> h(n), t(n);
    CK(hipMemcpy(h.data(), w_dev, n * 4, hipMemcpyDeviceToHost));
    for (int tap = 0; tap < 9; ++tap)
        for (int ci = 0; ci < Cin; ++ci)
            for (int co = 0; co < WP; ++co)
                t[((((size_t)tap * (Cin / 4) + ci / 4) * WP + co) << 2) + (ci & 3)] = h[((size_t)tap * Cin + ci) * WP + co];
    float* d;
    CK(hipMalloc(&d, n * 4));
    CK(hipMemcpy(d, t.data(), n * 4, hipMemcpyHostToDevice));
    return d;
}

// Reference result for the conv kernel: one thread per output element, plain fp32 FMAs over (tap, input channel) of the row pack
// w[tap][Cin][WP] (independent of conv_t_kernel's tiling, operand layout and epilogue).
__global__ void conv_ref_kernel(ConvGeomDesc g, const float* __restrict__ in, const float* __restrict__ w, int WP, float* __restrict__ out) {
    const int64_t total = (int64_t)g.N * g.LH * g.LW * g.Cout;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int co = (int)(i % g.Cout);
        int64_t r = i / g.Cout;
        const int lx = (int)(r % g.LW); r /= g.LW;
        const int ly = (int)(r % g.LH);
        const int n = (int)(r / g.LH);
        float acc = 0.f;
        for (int t = 0; t < g.ntaps; ++t) {
            const int iy = ly * g.is + g.tdy[t], ix = lx * g.is + g.tdx[t];
            if (iy < 0 || iy >= g.Hin || ix < 0 || ix >= g.Win) continue;
            const float* xp = in + (((int64_t)n * g.Hin + iy) * g.Win + ix) * g.Cin;
            const float* wp = w + (int64_t)g.tw[t] * g.Cin * WP + co;
            for (int ci = 0; ci < g.Cin; ++ci) acc = fmaf(xp[ci], wp[(int64_t)ci * WP], acc);
        }
        out[(((int64_t)n * g.Hout + ly * g.os + g.oy0) * g.Wout + lx * g.os + g.ox0) * g.Cout + co] = acc;
    }
}

// Reference result for the weight-gradient kernels: one thread per (co, ci, tap), fp64 sum over every output pixel of
// x[n][oy*stride + dy][ox*stride + dx][ci] * dy[n][oy][ox][co] (independent of the split-K tiling and of the slab reduction).
__global__ void wgrad_ref_kernel(const float* __restrict__ x, const float* __restrict__ dy, int N, int Hin, int Win, int CinT, int CinReal, int Ho,
                                 int Wo, int Cout, int k, int stride, float* __restrict__ grad) {
    const int total = Cout * CinReal * k * k;
    const int pad = k == 3 ? 1 : 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int t = i % (k * k);
        const int ci = (i / (k * k)) % CinReal, co = i / (k * k * CinReal);
        const int ddy = t / k - pad, ddx = t % k - pad;
        double acc = 0.0;
        for (int n = 0; n < N; ++n)
            for (int oy = 0; oy < Ho; ++oy) {
                const int iy = oy * stride + ddy;
                if (iy < 0 || iy >= Hin) continue;
                for (int ox = 0; ox < Wo; ++ox) {
                    const int ix = ox * stride + ddx;
                    if (ix < 0 || ix >= Win) continue;
                    acc += (double)x[(((int64_t)n * Hin + iy) * Win + ix) * CinT + ci] * (double)dy[(((int64_t)n * Ho + oy) * Wo + ox) * Cout + co];
                }
            }
        grad[i] = (float)acc;   // OIHW: (co * CinReal + ci) * k*k + t
    }
}

static void bench_geom(const char* lname, const char* kind, ConvGeomDesc g, const float* in, const float* w, float* out,
                       float* out_ref, StatCell* stats, int flags, double flops, size_t out_elems, bool sweep) {
    float* wT = make_packT(w, g.Cin, g.WPT);
    auto run = [&](ConvPlan p, float* o) {
        p.a.in = in; p.a.wT = wT; p.a.out = o; p.a.flags = flags; p.a.stats = stats; p.a.stat_rep_stride = 8 * 2 * 1024;
        OK(launch_conv(p, 0));
    };
    CK(hipMemset(out_ref, 0, out_elems * 4));
    hipLaunchKernelGGL(conv_ref_kernel, dim3(2048), dim3(256), 0, 0, g, in, w, g.WPT, out_ref);
    CK(hipDeviceSynchronize());
    // reference statistics (per BatchNorm group and channel: sum, sum of squares) from the reference output, in fp64 on the host
    std::vector<double> ref_stats((size_t)g.groups * 2 * g.Cout, 0.0);
    if (flags & EPI_STATS) {
        std::vector<float> h(out_elems);
        CK(hipMemcpy(h.data(), out_ref, out_elems * 4, hipMemcpyDeviceToHost));
        const int64_t per_img = (int64_t)g.Hout * g.Wout * g.Cout;
        const int gsz = g.N / g.groups;
        for (int n = 0; n < g.N; ++n)
            for (int64_t e = 0; e < per_img; ++e) {
                const double v = h[(size_t)n * per_img + e];
                const int c = (int)(e % g.Cout), gg = n / gsz;
                ref_stats[((size_t)gg * 2 + 0) * g.Cout + c] += v;
                ref_stats[((size_t)gg * 2 + 1) * g.Cout + c] += v * v;
            }
    }
    printf("%-20s %-7s M=%7d N=%3d K=%4d\n", lname, kind, g.N * g.LH * g.LW, g.Cout, g.ntaps * g.Cin);
    {   // conv_t_kernel: planner's choice and (sweep) every admissible (MT, NT)
        StatCell* stats2;
        CK(hipMalloc(&stats2, kStatReps * 8 * 2 * 1024 * sizeof(StatCell)));
        for (int mt = 0; mt <= (sweep ? 5 : 0); ++mt)
            for (int nt = (mt ? 1 : 0); nt <= (mt ? 2 : 0); ++nt)
            for (int pipe = 0; pipe < 4; ++pipe) {   // staged-weight plans: the two-buffer schedule, then the three-buffer ring; 2: conv_t_kernel where the planner takes conv_q_kernel; 3: conv_w_kernel
                ConvGeomDesc gt = g;
                gt.force_MT = mt; gt.force_NT = nt;
                gt.force_cw = pipe == 3 ? 1 : -1;
                if (pipe == 3 && getenv("KBENCH_NO_CW")) continue;
                gt.force_pipe = pipe == 1 ? 1 : pipe == 2 ? 0 : -1;   // (the comparison line runs conv_t_kernel's DEFAULT plan, ring included)
                gt.force_q4 = pipe == 2 ? -1 : 0;
                gt.force_cs = pipe == 2 ? -1 : 0;
                ConvPlan pt;
                if (plan_conv(gt, &pt) != OCL_OK) continue;
                if (pipe == 1 && !pt.a.pipe) continue;
                if (pipe == 3 && !pt.cw) continue;
                if (pipe == 2) {   // only when the default plan is the 4x4x1 form
                    ConvGeomDesc g0 = gt;
                    g0.force_q4 = 0;
                    ConvPlan p0;
                    g0.force_cs = 0;
                    if (mt || plan_conv(g0, &p0) != OCL_OK || !(p0.q4 || p0.cs)) continue;
                }
                OK(conv_plan_finalize(&pt));   // (kbench leaks the plans' device tables: a measurement tool that exits right after)
                CK(hipMemset(out, 0, out_elems * 4));
                CK(hipMemset(stats2, 0, kStatReps * 8 * 2 * 1024 * sizeof(StatCell)));
                StatCell* keep = stats;
                stats = stats2;
                run(pt, out);
                stats = keep;
                const double d = max_diff(out, out_ref, out_elems);
                double ds = 0.0;
                if (flags & EPI_STATS) {   // statistics: the kernel's replicas summed, against the fp64 sums of the reference output
                    std::vector<StatCell> b(kStatReps * 8 * 2 * 1024);
                    CK(hipMemcpy(b.data(), stats2, b.size() * sizeof(StatCell), hipMemcpyDeviceToHost));
                    for (int i = 0; i < g.groups * 2 * g.Cout; ++i) {
                        double y = 0;   // (deterministic mode: 2^-40 fixed point in two integer words; default: a double in the first word -- conv.h StatCell)
                        static const bool det = [] { const char* e = getenv("OCL_DETERMINISTIC"); return e && e[0] == '1'; }();
                        for (int r = 0; r < kStatReps; ++r) {
                            const StatCell& cell = b[(size_t)r * 8 * 2 * 1024 + i];
                            double dv;
                            memcpy(&dv, &cell.lo, 8);
                            y += det ? (double)cell.hi / 256.0 + (double)cell.lo / 1099511627776.0 : dv;
                        }
                        ds = fmax(ds, fabs(ref_stats[i] - y) / (1.0 + fabs(ref_stats[i])));
                    }
                }
                StatCell* keep2 = stats;
                stats = stats2;
                const double t = time_us([&] { run(pt, out); });
                stats = keep2;
                if (mt == 0 && pt.cs && getenv("KBENCH_TRACE")) {   // conv_s_kernel: 8 stamps per wave
                    const int nwg = pt.grid_x * pt.grid_y;
                    unsigned long long* tr;
                    CK(hipMalloc(&tr, (size_t)nwg * 64 * 8));
                    CK(hipMemset(tr, 0, (size_t)nwg * 64 * 8));
                    ConvPlan ptt = pt;
                    ptt.a.trace = tr;
                    stats = stats2;
                    run(ptt, out);
                    stats = keep2;
                    CK(hipDeviceSynchronize());
                    std::vector<unsigned long long> h((size_t)nwg * 64);
                    CK(hipMemcpy(h.data(), tr, h.size() * 8, hipMemcpyDeviceToHost));
                    unsigned long long t0 = ~0ull, t1 = 0;
                    for (int w = 0; w < nwg; ++w)
                        for (int e = 0; e < 32; ++e) if (h[(size_t)w * 64 + e]) { t0 = std::min(t0, h[(size_t)w * 64 + e]); t1 = std::max(t1, h[(size_t)w * 64 + e]); }
                    // averages over all workgroups, wave 0 and wave 3: phase lengths in s_memtime ticks (100 MHz on gfx950 -> 10 ns)
                    double av[2][8] = {};
                    double life = 0;
                    for (int w = 0; w < nwg; ++w)
                        for (int wv = 0; wv < 2; ++wv) {
                            const unsigned long long* r = &h[(size_t)w * 64 + (wv ? 3 : 0) * 8];
                            for (int e = 1; e < 7; ++e) av[wv][e] += (double)(r[e] - r[e - 1]);
                            if (!wv) { av[0][7] += (double)(r[7] ? r[7] - r[6] : 0); life += (double)((r[7] ? r[7] : r[6]) - r[0]); }
                        }
                    printf("      trace conv_s: kernel span %llu ticks, %d workgroups; mean wave-0 lifetime %.0f ticks\n", t1 - t0, nwg, life / nwg);
                    for (int wv = 0; wv < 2; ++wv)
                        printf("        wave %d mean: tables+first loads issued %.0f | tables landed, barrier %.0f | weights issued %.0f | patch wait+store %.0f | K loop %.0f | partial store + barrier %.0f | epilogue %.0f\n",
                               wv ? 3 : 0, av[wv][1] / nwg, av[wv][2] / nwg, av[wv][3] / nwg, av[wv][4] / nwg, av[wv][5] / nwg, av[wv][6] / nwg, av[wv][7] / nwg);
                    for (int w : {0, nwg / 2, nwg - 1}) {
                        const unsigned long long* r = &h[(size_t)w * 64];
                        printf("        wg %5d wave 0: +%6llu |", w, r[0] - t0);
                        for (int e = 1; e < 8 && r[e]; ++e) printf(" %llu", r[e] - r[e - 1]);
                        printf("\n");
                    }
                    CK(hipFree(tr));
                }
                if (mt == 0 && !pt.cs && !pt.a.wres && getenv("KBENCH_TRACE")) {   // staged weights: raw phase deltas of wave 0 of two workgroups
                    const int nwg = pt.grid_x * pt.grid_y;
                    unsigned long long* tr;
                    CK(hipMalloc(&tr, (size_t)nwg * 64 * 8));
                    CK(hipMemset(tr, 0, (size_t)nwg * 64 * 8));
                    ConvPlan ptt = pt;
                    ptt.a.trace = tr;
                    stats = stats2;
                    run(ptt, out);
                    stats = keep2;
                    CK(hipDeviceSynchronize());
                    std::vector<unsigned long long> h((size_t)nwg * 64);
                    CK(hipMemcpy(h.data(), tr, h.size() * 8, hipMemcpyDeviceToHost));
                    unsigned long long t0 = ~0ull, t1 = 0;
                    for (int w = 0; w < nwg; ++w) { t0 = std::min(t0, h[(size_t)w * 64]); for (int e = 0; e < 64; ++e) t1 = std::max(t1, h[(size_t)w * 64 + e]); }
                    printf("      trace (staged): kernel span %llu cycles; nstage=%d chunks=%d; per workgroup: start | prologue x5 | per chunk: setup barrier patch-store, then per stage [commit-wait barrier mfma] ... | epilogue\n",
                           t1 - t0, pt.a.nstage, pt.a.Cin / pt.a.KC);
                    for (int w : {0, nwg / 2}) {
                        const unsigned long long* r = &h[(size_t)w * 64];
                        printf("      wg %4d: +%6llu |", w, r[0] - t0);
                        for (int e = 1; e < 64 && r[e]; ++e) printf(" %llu", r[e] - r[e - 1]);
                        printf("\n");
                    }
                    CK(hipFree(tr));
                }
                if (mt == 0 && pt.cw && getenv("KBENCH_TRACE")) {   // conv_w_kernel: 32 stamps per wave, 4 waves per workgroup
                    const int nwg = pt.grid_x * pt.grid_y;
                    unsigned long long* tr;
                    CK(hipMalloc(&tr, (size_t)nwg * 128 * 8));
                    CK(hipMemset(tr, 0, (size_t)nwg * 128 * 8));
                    ConvPlan ptt = pt;
                    ptt.a.trace = tr;
                    stats = stats2;
                    run(ptt, out);
                    CK(hipMemset(tr, 0, (size_t)nwg * 128 * 8));
                    run(ptt, out);   // (second launch: warm caches)
                    stats = keep2;
                    CK(hipDeviceSynchronize());
                    std::vector<unsigned long long> h((size_t)nwg * 128);
                    CK(hipMemcpy(h.data(), tr, h.size() * 8, hipMemcpyDeviceToHost));
                    unsigned long long t0 = ~0ull, t1 = 0;
                    for (size_t i = 0; i < h.size(); ++i) if (h[i]) { t0 = std::min(t0, h[i]); t1 = std::max(t1, h[i]); }
                    double pro[3] = {0, 0, 0}, ph[2] = {0, 0}, tail = 0, life = 0;
                    long nt_ = 0, nw_ = 0;
                    for (int w = 0; w < nwg * 4; ++w) {
                        const unsigned long long* r = &h[(size_t)w * 32];
                        if (!r[0] || !r[31]) continue;
                        ++nw_;
                        for (int k = 0; k < 3; ++k) pro[k] += (double)(r[k + 1] - r[k]);
                        int e = 4, last = 3;
                        for (; e + 1 < 31 && r[e + 1]; e += 2) {
                            ph[0] += (double)(r[e] - r[e - 1]); ph[1] += (double)(r[e + 1] - r[e]);
                            ++nt_;
                            last = e + 1;
                        }
                        tail += (double)(r[31] - r[last]);
                        life += (double)(r[31] - r[0]);
                    }
                    if (nw_ && nt_)
                        printf("      trace conv_w: span %llu ticks; %ld waves, %.2f items each | requests %5.0f  tables %5.0f  wait+barrier %5.0f | per item: K loop %6.0f  next request + epilogue %5.0f | flush %5.0f  lifetime %6.0f\n",
                               t1 - t0, nw_, (double)nt_ / nw_, pro[0] / nw_, pro[1] / nw_, pro[2] / nw_, ph[0] / nt_, ph[1] / nt_, tail / nw_, life / nw_);
                    for (int wg : {0, nwg / 2}) {
                        for (int wv : {0, 3}) {
                            const unsigned long long* r = &h[((size_t)wg * 4 + wv) * 32];
                            if (pt.cw == 2 && wv == 0 && wg == 0) printf("        (conv_wx: per item four stamps: first third of the K loop (loads), second third, last third (stores), epilogue)\n");
                            printf("        wg %4d wave %d: +%6llu |", wg, wv, r[0] - t0);
                            for (int e = 1; e < 31 && r[e]; ++e) printf(" %llu", r[e] - r[e - 1]);
                            printf(" | end +%llu\n", r[31] - t0);
                        }
                    }
                    CK(hipFree(tr));
                }
                if (mt == 0 && pt.q4 == 2 && getenv("KBENCH_TRACE")) {   // conv_q_kernel<2, 12, *>: mean phase lengths over the workgroups (s_memtime ticks)
                    const int nwg = pt.grid_x * pt.grid_y;
                    unsigned long long* tr;
                    CK(hipMalloc(&tr, (size_t)nwg * 64 * 8));
                    CK(hipMemset(tr, 0, (size_t)nwg * 64 * 8));
                    ConvPlan ptt = pt;
                    ptt.a.trace = tr;
                    stats = stats2;
                    run(ptt, out);
                    CK(hipMemset(tr, 0, (size_t)nwg * 64 * 8));
                    run(ptt, out);   // (second launch: warm caches)
                    stats = keep2;
                    CK(hipDeviceSynchronize());
                    std::vector<unsigned long long> h((size_t)nwg * 64);
                    CK(hipMemcpy(h.data(), tr, h.size() * 8, hipMemcpyDeviceToHost));
                    double pro = 0, ph[4] = {0, 0, 0, 0}, fl = 0, life = 0;
                    long ntile = 0;
                    int nlive = 0;
                    unsigned long long t0 = ~0ull, t1 = 0;
                    for (int w = 0; w < nwg; ++w) {
                        const unsigned long long* r = &h[(size_t)w * 64];
                        if (!r[0]) continue;   // (a workgroup without tiles, or not a trace build)
                        int last = 0;
                        while (last + 1 < 64 && r[last + 1]) ++last;
                        if (last < 2) continue;
                        ++nlive;
                        t0 = std::min(t0, r[0]); t1 = std::max(t1, r[last]);
                        pro += (double)(r[1] - r[0]);
                        life += (double)(r[last] - r[0]);
                        int e = 1;
                        for (; e + 4 <= last; e += 4) {
                            for (int k = 0; k < 4; ++k) ph[k] += (double)(r[e + k + 1] - r[e + k]);
                            ++ntile;
                        }
                        if (e < last) fl += (double)(r[last] - r[e]);
                    }
                    if (nlive && ntile)
                        printf("      trace conv_q: %d workgroups, %.1f tiles each, span %llu | prologue (tables, weights, first patch) %6.0f  lifetime %7.0f  flush %5.0f | per tile: barrier1 %5.0f  store + next request + barrier2 %5.0f  K loop %6.0f  epilogue %6.0f\n",
                               nlive, (double)ntile / nlive, t1 - t0, pro / nlive, life / nlive, fl / nlive, ph[0] / ntile, ph[1] / ntile, ph[2] / ntile, ph[3] / ntile);
                    else
                        printf("      trace conv_q: no stamps (not a trace build of this plan)\n");
                    CK(hipFree(tr));
                }
                if (mt == 0 && !pt.cs && !pt.q4 && !pt.cw && pt.a.wres && getenv("KBENCH_TRACE")) {   // phase timeline of wave 0 of a few workgroups (s_memtime cycles)
                    const int nwg = pt.grid_x * pt.grid_y;
                    unsigned long long* tr;
                    CK(hipMalloc(&tr, (size_t)nwg * 64 * 8));
                    CK(hipMemset(tr, 0, (size_t)nwg * 64 * 8));
                    ConvPlan ptt = pt;
                    ptt.a.trace = tr;
                    stats = stats2;
                    run(ptt, out);
                    stats = keep2;
                    CK(hipDeviceSynchronize());
                    std::vector<unsigned long long> h((size_t)nwg * 64);
                    CK(hipMemcpy(h.data(), tr, h.size() * 8, hipMemcpyDeviceToHost));
                    unsigned long long t0 = ~0ull, t1 = 0;
                    for (int w = 0; w < nwg; ++w) { t0 = std::min(t0, h[(size_t)w * 64]); for (int e = 0; e < 64; ++e) t1 = std::max(t1, h[(size_t)w * 64 + e]); }
                    printf("      trace: kernel span %llu cycles; per workgroup [start-offset | setup | per tile: wait-barrier1, patch-wait+store, barrier2, mfma, epilogue, next-setup ...]\n", t1 - t0);
                    for (int w : {0, 1, nwg / 2, nwg - 1}) {
                        const unsigned long long* r = &h[(size_t)w * 64];
                        printf("      wg %4d: +%6llu | tables %4llu barrier %4llu patch-issue %4llu dma-issue %4llu wait+rest %5llu |", w, r[0] - t0, r[1] - r[0],
                               r[2] - r[1], r[3] - r[2], r[4] - r[3], r[5] - r[4]);
                        for (int e = 6; e + 5 < 64 && r[e + 5]; e += 6)
                            printf(" %4llu %5llu %4llu %5llu %5llu %4llu |", r[e + 1] - r[e], r[e + 2] - r[e + 1], r[e + 3] - r[e + 2], r[e + 4] - r[e + 3],
                                   r[e + 5] - r[e + 4], r[e + 6] ? r[e + 6] - r[e + 5] : 0ull);
                        printf("\n");
                    }
                    CK(hipFree(tr));
                }
                printf("    conv_%c%s%s MT=%d NT=%d grid=%5dx%d lds=%6zu KC=%3d Q=%3d QS=%3d res=%d  %7.1f us %6.1f TF/s  maxdiff=%.2e statdiff=%.1e%s\n",
                       pt.cw ? 'w' : pt.cs ? 's' : pt.q4 ? 'q' : 't', mt ? "      " : (pipe == 2 ? " (no-q)" : " (auto)"), pipe == 1 ? " ring" : "     ", pt.MT, pt.NT, pt.grid_x, pt.grid_y, pt.lds_bytes, pt.a.KC, pt.a.Qc, pt.a.QS, pt.a.wres, t,
                       flops / t * 1e-6, d, ds, (d > 1e-3 || ds > 1e-3) ? "  <-- MISMATCH" : "");
            }
        CK(hipFree(stats2));
    }
    CK(hipFree(wT));
}

// ---- calibration: what the f32 MFMA pipe delivers on this box ---------------------------------------------------------
// __launch_bounds__(256, 2) keeps the accumulators in ArchVGPRs, as in conv_t_kernel.  Without it (round 1 / 2 measurements,
// profiles/r1_mfma_peak_calibration.txt) the compiler put them in AccVGPRs and, unable to coalesce the tuples across the loop's back
// edge, added `s_nop 7` + 8 accvgpr copies to EVERY iteration (~70 cycles per 4 or 8 MFMAs): those files UNDERSTATE the 16x16x4
// rate (32 + 70/4 = 50 and 32 + 70/8 = 41 cycles are exactly what they show); the 32x32x2 loop was clean.
// rnd (KBENCH_PEAK_RANDOM=1): per-lane pseudo-random operands instead of the few constant values (does the rate depend on the data?)
template <int NACC, bool LDS>
__global__ void __launch_bounds__(256, 2) mfma_peak_kernel(float* out, int iters, int rnd) {
    __shared__ float sm[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) sm[i] = (float)(i & 7) * 0.125f;
    __syncthreads();
    f32x4 acc[NACC];
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a = (float)(threadIdx.x & 3) * 0.25f, b = 1.0f;
    if (rnd) {
        unsigned h = (threadIdx.x + 1u) * 2654435761u + blockIdx.x * 40503u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        a = __uint_as_float(0x3f000000u | (h & 0x007fffffu)) - 0.75f;              // mantissa bits all random, |a| < 0.25
        h *= 3266489917u; h ^= h >> 16;
        b = __uint_as_float(0x3f000000u | (h & 0x007fffffu)) - 0.75f;
    }
    const int lane = threadIdx.x & 63;
    for (int i = 0; i < iters; ++i) {
        if (LDS) {   // the conv main loop's operand traffic: one A and one B ds_read_b32 per pair of MFMAs
            float av[NACC / 2 > 0 ? NACC / 2 : 1], bv[2];
#pragma unroll
            for (int j = 0; j < NACC / 2; ++j) av[j] = sm[(lane * 22 + j * 64 + i * 4) & 4095];
            bv[0] = sm[(lane + i * 48) & 4095];
            bv[1] = sm[(lane + 16 + i * 48) & 4095];
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j / 2], bv[j & 1], acc[j], 0, 0, 0);
        } else {
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// VGPR banks of the two source operands.  A and B come from two 16-byte LDS reads (register tuples); SAME pairs component k with
// component k (the same register index mod 4 when the tuples are 4-aligned -- every conv kernel here does that), otherwise k with
// (k + 2) % 4.  The ISA decides what is really measured: check `v_mfma ... vA, vB` register numbers.
template <int NACC, bool SAME>
__global__ void __launch_bounds__(256, 2) mfma_bank_kernel(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float sm[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) sm[i] = (float)(i & 7) * 0.125f;
    __syncthreads();
    f32x4 acc[NACC];
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float4 a = *(const float4*)&sm[(threadIdx.x & 63) * 4], b = *(const float4*)&sm[1024 + (threadIdx.x & 63) * 4];
    for (int i = 0; i < iters; ++i) {
#define KB_STEP(E, F) _Pragma("unroll") for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.E, b.F, acc[j], 0, 0, 0);
        if (SAME) { KB_STEP(x, x) KB_STEP(y, y) KB_STEP(z, z) KB_STEP(w, w) }
        else      { KB_STEP(x, z) KB_STEP(y, w) KB_STEP(z, x) KB_STEP(w, y) }
#undef KB_STEP
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC, bool SAME>
static void bank_case(int blocks_per_cu, float* out) {
    const int blocks = 256 * blocks_per_cu, iters = 1000;
    const double t = time_us([&] { hipLaunchKernelGGL((mfma_bank_kernel<NACC, SAME>), dim3(blocks), dim3(256), 0, 0, out, iters); }, 5, 1);
    const double flops = (double)blocks * 4 * iters * 4 * NACC * 2.0 * 16 * 16 * 4;
    printf("peak 16x16x4, A / B in %-20s acc=%d blocks/CU=%d  %7.1f us  %6.1f TF/s  (%.1f cycles/MFMA/SIMD at 2.4 GHz)\n", SAME ? "the same VGPR bank" : "different VGPR banks",
           NACC, blocks_per_cu, t, flops / t * 1e-6, t * 2400.0 / ((double)iters * 4 * NACC * blocks_per_cu));
}

// ---- v_mfma_f32_4x4x1_16b_f32 (sixteen independent 4x4 outer products per instruction): the shape that would carry the 20-channel
// layers without padding the channels to 32 (blocks = 16 groups of 4 pixels, A = 4 output channels broadcast to every block, B = one
// input value per pixel lane).  Rate with NACC independent accumulators, registers only; and with the operand traffic such a
// convolution would have (per (tap, channel-quad) group: 5 A reads + NT B reads of 16 bytes for 20 * NT MFMAs).
template <int NACC>
__global__ void __launch_bounds__(256, 2) mfma_peak4_kernel(float* out, int iters) {
    f32x4 acc[NACC];
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float a = (float)(threadIdx.x & 3) * 0.25f, b = 1.0f + (float)(threadIdx.x & 7);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[j], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NT>
__global__ void __launch_bounds__(256, 2) mfma_peak4_lds_kernel(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float sm[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) sm[i] = (float)(i & 7) * 0.125f;
    __syncthreads();
    f32x4 acc[5][NT];
#pragma unroll
    for (int m = 0; m < 5; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int lane = threadIdx.x & 63;
    float4 av[2][5], bv[2][NT];
    auto fetch = [&](int set, int i) __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < 5; ++m) av[set][m] = *(const float4*)&sm[((i * 20 + m * 4 + (lane & 3)) * 4) & 4095];          // 4 distinct addresses per read (broadcast)
#pragma unroll
        for (int n = 0; n < NT; ++n) bv[set][n] = *(const float4*)&sm[4096 + ((lane * 5 + n * 320 + i * 4) * 4 & 4095)];    // one pixel per lane, odd 16-byte stride
    };
    auto fma = [&](int set) __attribute__((always_inline)) {
#define KB_STEP(E) _Pragma("unroll") for (int m = 0; m < 5; ++m) _Pragma("unroll") for (int n = 0; n < NT; ++n) \
        acc[m][n] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[set][m].E, bv[set][n].E, acc[m][n], 0, 0, 0);
        KB_STEP(x) KB_STEP(y) KB_STEP(z) KB_STEP(w)
#undef KB_STEP
    };
    fetch(0, 0);
    for (int i = 0; i < iters; i += 2) {
        fetch(1, i + 1);
        __builtin_amdgcn_sched_barrier(0);
        fma(0);
        __builtin_amdgcn_sched_barrier(0);
        fetch(0, i + 2);
        __builtin_amdgcn_sched_barrier(0);
        fma(1);
        __builtin_amdgcn_sched_barrier(0);
    }
    float s = 0.f;
#pragma unroll
    for (int m = 0; m < 5; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) s += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
static void peak4_case(int blocks_per_cu, float* out) {
    const int blocks = 256 * blocks_per_cu, iters = 2000;
    const double t = time_us([&] { hipLaunchKernelGGL((mfma_peak4_kernel<NACC>), dim3(blocks), dim3(256), 0, 0, out, iters); }, 5, 1);
    const double flops = (double)blocks * 4 * iters * NACC * 2.0 * 4 * 4 * 16;
    printf("peak 4x4x1_16b regs only              acc=%2d blocks/CU=%d  %7.1f us  %6.1f TF/s  (%.1f cycles/MFMA/SIMD at 2.4 GHz)\n", NACC, blocks_per_cu, t,
           flops / t * 1e-6, t * 2400.0 / ((double)iters * NACC * blocks_per_cu));
}
template <int NT>
static void peak4_lds_case(int blocks_per_cu, float* out) {
    const int blocks = 256 * blocks_per_cu, iters = 400;
    const double t = time_us([&] { hipLaunchKernelGGL((mfma_peak4_lds_kernel<NT>), dim3(blocks), dim3(256), 0, 0, out, iters); }, 5, 1);
    const double flops = (double)blocks * 4 * iters * 20 * NT * 2.0 * 4 * 4 * 16;
    printf("peak 4x4x1_16b 5 A + %d B reads / %3d MFMAs    blocks/CU=%d  %7.1f us  %6.1f TF/s  (%.1f cycles/MFMA/SIMD at 2.4 GHz)\n", NT, 20 * NT, blocks_per_cu, t,
           flops / t * 1e-6, t * 2400.0 / ((double)iters * 20 * NT * blocks_per_cu));
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
// 32x32x2 variant: lane l holds A[i=l&31][k=l>>5], B[k=l>>5][j=l&31]
template <int NACC, bool LDS>
__global__ void __launch_bounds__(256) mfma_peak32_kernel(float* out, int iters) {
    __shared__ float sm[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) sm[i] = (float)(i & 7) * 0.125f;
    __syncthreads();
    f32x16 acc[NACC];
#pragma unroll
    for (int j = 0; j < NACC; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    float a = (float)(threadIdx.x & 3) * 0.25f, b = 1.0f;
    const int lane = threadIdx.x & 63;
    for (int i = 0; i < iters; ++i) {
        if (LDS) {
            float av[NACC], bv;
#pragma unroll
            for (int j = 0; j < NACC; ++j) av[j] = sm[((lane & 31) * 22 + (lane >> 5) + j * 704 + i * 2) & 4095];
            bv = sm[((lane >> 5) * 48 + (lane & 31) + i * 96) & 4095];
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bv, acc[j], 0, 0, 0);
        } else {
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NACC; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[j][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC, bool LDS>
static void peak32_case(const char* name, int blocks_per_cu, float* out) {
    const int iters = 2000;
    const int blocks = 256 * blocks_per_cu;
    const double t = time_us([&] { hipLaunchKernelGGL((mfma_peak32_kernel<NACC, LDS>), dim3(blocks), dim3(256), 0, 0, out, iters); }, 5, 1);
    const double flops = (double)blocks * 4 * iters * NACC * 4096.0;
    printf("peak32x32x2 %-26s acc=%d blocks/CU=%d  %8.1f us  %6.1f TF/s  (%.1f cycles/MFMA/SIMD at 2.4 GHz)\n", name, NACC, blocks_per_cu, t,
           flops / t * 1e-6, t * 1e-6 * 2.4e9 / ((double)blocks_per_cu * iters * NACC));
}

template <int NACC, bool LDS>
static void peak_case(const char* name, int blocks_per_cu, float* out) {
    const int iters = 4000;
    const int blocks = 256 * blocks_per_cu;
    const double t = time_us([&] { hipLaunchKernelGGL((mfma_peak_kernel<NACC, LDS>), dim3(blocks), dim3(256), 0, 0, out, iters, getenv("KBENCH_PEAK_RANDOM") ? 1 : 0); }, 5, 1);
    const double flops = (double)blocks * 4 * iters * NACC * 2048.0;
    printf("peak %-28s acc=%d blocks/CU=%d  %8.1f us  %6.1f TF/s  (%.1f cycles/MFMA/SIMD at 2.4 GHz)\n", name, NACC, blocks_per_cu, t,
           flops / t * 1e-6, t * 1e-6 * 2.4e9 / ((double)blocks_per_cu * iters * NACC));
}


// ---- launch-cost probe: n dependent kernels of ~dur_us each, issued as stream launches or as one captured graph ----------
__global__ void spin_kernel(float* p, int iters) {
    float v = p[threadIdx.x];
    for (int i = 0; i < iters; ++i) v = fmaf(v, 1.0001f, 0.5f);
    p[threadIdx.x] = v;
}
struct FatArgs { float* p; int iters; int pad[90]; };   // ~376 B of kernel arguments, like ConvArgs
__global__ void spin_kernel_fat(const FatArgs a) {
    float v = a.p[threadIdx.x];
    for (int i = 0; i < a.iters; ++i) v = fmaf(v, 1.0001f, 0.5f);
    a.p[threadIdx.x] = v;
}
static double now_us() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}
static void launch_probe() {
    float* p;
    CK(hipMalloc(&p, 1 << 20));
    CK(hipMemset(p, 0, 1 << 20));
    hipStream_t st, s2;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    std::vector<hipEvent_t> evs(512);
    for (auto& e : evs) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    const int n = 200;
    // variant 0: linear chain; 1: + a memset node every 10 kernels; 2: fork/join to a second stream every 5 kernels (1 kernel on the
    // side branch, joined at the end only); 3: like 2 but joined back before the next fork
    for (int variant = 0; variant < 4; ++variant)
    for (int iters : {0, 2000}) {
        for (int fat = 1; fat < 2; ++fat) {
            auto k = [&](hipStream_t s, int off) {
                FatArgs a; memset(&a, 0, sizeof(a)); a.p = p + off * 256; a.iters = iters;
                hipLaunchKernelGGL(spin_kernel_fat, dim3(256), dim3(256), 0, s, a);
            };
            auto issue = [&](hipStream_t s) {
                int ne = 0;
                for (int i = 0; i < n; ++i) {
                    k(s, 0);
                    if (variant == 1 && i % 10 == 0) CK(hipMemsetAsync(p + 512 * 256, 0, 4096, s));
                    if (variant >= 2 && i % 5 == 0) {
                        CK(hipEventRecord(evs[ne], s));
                        CK(hipStreamWaitEvent(s2, evs[ne], 0));
                        ++ne;
                        k(s2, 1 + i);
                        if (variant == 3) {
                            CK(hipEventRecord(evs[ne], s2));
                            CK(hipStreamWaitEvent(s, evs[ne], 0));
                            ++ne;
                        }
                    }
                }
                if (variant >= 2) {
                    CK(hipEventRecord(evs[ne], s2));
                    CK(hipStreamWaitEvent(s, evs[ne], 0));
                }
            };
            issue(st);
            CK(hipStreamSynchronize(st));
            double h0 = now_us();
            issue(st);
            double h1 = now_us();
            CK(hipStreamSynchronize(st));
            double h2 = now_us();
            hipGraph_t g;
            hipGraphExec_t ge;
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            issue(st);
            CK(hipStreamEndCapture(st, &g));
            double i0 = now_us();
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            double i1 = now_us();
            CK(hipGraphLaunch(ge, st));
            CK(hipStreamSynchronize(st));
            double g0 = now_us();
            CK(hipGraphLaunch(ge, st));
            double g1 = now_us();
            CK(hipStreamSynchronize(st));
            double g2 = now_us();
            // launched on the legacy default stream, as torch does
            CK(hipGraphLaunch(ge, 0));
            CK(hipStreamSynchronize(0));
            double d0 = now_us();
            CK(hipGraphLaunch(ge, 0));
            double d1 = now_us();
            CK(hipStreamSynchronize(0));
            double d2 = now_us();
            printf("launch probe variant %d iters=%5d: stream: host %.2f us/launch, total %.2f | graph: instantiate %.0f us, host %.2f us/node, total %.2f | on stream 0: host %.2f total %.2f\n",
                   variant, iters, (h1 - h0) / n, (h2 - h0) / n, i1 - i0, (g1 - g0) / n, (g2 - g0) / n, (d1 - d0) / n, (d2 - d0) / n);
            CK(hipGraphExecDestroy(ge));
            CK(hipGraphDestroy(g));
        }
    }
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 220;
    const int groups = argc > 2 ? atoi(argv[2]) : 2;
    const int hw = argc > 3 ? atoi(argv[3]) : 32;
    const std::string mode = argc > 4 ? argv[4] : "all";
    const bool sweep = argc > 5 ? atoi(argv[5]) != 0 : true;
    std::vector<Layer> layers = make_layers(hw, 20);
    if (mode == "cover") {   // host-only: every plan's tables replayed on the host -- the lane -> pixel maps of all tiles must hit every
        // output pixel of the plan's lattice exactly once, every operand read must stay inside the patch, every patch unit inside the
        // patch and (where it is loaded at all) inside the input tensor.  Mirrors the pixel-geometry statements of conv_t_kernel /
        // conv_q_kernel / conv_s_kernel (aligned: the plan's per-lane table; unaligned: the per-tile division).
        int bad = 0, checked = 0;
        for (auto& l : layers) {
            const ConvShape& c = l.s;
            ConvGeomDesc g;
            geom_fwd(c, N, groups, &g);
            std::vector<ConvGeomDesc> all(1, g), dg;
            if (c.Cin != 3) {
                geom_dgrad(c, N, &dg, true);
                all.insert(all.end(), dg.begin(), dg.end());
                dg.clear();
                geom_dgrad(c, N, &dg, false);   // the four parity classes as separate launches
                all.insert(all.end(), dg.begin(), dg.end());
            }
            for (size_t gi = 0; gi < all.size(); ++gi) {
                ConvPlan p;
                OK(plan_conv(all[gi], &p));
                std::vector<int> b;
                conv_plan_tables(p, &b);
                // (self-test of the checker, KBENCH_COVER_SELFTEST=1: read the per-tile maps of unaligned plans as if they were aligned --
                // the check must then fail on the 84 x 84 lattices)
                if (getenv("KBENCH_COVER_SELFTEST") && !p.a.aligned) p.a.aligned = 1;
                const ConvArgs& a = p.a;
                if (p.cw) {   // conv_w_kernel / conv_wx_kernel: the wave tiles' geometry is arithmetic (decode), the lane -> pixel map and the staging units are tables
                    const int NTw = p.NT, T = (a.N / a.imgs) * a.tiles_per_img, SL = a.CP / 4, kc4 = a.KC / 4;
                    const int* qoffw = b.data() + 16;
                    const int* uw = b.data() + a.off_pu;
                    const int* lcw = b.data() + a.off_loc;
                    std::vector<unsigned char> hitw((size_t)a.N * a.Hout * a.Wout, 0);
                    int errs = 0;
                    auto failw = [&](const char* what, int tile, int r) {
                        if (errs++ < 3) printf("  COVER %s conv_w geom %zu: %s (tile %d, pixel %d)\n", l.name.c_str(), gi, what, tile, r);
                    };
                    const int64_t in_bytes = (int64_t)a.N * a.Hin * a.Win * a.Cin * 4;
                    for (int t = 0; t < T; ++t) {
                        const int ti = t / a.tiles_per_img, tp = t % a.tiles_per_img;
                        const int img0 = ti * a.imgs, p0 = tp * a.ppi;
                        const int ly0 = p0 / a.LW, lx0 = p0 % a.LW;
                        const int iy0 = ly0 * a.is + a.min_dy, ix0 = lx0 * a.is + a.min_dx;
                        const int64_t in_base = ((((int64_t)img0 * a.Hin + iy0) * a.Win + ix0) * a.Cin) * 4;
                        const int obase = ((img0 * a.Hout + ly0 * a.os + a.oy0) * a.Wout + lx0 * a.os + a.ox0) * a.Cout;
                        const int nimg = std::min(a.imgs, a.N - img0);
                        if (img0 / a.group_size != (img0 + nimg - 1) / a.group_size) failw("tile straddles two BatchNorm groups", t, 0);
                        for (int r = 0; r < 16 * NTw; ++r) {
                            const int nt = r / 16, r16 = r % 16;
                            const int lp = lcw[(3 * nt + 0) * 16 + r16], lo = lcw[(3 * nt + 1) * 16 + r16], il = lcw[(3 * nt + 2) * 16 + r16];
                            if (il >= nimg) continue;
                            const int o = obase + lo;
                            if (o % a.Cout || o < 0 || o / a.Cout >= (int)hitw.size()) { failw("output offset outside the tensor", t, r); continue; }
                            if (hitw[o / a.Cout]++) failw("output pixel written twice", t, r);
                            for (int q = 0; q < a.Qpad; ++q)
                                if (lp + qoffw[q] < 0 || lp + qoffw[q] + 4 > a.patch_floats) { failw("operand read outside the patch", t, r); break; }
                            // every tap of the pixel must read the unit the staging table puts there: patch slot -> (image, row, column, quad)
                            for (int tq = 0; tq < a.ntaps && errs < 3; ++tq) {
                                const int slot = (lp + a.tpo[tq]) / 4;                       // 16-byte slot of channel quad 0 of the tap
                                const int pix = slot / SL, pc = pix % a.PC, row = pix / a.PC, pr = row % a.PR, ilp = row / a.PR;
                                const int ly = (r % a.ppi) / a.LW + ly0, lx = (r % a.ppi) % a.LW + lx0;
                                const int want_y = ly * a.is + a.tdy[tq] - iy0, want_x = lx * a.is + a.tdx[tq] - ix0;
                                if (slot % SL || ilp != il || pr != want_y || pc != want_x) { failw("tap reads the wrong patch slot", t, r); break; }
                            }
                        }
                        if (t % std::max(1, T / 64) == 0)
                            for (int u = 0; u < a.nstage * 64; ++u) {
                                const int w = uw[u];
                                if (w < 0) continue;
                                const int upr = (w >> 19) & 15, upc = (w >> 23) & 63, uil = (w >> 29) & 3, c4 = (w >> 13) & 63;
                                const int s_ = u % SL, pix = u / SL;
                                if (s_ != c4 || c4 >= kc4 || pix % a.PC != upc || (pix / a.PC) % a.PR != upr || (pix / a.PC) / a.PR != uil) { failw("staging unit does not sit at its patch slot", t, u); break; }
                                if ((unsigned)(iy0 + upr) >= (unsigned)a.Hin || (unsigned)(ix0 + upc) >= (unsigned)a.Win || uil >= nimg) continue;
                                for (int ch = 0; ch < a.Cin / a.KC; ++ch) {
                                    const int64_t addr = in_base + (int64_t)ch * a.KC * 4 + ((int64_t)(w & 0x1fff) << 4);
                                    const int64_t want = (((int64_t)(img0 + uil) * a.Hin + iy0 + upr) * a.Win + ix0 + upc) * a.Cin * 4 + ch * a.KC * 4 + c4 * 16;
                                    if (addr != want || addr < 0 || addr + 16 > in_bytes) { failw("staging unit loads the wrong bytes", t, u); break; }
                                }
                            }
                    }
                    size_t want = 0, got = 0;
                    for (int n = 0; n < a.N; ++n)
                        for (int ly = 0; ly < a.LH; ++ly)
                            for (int lx = 0; lx < a.LW; ++lx) {
                                const size_t o = ((size_t)n * a.Hout + ly * a.os + a.oy0) * a.Wout + lx * a.os + a.ox0;
                                ++want;
                                if (o < hitw.size() && hitw[o] == 1) ++got;
                            }
                    if (got != want) failw("lattice pixels missing", -1, (int)(want - got));
                    bad += errs ? 1 : 0;
                    ++checked;
                    continue;
                }
                const int ncls = a.cls_pack & 15, NT = p.NT;
                const int ntiles = a.groups * a.tiles_per_group, LP = a.LH * a.LW;
                const int tile_px = p.cs ? 16 * NT : p.q4 ? 256 * NT : 64 * NT;
                const int* ctab = b.data();
                const int* qoff = ctab + 16;
                const int* td = b.data() + a.off_tdesc;
                const int* lc = b.data() + a.off_loc;
                std::vector<unsigned char> hit((size_t)a.N * a.Hout * a.Wout, 0);
                int errs = 0;
                auto fail = [&](const char* what, int tile, int r) {
                    if (errs++ < 3) printf("  COVER %s %s geom %zu: %s (tile %d, pixel %d)\n", l.name.c_str(), p.cs ? "conv_s" : p.q4 ? "conv_q" : "conv_t", gi, what, tile, r);
                };
                int max_nrows = 0;
                for (int tile = 0; tile < ntiles; ++tile) {
                    const int* d = td + (size_t)tile * 8;
                    const int nrows = d[2], obase = d[3], nimg = d[4], grp = d[5], p0 = d[6], img0 = d[7] & 0xfffff, ly0 = d[7] >> 20;
                    max_nrows = std::max(max_nrows, nrows);
                    const int grp_end = std::min(a.N, (grp + 1) * a.group_size);
                    for (int r = 0; r < tile_px; ++r) {
                        int pbase, ooff;
                        if (a.aligned) {
                            // the table is indexed by (pixel set nt, thread): find a thread whose pixel r is
                            int nt, tid;
                            if (p.cs) { nt = r / 16; tid = r % 16; }
                            else if (p.q4) { const int wave = r / (64 * NT); nt = (r / 64) % NT; tid = wave * 64 + r % 64; }
                            else { const int wave = r / (16 * NT); nt = (r / 16) % NT; tid = wave * 64 + r % 16; }
                            const int lp = lc[(3 * nt + 0) * 256 + tid], lo = lc[(3 * nt + 1) * 256 + tid], il = lc[(3 * nt + 2) * 256 + tid];
                            const bool v = il < nimg;
                            pbase = v ? lp : 0;
                            ooff = v ? obase + lo : -1;
                        } else {
                            const int il = p.cs ? 0 : r / a.ppi, pl = p.cs ? r : r % a.ppi;
                            const int pp = p0 + pl, n = img0 + il;
                            const bool v = il < a.imgs && n < grp_end && pp < LP;
                            const int ly = pp / a.LW, lx = pp % a.LW;
                            pbase = v ? ((il * a.PR + (ly - ly0) * a.is) * a.PC + lx * a.is) * a.CP : 0;
                            ooff = v ? ((n * a.Hout + ly * a.os + a.oy0) * a.Wout + lx * a.os + a.ox0) * a.Cout : -1;
                        }
                        if (ooff < 0) continue;
                        for (int cls = 0; cls < std::max(1, ncls); ++cls) {
                            const int o = ooff + (ncls > 1 ? ctab[cls * 4 + 2] : 0);
                            if (o % a.Cout || o / a.Cout >= (int)hit.size()) { fail("output offset outside the tensor", tile, r); continue; }
                            if (hit[o / a.Cout]++) fail("output pixel written twice", tile, r);
                        }
                        for (int q = 0; q < a.Qpad; ++q)
                            if (pbase + qoff[q] < 0 || pbase + qoff[q] + 4 > a.patch_floats) { fail("operand read outside the patch", tile, r); break; }
                    }
                }
                // expected: every lattice pixel of every image, once per class
                size_t want = 0, got = 0;
                for (int n = 0; n < a.N; ++n)
                    for (int ly = 0; ly < a.LH; ++ly)
                        for (int lx = 0; lx < a.LW; ++lx) {
                            if (ncls > 1) {
                                for (int cls = 0; cls < ncls; ++cls) {
                                    const int o = ((n * a.Hout + ly * a.os + a.oy0) * a.Wout + lx * a.os + a.ox0) * a.Cout + ctab[cls * 4 + 2];
                                    ++want;
                                    if (o / a.Cout < (int)hit.size() && hit[o / a.Cout] == 1) ++got;
                                }
                            } else {
                                const size_t o = ((size_t)n * a.Hout + ly * a.os + a.oy0) * a.Wout + lx * a.os + a.ox0;
                                ++want;
                                if (o < hit.size() && hit[o] == 1) ++got;
                            }
                        }
                if (got != want) { fail("lattice pixels missing", -1, (int)(want - got)); }
                // patch units: inside the patch; loaded ones inside the input tensor
                const int PF = (a.off_loc - a.off_pu) / (3 * 256), lanes = p.cs ? 64 : 256;
                const int* pu = b.data() + a.off_pu;
                const int64_t in_bytes = (int64_t)a.N * a.Hin * a.Win * a.Cin * 4;
                const int nchunk_c0 = p.cs ? 4 : a.Cin / a.KC;
                for (int tile = 0; tile < ntiles && errs < 3; tile += std::max(1, ntiles / 64)) {
                    const int* d = td + (size_t)tile * 8;
                    for (int i = 0; i < PF; ++i)
                        for (int t = 0; t < lanes; ++t) {
                            const int goff = pu[(3 * i + 0) * 256 + t], plds = pu[(3 * i + 1) * 256 + t], rp = pu[(3 * i + 2) * 256 + t];
                            const int row = rp & 0xffff, pr = (rp >> 16) & 0xff;
                            if (row >= d[2]) continue;
                            if (plds < 0 || plds + 4 > a.patch_floats) { fail("patch unit outside the patch", tile, t); break; }
                            if (goff < 0 || (unsigned)(d[1] + pr) >= (unsigned)a.Hin) continue;
                            for (int ch = 0; ch < nchunk_c0; ++ch) {
                                const int64_t addr = (int64_t)d[0] + (int64_t)ch * a.KC * 4 + goff;
                                if (addr < 0 || addr + 16 > in_bytes) { fail("patch load outside the input tensor", tile, t); break; }
                            }
                        }
                }
                (void)max_nrows;
                bad += errs ? 1 : 0;
                ++checked;
            }
        }
        printf("cover: %d plans checked, %d with errors\n", checked, bad);
        return bad ? 1 : 0;
    }
    if (mode == "plan") {   // host-only: print the planner's choices (works without a GPU)
        for (auto& l : layers) {
            const ConvShape& c = l.s;
            ConvGeomDesc g;
            geom_fwd(c, N, groups, &g);
            std::vector<ConvGeomDesc> all(1, g), dg;
            if (c.Cin != 3) geom_dgrad(c, N, &dg, true);
            all.insert(all.end(), dg.begin(), dg.end());
            for (size_t i = 0; i < all.size(); ++i) {
                ConvPlan p;
                ConvGeomDesc g2 = all[i];
                g2.force_pipe = -1;   // this line: resident weights or the two-buffer schedule; the default (ring) plan follows as "ring:"
                OK(plan_conv(g2, &p));
                printf("%-20s %-6s M=%7d N=%3d K=%4d  MT=%d NT=%d grid=%5dx%d lds=%6zu KC=%3d CP=%3d Qpad=%d QS=%d res=%d classes=%d imgs=%d ppi=%d PR=%d PC=%d%s\n",
                       l.name.c_str(), i == 0 ? "fwd" : "dgrad", all[i].N * all[i].LH * all[i].LW, all[i].Cout, all[i].ntaps * all[i].Cin,
                       p.MT, p.NT, p.grid_x, p.grid_y, p.lds_bytes, p.a.KC, p.a.CP, p.a.Qpad, p.a.QS, p.a.wres, p.a.cls_pack & 15, p.a.imgs, p.a.ppi, p.a.PR, p.a.PC,
                       p.cs ? "  conv_s" : p.q4 ? "  conv_q" : "");
                if (!p.a.wres) {   // staged weights: the default plan, when it is the three-buffer ring
                    ConvGeomDesc gp = all[i];
                    gp.force_pipe = 0;
                    ConvPlan pp;
                    if (plan_conv(gp, &pp) == OCL_OK && pp.a.pipe)
                        printf("%-20s %-6s   ring: MT=%d NT=%d grid=%5dx%d lds=%6zu KC=%3d Qpad=%d QS=%d stages=%d\n", "", "", pp.MT, pp.NT, pp.grid_x, pp.grid_y,
                               pp.lds_bytes, pp.a.KC, pp.a.Qpad, pp.a.QS, pp.a.nstage);
                }
            }
            WgradPlan wp;
            const char* wt = getenv("KBENCH_WG_TARGET");   // (the layer's split inside the merged launch of a replay-sized pass: net.hip passes 96)
            OK(plan_wgrad(N, c.Hin, c.Win, c.CinT, c.Ho, c.Wo, c.Cout, c.k, c.stride, &wp, 0, wt ? atoi(wt) : 0));
            printf("%-20s wgrad  M=%4d N=%3d K=%7d  MTW=%d NTW=%d grid=%4dx%3d lds=%6zu KC=%3d CP=%3d S=%4d tiles=%d KP=%d q4=%d xcd=%d multi=%d partial=%6.2f MB\n", l.name.c_str(),
                   c.k * c.k * c.CinT, c.Cout, N * c.Ho * c.Wo, wp.MTW, wp.NTW, wp.grid_x, wp.grid_y, wp.lds_bytes, wp.a.KC, wp.a.CP, wp.a.S,
                   wp.a.total_tiles, wp.a.KP, wp.q_rgw, wp.a.xcd_by, wgrad_multi_variant(wp), wp.partial_floats * 4e-6);
        }
        return 0;
    }
    OK(ocl_init(0));
    OK(conv_kernels_init());
    size_t max_act = 0, max_w = 0;
    for (auto& l : layers) {
        max_act = std::max(max_act, (size_t)N * l.s.Hin * l.s.Win * l.s.CinT);
        max_act = std::max(max_act, (size_t)N * l.s.Ho * l.s.Wo * l.s.Cout);
        max_w = std::max(max_w, (size_t)9 * std::max(l.s.CinT * l.s.CoutP, l.s.Cout * std::max(l.s.CiP, 16)));
    }
    float* bufA = dev_rand(max_act, 1);
    float* bufB = dev_rand(max_act, 2);
    float* bufC = dev_rand(max_act, 3);
    float* bufD = dev_rand(max_act, 4);
    float* w = dev_rand(max_w + 4096, 5, 0.2f);
    StatCell* stats;
    CK(hipMalloc(&stats, kStatReps * 8 * 2 * 1024 * sizeof(StatCell)));
    CK(hipMemset(stats, 0, kStatReps * 8 * 2 * 1024 * sizeof(StatCell)));
    printf("# kbench N=%d groups=%d hw=%d\n", N, groups, hw);
    if (mode == "launch") { launch_probe(); return 0; }
    if (mode == "peak4") {
        peak4_case<4>(1, bufB); peak4_case<10>(1, bufB); peak4_case<20>(1, bufB); peak4_case<20>(2, bufB);
        peak4_lds_case<2>(1, bufB); peak4_lds_case<2>(2, bufB); peak4_lds_case<4>(1, bufB); peak4_lds_case<4>(2, bufB);
        return 0;
    }
    if (mode == "all" || mode == "peak") {
        // the two source operands in the same / in different VGPR banks
        bank_case<4, true>(1, bufB);
        bank_case<4, false>(1, bufB);
        bank_case<5, true>(1, bufB);
        bank_case<5, false>(1, bufB);
        // one / two accumulators: every MFMA (every other one) depends on the one before it -- the price of a dependent issue
        peak_case<1, false>("regs only", 1, bufB);
        peak_case<2, false>("regs only", 1, bufB);
        peak_case<4, false>("regs only", 1, bufB);
        peak_case<4, false>("regs only", 2, bufB);
        peak_case<8, false>("regs only", 1, bufB);
        peak_case<8, false>("regs only", 2, bufB);
        peak_case<4, true>("with ds_read_b32 operands", 1, bufB);
        peak_case<4, true>("with ds_read_b32 operands", 2, bufB);
        peak_case<4, true>("with ds_read_b32 operands", 3, bufB);
        peak_case<8, true>("with ds_read_b32 operands", 2, bufB);
        peak_case<4, true>("with ds_read_b32 operands", 4, bufB);
        peak_case<6, true>("with ds_read_b32 operands", 3, bufB);
        peak_case<6, true>("with ds_read_b32 operands", 4, bufB);
        peak_case<4, false>("regs only", 4, bufB);
        peak32_case<1, false>("regs only", 1, bufB);
        peak32_case<2, false>("regs only", 1, bufB);
        peak32_case<2, false>("regs only", 2, bufB);
        peak32_case<2, false>("regs only", 3, bufB);
        peak32_case<2, true>("with ds_read_b32 operands", 1, bufB);
        peak32_case<2, true>("with ds_read_b32 operands", 2, bufB);
        peak32_case<2, true>("with ds_read_b32 operands", 3, bufB);
        peak32_case<3, true>("with ds_read_b32 operands", 2, bufB);
    }

    if (mode == "all" || mode == "conv") {
        double tot_auto = 0.0;
        const char* only = getenv("KBENCH_ONLY");   // substring of the layer names to run (e.g. KBENCH_ONLY=layer3.0)
        for (auto& l : layers) {
            if (only && l.name.find(only) == std::string::npos) continue;
            const ConvShape& c = l.s;
            const double macs = (double)N * c.Ho * c.Wo * c.Cout * c.Cin * c.k * c.k;
            ConvGeomDesc g;
            geom_fwd(c, N, groups, &g);
            bench_geom(l.name.c_str(), "fwd", g, bufA, w, bufB, bufC, stats, EPI_STATS, 2.0 * macs, (size_t)N * c.Ho * c.Wo * c.Cout,
                       sweep);
            if (c.Cin != 3) {
                std::vector<ConvGeomDesc> dg;
                geom_dgrad(c, N, &dg);
                int qi = 0;
                for (auto& q : dg) {
                    char kind[16];
                    snprintf(kind, sizeof(kind), "dgrad%d", qi++);
                    const double f = 2.0 * (double)N * q.LH * q.LW * q.Cout * q.Cin * q.ntaps;
                    bench_geom(l.name.c_str(), kind, q, bufA, w, bufB, bufC, stats, 0, f, (size_t)N * c.Hin * c.Win * c.Cin, sweep);
                }
                std::vector<ConvGeomDesc> dm;
                geom_dgrad(c, N, &dm, true);
                if (dg.size() == 4 && dm.size() == 1 && dm[0].ncls > 1) {   // the four parity classes as ONE launch vs four
                    const size_t out_elems = (size_t)N * c.Hin * c.Win * c.Cin;
                    float* wT = make_packT(w, dm[0].Cin, dm[0].WPT);
                    std::vector<ConvPlan> p4(4);
                    for (int i = 0; i < 4; ++i) { OK(plan_conv(dg[i], &p4[i])); OK(conv_plan_finalize(&p4[i])); }
                    for (int fmt = 0; fmt <= (sweep ? 5 : 0); ++fmt)
                    for (int pipe = 0; pipe < 2; ++pipe) {   // merged plan with staged weights: two-buffer schedule, then the ring
                    ConvGeomDesc gm = dm[0];
                    gm.force_MT = fmt;
                    gm.force_pipe = pipe ? 1 : -1;
                    ConvPlan pm;
                    if (plan_conv(gm, &pm) == OCL_OK) {
                        if (pipe && !pm.a.pipe) continue;
                        OK(conv_plan_finalize(&pm));
                        auto run = [&](ConvPlan p, float* o) {
                            p.a.in = bufA; p.a.wT = wT; p.a.out = o; p.a.flags = 0; p.a.stats = stats; p.a.stat_rep_stride = 8 * 2 * 1024;
                            OK(launch_conv(p, 0));
                        };
                        CK(hipMemset(bufC, 0, out_elems * 4));
                        CK(hipMemset(bufB, 0, out_elems * 4));
                        for (auto& p : p4) run(p, bufC);
                        run(pm, bufB);
                        const double d = max_diff(bufB, bufC, out_elems);
                        const double t4 = time_us([&] { for (auto& p : p4) run(p, bufC); });
                        const double tm = time_us([&] { run(pm, bufB); });
                        const double f = 2.0 * (double)N * c.Ho * c.Wo * c.Cout * c.Cin * 9;
                        printf("%-20s dgradM%s 4 launches %7.1f us; merged%s MT=%d NT=%d grid=%5dx%d lds=%6zu KC=%3d Qpad=%3d QS=%3d res=%d  %7.1f us %6.1f TF/s  maxdiff=%.2e%s\n",
                               l.name.c_str(), fmt ? " " : "*", t4, pipe ? " (ring)" : "", pm.MT, pm.NT, pm.grid_x, pm.grid_y, pm.lds_bytes, pm.a.KC, pm.a.Qpad, pm.a.QS, pm.a.wres, tm,
                               f / tm * 1e-6, d, d > 1e-3 ? "  <-- MISMATCH" : "");
                    } else if (!fmt && !pipe) {
                        printf("%-20s dgradM  no merged plan: %s\n", l.name.c_str(), ocl_last_error());
                    }
                    }
                    CK(hipFree(wT));
                }
            }
            (void)tot_auto;
        }
    }
    if (mode == "wgradtrace") {   // where a pixel tile's time goes: s_memtime stamps of thread 0 of every workgroup (measurement builds of the hot forms)
        float* partial = nullptr;
        size_t pf = 0;
        for (auto& l : layers) {
            WgradPlan wp;
            OK(plan_wgrad(N, l.s.Hin, l.s.Win, l.s.CinT, l.s.Ho, l.s.Wo, l.s.Cout, l.s.k, l.s.stride, &wp));
            pf = std::max(pf, wp.partial_floats);
        }
        CK(hipMalloc(&partial, pf * 4 + 4096));
        for (auto& l : layers) {
            const ConvShape& c = l.s;
            WgradPlan wp;
            // (KBENCH_WG_TARGET: the pixel split of a layer inside the merged launch of a replay-sized pass, net.hip: 96)
            static const int env_wt = [] { const char* e = getenv("KBENCH_WG_TARGET"); return e ? atoi(e) : 0; }();
            OK(plan_wgrad(N, c.Hin, c.Win, c.CinT, c.Ho, c.Wo, c.Cout, c.k, c.stride, &wp, 0, env_wt));
            wp.a.x = bufA; wp.a.dy = bufB; wp.a.partial = partial;
            const int nwg = wp.grid_x * wp.grid_y;
            unsigned long long* tr;
            CK(hipMalloc(&tr, (size_t)nwg * 64 * 8));
            CK(hipMemset(tr, 0, (size_t)nwg * 64 * 8));
            wp.a.trace = tr;
            if (launch_wgrad(wp, 0) != OCL_OK) { printf("%-20s (no trace build: %s)\n", l.name.c_str(), ocl_last_error()); CK(hipFree(tr)); continue; }
            OK(launch_wgrad(wp, 0));   // (second launch: warm caches; the stamps of this one are read)
            CK(hipDeviceSynchronize());
            std::vector<unsigned long long> h((size_t)nwg * 64);
            CK(hipMemcpy(h.data(), tr, h.size() * 8, hipMemcpyDeviceToHost));
            // mean over workgroups and tiles of each phase (ticks of s_memtime)
            double ph[5] = {0, 0, 0, 0, 0}, pro = 0, epi = 0, life = 0;
            long ntile = 0;
            unsigned long long t0 = ~0ull, t1 = 0;
            for (int w = 0; w < nwg; ++w) {
                const unsigned long long* r = &h[(size_t)w * 64];
                int last = 0;
                while (last + 1 < 64 && r[last + 1]) ++last;
                t0 = std::min(t0, r[0]); t1 = std::max(t1, r[last]);
                pro += (double)(r[1] - r[0]);
                life += (double)(r[last] - r[0]);
                int e = 1;
                for (; e + 5 <= last; e += 5) {
                    for (int k = 0; k < 5; ++k) ph[k] += (double)(r[e + k + 1] - r[e + k]);
                    ++ntile;
                }
                if (e < last) epi += (double)(r[last] - r[e]);
            }
            printf("%-20s MTW=%d NTW=%d q4=%d grid=%4dx%d tiles/wg=%.1f  span %6llu | per workgroup: prologue %6.0f  lifetime %7.0f  epilogue %6.0f | per tile: barrier1 %5.0f  store %5.0f  barrier2 %5.0f  next-load issue %5.0f  K loop %6.0f  (sum %6.0f ticks)\n",
                   l.name.c_str(), wp.MTW, wp.NTW, wp.q_rgw, wp.grid_x, wp.grid_y, (double)ntile / nwg, t1 - t0, pro / nwg, life / nwg, epi / nwg,
                   ph[0] / ntile, ph[1] / ntile, ph[2] / ntile, ph[3] / ntile, ph[4] / ntile, (ph[0] + ph[1] + ph[2] + ph[3] + ph[4]) / ntile);
            CK(hipFree(tr));
        }
        CK(hipFree(partial));
        return 0;
    }
    if (mode == "all" || mode == "wgrad") {
        float* partial = nullptr;
        size_t pf = 0;
        for (auto& l : layers) {
            WgradPlan wp;
            OK(plan_wgrad(N, l.s.Hin, l.s.Win, l.s.CinT, l.s.Ho, l.s.Wo, l.s.Cout, l.s.k, l.s.stride, &wp));
            pf = std::max(pf, wp.partial_floats);
        }
        CK(hipMalloc(&partial, pf * 4 + 4096));
        float* grad = dev_rand(max_w + 4096, 7);
        float* grad_ref = dev_rand(max_w + 4096, 8);
        for (auto& l : layers) {
            const ConvShape& c = l.s;
            WgradPlan wp;
            OK(plan_wgrad(N, c.Hin, c.Win, c.CinT, c.Ho, c.Wo, c.Cout, c.k, c.stride, &wp));
            wp.a.x = bufA; wp.a.dy = bufB; wp.a.partial = partial;
            const double macs = (double)N * c.Ho * c.Wo * c.Cout * c.Cin * c.k * c.k;
            const double t1 = time_us([&] { OK(launch_wgrad(wp, 0)); });
            const double t2 = time_us([&] { OK(launch_wgrad_reduce(wp, grad, 0, 0)); });
            // against the reference kernel (relative to the largest gradient entry: the sums run over up to 2e5 products)
            const int n_w = c.Cout * c.Cin * c.k * c.k;
            hipLaunchKernelGGL(wgrad_ref_kernel, dim3(cdiv(n_w, 64)), dim3(64), 0, 0, bufA, bufB, N, c.Hin, c.Win, c.CinT, c.Cin, c.Ho, c.Wo, c.Cout, c.k,
                               c.stride, grad_ref);
            CK(hipDeviceSynchronize());
            std::vector<float> hg(n_w), hr(n_w);
            CK(hipMemcpy(hg.data(), grad, (size_t)n_w * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(hr.data(), grad_ref, (size_t)n_w * 4, hipMemcpyDeviceToHost));
            double dmax = 0.0, rmax = 0.0;
            for (int i = 0; i < n_w; ++i) { dmax = fmax(dmax, fabs((double)hg[i] - (double)hr[i])); rmax = fmax(rmax, fabs((double)hr[i])); }
            const double rel = dmax / (rmax + 1e-30);
            printf("%-20s wgrad   M=%4d N=%3d K=%7d  MTW=%d NTW=%d grid=%4dx%3d lds=%6zu KC=%3d S=%4d partial=%6.2f MB  %7.1f us + reduce %6.1f us  %6.1f TF/s  reldiff=%.1e%s\n",
                   l.name.c_str(), c.k * c.k * c.CinT, c.Cout, N * c.Ho * c.Wo, wp.MTW, wp.NTW, wp.grid_x, wp.grid_y, wp.lds_bytes, wp.a.KC,
                   wp.a.S, wp.partial_floats * 4e-6, t1, t2, 2.0 * macs / (t1 + t2) * 1e-6, rel, rel > 1e-4 ? "  <-- MISMATCH" : "");
        }
    }
    if (mode == "all" || mode == "bn") {
        int lastC = -1, lastH = -1;
        float* small = dev_rand(64 * 1024, 9);
        StatCell* sums;
        CK(hipMalloc(&sums, (8 * 4 * 1024 + (size_t)2048 * 4 * 1024) * sizeof(StatCell)));
        for (auto& l : layers) {
            const ConvShape& c = l.s;
            if (c.Cout == lastC && c.Ho == lastH) continue;
            lastC = c.Cout; lastH = c.Ho;
            const int64_t mpg = (int64_t)(N / groups) * c.Ho * c.Wo;
            BnFwdArgs f;
            memset(&f, 0, sizeof(f));
            f.y = bufA; f.z = bufB; f.res = bufC; f.stats = stats; f.stat_rep_stride = 8 * 2 * 1024; f.gamma = small; f.beta = small + 1024;
            f.running_mean = small + 2048; f.running_var = small + 3072; f.save_mean = small + 4096; f.save_invstd = small + 8192;
            f.m_per_group = mpg; f.G = groups; f.C = c.Cout; f.relu = 1; f.momentum = 0.1f; f.eps = 1e-5f;
            const double tf = time_us([&] { OK(launch_bn_fwd(f, 0)); });
            BnBwdArgs b;
            memset(&b, 0, sizeof(b));
            b.dz = bufA; b.z = bufB; b.m_per_group = mpg; b.G = groups; b.C = c.Cout; b.nsets = 1;
            b.y[0] = bufC; b.mean[0] = small + 4096; b.invstd[0] = small + 8192; b.gamma[0] = small; b.dy[0] = bufD;
            b.dgamma[0] = small + 12288; b.dbeta[0] = small + 13312; b.sums = sums;
            const double tb = time_us([&] {
                CK(hipMemsetAsync(sums, 0, 8 * 4 * 1024 * sizeof(StatCell), 0));
                OK(launch_bn_bwd(b, 0));
            });
            b.barrier = (unsigned*)(sums + 4096);
            b.fsums = sums + 8192;
            const double tfz = time_us([&] {
                CK(hipMemsetAsync(sums, 0, 8 * 4 * 1024 * sizeof(StatCell), 0));
                OK(launch_bn_bwd(b, 0));
            });
            b.barrier = nullptr;
            b.fsums = nullptr;
            const double bytes = (double)N * c.Ho * c.Wo * c.Cout * 4;
            printf("bn C=%3d HW=%2d  one-pass bwd %6.1f us (%.2f TB/s over 4 passes)\n", c.Cout, c.Ho, tfz, 4 * bytes / tfz * 1e-6);
            printf("bn C=%3d HW=%2d  elems=%9.0f  fwd(3 tensors) %6.1f us %5.2f TB/s   bwd(reduce+apply, 7 tensor passes) %6.1f us %5.2f TB/s\n",
                   c.Cout, c.Ho, bytes / 4, tf, 3 * bytes / tf * 1e-6, tb, 7 * bytes / tb * 1e-6);
            if (sweep) {
                bn_bwd_tune(0, 0, 2);
                const double ta = time_us([&] { OK(launch_bn_bwd(b, 0)); });
                printf("    apply only %6.1f us %5.2f TB/s;  reduce only (3 tensors) cap x unroll:", ta, 4 * bytes / ta * 1e-6);
                for (int cap : {128, 256, 512, 1024, 2048})
                    for (int U : {1, 2, 4}) {
                        bn_bwd_tune(cap, U, 1);
                        const double tr = time_us([&] { OK(launch_bn_bwd(b, 0)); });
                        printf(" %dx%d:%.1f", cap, U, tr);
                    }
                printf("\n");
                bn_bwd_tune(0, 0, 0);
                // HBM-cold: rotate through R tensor sets whose footprint exceeds the 256 MB MALL (y, z as in the real step: written long ago)
                {
                    const int R = 12;
                    const size_t el = (size_t)N * c.Ho * c.Wo * c.Cout;
                    static float* pool = nullptr;
                    if (!pool) pool = dev_rand((size_t)R * 4 * max_act, 77);
                    int it = 0;
                    auto setb = [&](bool hot_dz) {
                        float* base = pool + (size_t)(it % R) * 4 * el;
                        b.dz = hot_dz ? bufA : base; b.z = base + el; b.y[0] = base + 2 * el; b.dy[0] = base + 3 * el;
                        f.y = base; f.z = base + el; f.res = base + 2 * el;
                        ++it;
                    };
                    const double tfc = time_us([&] { setb(false); OK(launch_bn_fwd(f, 0)); }, 24, 12);
                    printf("    cold: fwd %.1f us (%.2f TB/s)", tfc, 3 * bytes / tfc * 1e-6);
                    bn_bwd_tune(0, 0, 2);
                    const double tac = time_us([&] { setb(true); OK(launch_bn_bwd(b, 0)); }, 24, 12);
                    printf("  apply %.1f us (%.2f TB/s)  reduce capxU:", tac, 4 * bytes / tac * 1e-6);
                    for (int cap : {512, 1024, 2048, 4096})
                        for (int U : {1, 2, 4}) {
                            bn_bwd_tune(cap, U, 1);
                            const double tr = time_us([&] { setb(true); OK(launch_bn_bwd(b, 0)); }, 24, 12);
                            printf(" %dx%d:%.1f", cap, U, tr);
                        }
                    printf("\n");
                    bn_bwd_tune(0, 0, 0);
                    b.dz = bufA; b.z = bufB; b.y[0] = bufC; b.dy[0] = bufD; f.y = bufA; f.z = bufB; f.res = bufC;
                }
            }
        }
    }
    return 0;
}
