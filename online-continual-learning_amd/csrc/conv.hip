// K1-K4: Reduced-ResNet18 convolution / batch-norm kernels for gfx950.
//
//  conv_gemm_kernel   implicit-GEMM 3x3 / 1x1 convolution on exact-fp32 MFMA (v_mfma_f32_16x16x4_f32).
//                     One generic "tap list + output lattice" geometry covers forward (stride 1/2), data
//                     gradient (stride 1, and stride 2 as four dense parity classes) and the 1x1 shortcut.
//                     The input patch (with halo) of a 64/128-pixel tile is staged ONCE in LDS and reused by
//                     all taps; weights stream through LDS per (tap, channel chunk).  Epilogues: BN batch
//                     statistics (fp64 atomics), folded eval-mode BN, residual, ReLU, masked residual.
//  conv_wgrad_kernel  weight gradient as a (tap,ci) x co GEMM reduced over pixels, split-K over pixel tiles,
//                     partials summed by wgrad_reduce_kernel straight into PyTorch's OIHW gradient.
//  bn_*               train-mode BatchNorm forward (normalise+residual+ReLU, running-stat update) and backward.
//
// Replaces the ATen sequences behind models/resnet.py:10-12,32-37,90-99 and their autograd.
#include "conv.h"
#include <string.h>
#include <algorithm>

namespace ocl {

static const size_t kLdsLimit = 160 * 1024;      // hardware: 160 KiB per workgroup
static const size_t kLdsTarget = 72 * 1024;      // planner target (2 workgroups per CU)

// =====================================================================================================
// implicit-GEMM convolution
// =====================================================================================================
template <int MT, int NT>
__global__ void __launch_bounds__(256) conv_gemm_kernel(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    constexpr int BM = 64 * MT;
    constexpr int BN = 16 * NT;
    double* red = (double*)lds_raw;                 // [4 waves][2][BN]
    int* rowoff = (int*)(red + 8 * BN);             // [BM]
    float* wl = (float*)(rowoff + BM);              // [KC][BNP]
    float* patch = wl + ((a.KC * a.BNP + 3) & ~3);  // [imgs][PR][PC][CP]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.y * BN;

    // ---- tile decode ------------------------------------------------------------------------------
    const int tile = blockIdx.x;
    const int grp = tile / a.tiles_per_group;
    const int tg = tile - grp * a.tiles_per_group;
    const int ti = tg / a.tiles_per_img;
    const int img0 = grp * a.group_size + ti * a.imgs;
    const int p0 = (tg - ti * a.tiles_per_img) * a.ppi;
    const int grp_end = min(a.N, (grp + 1) * a.group_size);
    const int LP = a.LH * a.LW;
    const int ly0 = p0 / a.LW;
    const int pend = min(p0 + a.ppi, LP);
    const int ly1 = (pend - 1) / a.LW;
    const int pr_use = (ly1 - ly0) * a.is + (a.max_dy - a.min_dy) + 1;

    // per-row decode: LDS patch offset of the pixel's origin and output element offset
    auto decode = [&](int r, int& poff, int& ooff) -> bool {
        const int il = r / a.ppi;
        const int pl = r - il * a.ppi;
        const int p = p0 + pl;
        const int n = img0 + il;
        const bool v = (il < a.imgs) && (n < grp_end) && (p < LP);
        const int ly = p / a.LW, lx = p - ly * a.LW;
        poff = v ? ((il * a.PR + (ly - ly0) * a.is) * a.PC + lx * a.is) * a.CP : 0;
        ooff = ((n * a.Hout + ly * a.os + a.oy0) * a.Wout + lx * a.os + a.ox0) * a.Cout;
        return v;
    };
    if (tid < BM) {
        int po, oo;
        const bool v = decode(tid, po, oo);
        rowoff[tid] = v ? oo : -1;
    }
    int abase[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        int po, oo;
        decode(wave * 16 * MT + mt * 16 + r16, po, oo);
        abase[mt] = po + g;
    }
    const int bbase = g * a.BNP + r16;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int kc4 = a.KC >> 2;
    for (int c0 = 0; c0 < a.Cin; c0 += a.KC) {
        __syncthreads();  // previous chunk fully consumed (also publishes rowoff on the first trip)
        // ---- stage the input patch for channels [c0, c0+KC) -----------------------------------------
        const int units = a.imgs * pr_use * a.PC * kc4;
        for (int u = tid; u < units; u += 256) {
            const int c4 = u % kc4;
            const int t1 = u / kc4;
            const int pc = t1 % a.PC;
            const int t2 = t1 / a.PC;
            const int pr = t2 % pr_use;
            const int il = t2 / pr_use;
            const int iy = ly0 * a.is + a.min_dy + pr;
            const int ix = a.min_dx + pc;
            const int n = img0 + il;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (n < grp_end && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win)
                v = *(const float4*)(a.in + ((int64_t)(n * a.Hin + iy) * a.Win + ix) * a.Cin + c0 + c4 * 4);
            float* d = patch + ((il * a.PR + pr) * a.PC + pc) * a.CP + c4 * 4;  // 8-B aligned (CP even)
            *(float2*)d = make_float2(v.x, v.y);
            *(float2*)(d + 2) = make_float2(v.z, v.w);
        }
        for (int t = 0; t < a.ntaps; ++t) {
            if (t > 0) __syncthreads();  // previous tap's weight tile consumed
            // ---- stage W[tap][c0:c0+KC][n0:n0+BN] ---------------------------------------------------
            const float* wsrc = a.w + ((int64_t)a.tw[t] * a.Cin + c0) * a.CoutP + n0;
            constexpr int Q = BN / 4;
            for (int u = tid; u < a.KC * Q; u += 256) {
                const int kc = u / Q, q = u - kc * Q;
                *(float4*)(wl + kc * a.BNP + q * 4) = *(const float4*)(wsrc + (int64_t)kc * a.CoutP + q * 4);
            }
            __syncthreads();
            const float* pa = patch + ((a.tdy[t] - a.min_dy) * a.PC + (a.tdx[t] - a.min_dx)) * a.CP;
            const float* pb = wl + bbase;
            for (int s = 0; s < a.KC; s += 4) {
                float av[MT], bv[NT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) av[mt] = pa[abase[mt] + s];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bv[nt] = pb[s * a.BNP + nt * 16];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt], bv[nt], acc[mt][nt], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: D layout col = lane&15, row = (lane>>4)*4 + reg -----------------------------------
    const int flags = a.flags;
    double s1[NT], s2[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) s1[nt] = s2[nt] = 0.0;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int off = rowoff[wave * 16 * MT + mt * 16 + g * 4 + reg];
            if (off < 0) continue;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int co = n0 + nt * 16 + r16;
                float v = acc[mt][nt][reg];
                if (flags & EPI_STATS) {
                    s1[nt] += (double)v;
                    s2[nt] += (double)v * (double)v;
                }
                if (co < a.Cout) {
                    const int64_t o = (int64_t)off + co;
                    if (flags & EPI_AFFINE) v = fmaf(v, a.scale[co], a.shift[co]);
                    if (flags & EPI_RES) v += a.res[o];
                    if (flags & EPI_RESMASK) v += (a.resmask[o] > 0.f) ? a.res[o] : 0.f;
                    if (flags & EPI_ACCUM) v += a.out[o];
                    if (flags & EPI_RELU) v = fmaxf(v, 0.f);
                    a.out[o] = v;
                }
            }
        }
    }
    if (flags & EPI_STATS) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            double x = s1[nt], y = s2[nt];
            x += __shfl_xor(x, 16, 64);
            x += __shfl_xor(x, 32, 64);
            y += __shfl_xor(y, 16, 64);
            y += __shfl_xor(y, 32, 64);
            if (g == 0) {
                red[(wave * 2 + 0) * BN + nt * 16 + r16] = x;
                red[(wave * 2 + 1) * BN + nt * 16 + r16] = y;
            }
        }
        __syncthreads();
        if (tid < BN) {
            const int co = n0 + tid;
            if (co < a.Cout) {
                const double t1 = red[0 * BN + tid] + red[2 * BN + tid] + red[4 * BN + tid] + red[6 * BN + tid];
                const double t2 = red[1 * BN + tid] + red[3 * BN + tid] + red[5 * BN + tid] + red[7 * BN + tid];
                atomicAdd(&a.stats[((int64_t)grp * 2 + 0) * a.Cout + co], t1);
                atomicAdd(&a.stats[((int64_t)grp * 2 + 1) * a.Cout + co], t2);
            }
        }
    }
}

typedef void (*conv_fn_t)(const ConvArgs);
static conv_fn_t conv_fn(int MT, int NT) {
#define OCL_CASE(M, N) \
    if (MT == M && NT == N) return conv_gemm_kernel<M, N>;
    OCL_CASE(1, 1) OCL_CASE(1, 2) OCL_CASE(1, 3) OCL_CASE(1, 4) OCL_CASE(1, 5)
    OCL_CASE(2, 1) OCL_CASE(2, 2) OCL_CASE(2, 3) OCL_CASE(2, 4) OCL_CASE(2, 5)
#undef OCL_CASE
    return nullptr;
}

static int bnp_for(int bn) {  // LDS weight row stride with (stride mod 32) == 16: B reads conflict-free
    int p = bn;
    while ((p & 31) != 16) p += 16;
    return p;
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
    const int ntiles16 = cdiv(g.Cout, 16);
    int NT = ntiles16 <= 5 ? ntiles16 : 5;
    const int splits = cdiv(ntiles16, NT);
    if (splits > 1) NT = cdiv(ntiles16, splits);  // balance (10 tiles -> 2 x 5)
    a.n_splits = splits;
    a.CoutP = splits * NT * 16;
    a.BNP = bnp_for(NT * 16);
    a.group_size = g.N / g.groups;
    const int LP = g.LH * g.LW;
    const int64_t tiles128 = (int64_t)g.groups * (LP >= 128 ? (int64_t)a.group_size * cdiv(LP, 128)
                                                             : cdiv(a.group_size, std::max(1, 128 / LP)));
    int MT = (tiles128 * splits >= 384) ? 2 : 1;
    for (;; ) {
        const int BM = 64 * MT;
        if (LP >= BM) {
            a.imgs = 1; a.ppi = BM; a.tiles_per_img = cdiv(LP, BM);
        } else {
            a.imgs = std::min(BM / LP, a.group_size); a.ppi = LP; a.tiles_per_img = 1;
        }
        a.tiles_per_group = cdiv(a.group_size, a.imgs) * a.tiles_per_img;
        a.PC = (g.LW - 1) * g.is + (a.max_dx - a.min_dx) + 1;
        int rows_l;
        if (a.imgs == 1 && LP >= BM) rows_l = std::min(g.LH, (BM + g.LW - 2) / g.LW + 1);
        else rows_l = g.LH;
        a.PR = (rows_l - 1) * g.is + (a.max_dy - a.min_dy) + 1;
        // channel chunk: largest divisor of Cin (multiple of 4) whose patch + weights fit the LDS target
        int KC = g.Cin;
        size_t bytes = 0;
        for (;;) {
            a.KC = KC; a.CP = KC + 2;
            bytes = (size_t)8 * NT * 16 * 8 + (size_t)BM * 4 + (size_t)((KC * a.BNP + 3) & ~3) * 4 +
                    (size_t)a.imgs * a.PR * a.PC * a.CP * 4;
            if (bytes <= kLdsTarget) break;
            if (KC % 8 == 0 && g.Cin % (KC / 2) == 0 && KC / 2 >= 4) KC /= 2; else break;
        }
        if (bytes > kLdsLimit - 1024 && a.imgs > 1) {  // shrink the tile (fewer images)
            if (MT == 2) { MT = 1; continue; }
            while (bytes > kLdsLimit - 1024 && a.imgs > 1) {
                a.imgs -= 1;
                a.tiles_per_group = cdiv(a.group_size, a.imgs) * a.tiles_per_img;
                bytes = (size_t)8 * NT * 16 * 8 + (size_t)BM * 4 + (size_t)((a.KC * a.BNP + 3) & ~3) * 4 +
                        (size_t)a.imgs * a.PR * a.PC * a.CP * 4;
            }
        }
        OCL_REQUIRE(bytes <= kLdsLimit - 1024, "plan_conv: tile needs %zu B of LDS (Hin=%d Win=%d Cin=%d)", bytes, g.Hin, g.Win,
                    g.Cin);
        p->lds_bytes = bytes;
        break;
    }
    p->MT = MT; p->NT = NT;
    p->grid_x = g.groups * a.tiles_per_group;
    p->grid_y = splits;
    return OCL_OK;
}

int launch_conv(const ConvPlan& p, hipStream_t s) {
    conv_fn_t fn = conv_fn(p.MT, p.NT);
    if (!fn) {
        set_error("launch_conv: no kernel for MT=%d NT=%d", p.MT, p.NT);
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
template <int MTW, int NTW>
__global__ void __launch_bounds__(256) conv_wgrad_kernel(const WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    int* pixoff = (int*)lds_raw;                    // [KP]
    float* dyt = (float*)(pixoff + a.KP);           // [KP][DP]
    float* patch = dyt + (size_t)a.KP * a.DP;       // [imgs][PR][PC][CP]
    constexpr int BNW = 16 * NTW;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, g = lane >> 4;
    const int by = blockIdx.y;
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

    const int kc4 = a.KC >> 2;
    constexpr int Q = BNW / 4;
    for (int tile = blockIdx.x; tile < a.total_tiles; tile += a.S) {
        const int ti = tile / a.tiles_per_img;
        const int img0 = ti * a.imgs;
        const int p0 = (tile - ti * a.tiles_per_img) * a.ppi;
        const int oy0 = p0 / a.Wo;
        const int pend = min(p0 + a.ppi, LP);
        const int oy1 = (pend - 1) / a.Wo;
        const int pr_use = (oy1 - oy0) * a.stride + (a.max_dy - a.min_dy) + 1;
        __syncthreads();  // previous tile consumed
        // pixel table + dy tile
        for (int u = tid; u < a.KP * Q; u += 256) {
            const int q = u / Q, c4 = u - q * Q;
            const int il = q / a.ppi, pl = q - il * a.ppi;
            const int p = p0 + pl, n = img0 + il;
            const bool v = (il < a.imgs) && (n < a.N) && (p < LP);
            const int oy = p / a.Wo, ox = p - oy * a.Wo;
            if (c4 == 0) pixoff[q] = v ? ((il * a.PR + (oy - oy0) * a.stride) * a.PC + ox * a.stride) * a.CP : 0;
            float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
            const int co = n0 + c4 * 4;
            if (v && co < a.Cout) d = *(const float4*)(a.dy + ((int64_t)(n * a.Ho + oy) * a.Wo + ox) * a.Cout + co);
            *(float4*)(dyt + (size_t)q * a.DP + c4 * 4) = d;
        }
        // input patch, channels [c0, c0+KC)
        const int units = a.imgs * pr_use * a.PC * kc4;
        for (int u = tid; u < units; u += 256) {
            const int c4 = u % kc4;
            const int u1 = u / kc4;
            const int pc = u1 % a.PC;
            const int u2 = u1 / a.PC;
            const int pr = u2 % pr_use;
            const int il = u2 / pr_use;
            const int iy = oy0 * a.stride + a.min_dy + pr;
            const int ix = a.min_dx + pc;
            const int n = img0 + il;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (n < a.N && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win)
                v = *(const float4*)(a.x + ((int64_t)(n * a.Hin + iy) * a.Win + ix) * a.Cin + c0 + c4 * 4);
            float* d = patch + ((il * a.PR + pr) * a.PC + pc) * a.CP + c4 * 4;
            *(float2*)d = make_float2(v.x, v.y);
            *(float2*)(d + 2) = make_float2(v.z, v.w);
        }
        __syncthreads();
        const float* pb = dyt + (size_t)g * a.DP + r16;
        for (int s = 0; s < a.KP; s += 4) {
            const int po = pixoff[s + g];
            float av[MTW], bv[NTW];
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt) av[mt] = patch[po + aoff[mt]];
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) bv[nt] = pb[(size_t)s * a.DP + nt * 16];
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt], bv[nt], acc[mt][nt], 0, 0, 0);
        }
    }
    // partial tile out: rows (chunk, mblock, m), cols co
    const int mrows_chunk = a.mblocks_per_chunk * 64 * MTW;
    float* dst = a.partial + (int64_t)blockIdx.x * a.Mrows_total * a.CoutP;
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int row = chunk * mrows_chunk + m0 + wave * 16 * MTW + mt * 16 + g * 4 + reg;
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) dst[(int64_t)row * a.CoutP + n0 + nt * 16 + r16] = acc[mt][nt][reg];
        }
}

typedef void (*wgrad_fn_t)(const WgradArgs);
static wgrad_fn_t wgrad_fn(int M, int N) {
#define OCL_CASE(A, B) \
    if (M == A && N == B) return conv_wgrad_kernel<A, B>;
    OCL_CASE(1, 1) OCL_CASE(1, 2) OCL_CASE(1, 3) OCL_CASE(1, 4) OCL_CASE(1, 5)
    OCL_CASE(2, 1) OCL_CASE(2, 2) OCL_CASE(2, 3) OCL_CASE(2, 4) OCL_CASE(2, 5)
    OCL_CASE(3, 1) OCL_CASE(3, 2) OCL_CASE(3, 3) OCL_CASE(3, 4) OCL_CASE(3, 5)
    OCL_CASE(4, 1) OCL_CASE(4, 2) OCL_CASE(4, 3) OCL_CASE(4, 4) OCL_CASE(4, 5)
#undef OCL_CASE
    return nullptr;
}

// sums the split-K partials into the OIHW gradient: grad[co][ci][t] (+)= sum_s partial[s][(chunk,t,cc)][co]
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float* __restrict__ partial, int S, int Mrows_total, int CoutP,
                                                           int mrows_chunk, int KC, int ntaps, int CinReal, int Cout,
                                                           float* __restrict__ grad, int accumulate) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;  // (t, ci, co) with co fastest
    const int total = ntaps * CinReal * Cout;
    if (idx >= total) return;
    const int co = idx % Cout;
    const int r = idx / Cout;
    const int ci = r % CinReal, t = r / CinReal;
    const int chunk = ci / KC, cc = ci - chunk * KC;
    const int row = chunk * mrows_chunk + t * KC + cc;
    const float* p = partial + (int64_t)row * CoutP + co;
    const int64_t stride = (int64_t)Mrows_total * CoutP;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int s = 0;
    for (; s + 4 <= S; s += 4) {
        s0 += p[(int64_t)s * stride];
        s1 += p[(int64_t)(s + 1) * stride];
        s2 += p[(int64_t)(s + 2) * stride];
        s3 += p[(int64_t)(s + 3) * stride];
    }
    for (; s < S; ++s) s0 += p[(int64_t)s * stride];
    float v = (s0 + s1) + (s2 + s3);
    float* gp = grad + ((int64_t)co * CinReal + ci) * ntaps + t;
    if (accumulate) v += *gp;
    *gp = v;
}

static int wg_cp(int kc, int stride) {  // LDS pixel stride for the wgrad A reads
    int cp = kc;
    if (stride == 1) { while ((cp & 31) != 16) cp += 2; }
    else { while ((cp & 15) != 8) cp += 2; }
    return cp;
}

int plan_wgrad(int N, int Hin, int Win, int Cin, int Ho, int Wo, int Cout, int ksize, int stride, WgradPlan* p) {
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
    int NTW = ntile <= 5 ? ntile : 5;
    a.nblocks = cdiv(ntile, NTW);
    if (a.nblocks > 1) NTW = cdiv(ntile, a.nblocks);
    a.CoutP = a.nblocks * NTW * 16;
    int dp = NTW * 16;
    while ((dp & 31) != 16) dp += 16;
    a.DP = dp;
    const int LP = Ho * Wo;
    int KPmax = 128;
    for (;;) {
        if (LP >= KPmax) {
            a.imgs = 1; a.ppi = KPmax; a.tiles_per_img = cdiv(LP, KPmax); a.KP = KPmax;
        } else {
            a.imgs = std::min(KPmax / LP, N); a.ppi = LP; a.tiles_per_img = 1; a.KP = (int)round_up((int64_t)a.imgs * LP, 4);
        }
        a.PC = (Wo - 1) * stride + (a.max_dx - a.min_dx) + 1;
        int rows_l = (a.imgs == 1 && LP >= KPmax) ? std::min(Ho, (KPmax + Wo - 2) / Wo + 1) : Ho;
        a.PR = (rows_l - 1) * stride + (a.max_dy - a.min_dy) + 1;
        int KC = Cin;
        size_t bytes = 0;
        for (;;) {
            a.KC = KC; a.CP = wg_cp(KC, stride);
            bytes = (size_t)a.KP * 4 + (size_t)a.KP * a.DP * 4 + (size_t)a.imgs * a.PR * a.PC * a.CP * 4;
            if (bytes <= kLdsTarget) break;
            if (KC % 8 == 0 && Cin % (KC / 2) == 0 && KC / 2 >= 4) KC /= 2; else break;
        }
        if (bytes > kLdsLimit - 1024) {
            if (KPmax > 32) { KPmax /= 2; continue; }
            set_error("plan_wgrad: tile needs %zu B of LDS", bytes);
            return OCL_ERR_ARG;
        }
        p->lds_bytes = bytes;
        break;
    }
    a.nchunks = Cin / a.KC;
    a.Mchunk = a.ntaps * a.KC;
    const int mtiles = cdiv(a.Mchunk, 16);
    int MTW = std::min(4, cdiv(mtiles, 4));
    if (MTW * NTW > 20) MTW = std::max(1, 20 / NTW);
    a.mblocks_per_chunk = cdiv(mtiles, 4 * MTW);
    a.Mrows_total = a.nchunks * a.mblocks_per_chunk * 64 * MTW;
    a.total_tiles = cdiv(N, a.imgs) * a.tiles_per_img;
    const int by = a.nchunks * a.mblocks_per_chunk * a.nblocks;
    a.S = std::max(1, std::min(a.total_tiles, std::max(1, 768 / by)));
    p->MTW = MTW; p->NTW = NTW;
    p->grid_x = a.S; p->grid_y = by;
    p->partial_floats = (size_t)a.S * a.Mrows_total * a.CoutP;
    return OCL_OK;
}

int launch_wgrad(const WgradPlan& p, hipStream_t s) {
    wgrad_fn_t fn = wgrad_fn(p.MTW, p.NTW);
    if (!fn) {
        set_error("launch_wgrad: no kernel for MTW=%d NTW=%d", p.MTW, p.NTW);
        return OCL_ERR_STATE;
    }
    ProfScope ps(PROF_WGRAD, s);
    hipLaunchKernelGGL(fn, dim3(p.grid_x, p.grid_y), dim3(256), p.lds_bytes, s, p.a);
    OCL_LAUNCH_CHECK();
    return OCL_OK;
}

int launch_wgrad_reduce(const WgradPlan& p, float* grad_oihw, int accumulate, hipStream_t s) {
    const WgradArgs& a = p.a;
    const int cin_real = a.Cin == 4 ? 3 : a.Cin;  // the stem's NHWC4 input carries a zero 4th channel
    const int total = a.ntaps * cin_real * a.Cout;
    ProfScope ps(PROF_WGRAD, s);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, a.partial, a.S, a.Mrows_total, a.CoutP,
                       a.mblocks_per_chunk * 64 * p.MTW, a.KC, a.ntaps, cin_real, a.Cout, grad_oihw, accumulate);
    OCL_LAUNCH_CHECK();
    return OCL_OK;
}

// =====================================================================================================
// weight packing (all conv layers in one launch)
// =====================================================================================================
__global__ void __launch_bounds__(256) pack_weights_kernel(const float* __restrict__ params, float* __restrict__ arena,
                                                           const PackDesc* __restrict__ descs) {
    const PackDesc d = descs[blockIdx.y];
    const int total = d.Cout * d.Cin * d.ntaps;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int co = e / (d.Cin * d.ntaps);
        const int rem = e - co * d.Cin * d.ntaps;
        const int ci = rem / d.ntaps, t = rem - ci * d.ntaps;
        const float v = params[d.w_off + e];
        if (d.f_off >= 0) arena[d.f_off + ((int64_t)t * d.CinP + ci) * d.CoutP + co] = v;
        if (d.d_off >= 0) arena[d.d_off + ((int64_t)t * d.Cout + co) * d.CiP + ci] = v;
    }
}

int launch_pack_weights(const float* params, float* arena, const PackDesc* descs_dev, int n_layers, int max_elems, hipStream_t s) {
    ProfScope ps(PROF_BN, s);
    hipLaunchKernelGGL(pack_weights_kernel, dim3(std::min(64, cdiv(max_elems, 256)), n_layers), dim3(256), 0, s, params, arena,
                       descs_dev);
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
        const double s1 = a.stats[((int64_t)g * 2 + 0) * a.C + c], s2 = a.stats[((int64_t)g * 2 + 1) * a.C + c];
        const double mean = s1 / M;
        double var = s2 / M - mean * mean;
        if (var < 0.0) var = 0.0;
        const double invstd = 1.0 / sqrt(var + (double)a.eps);
        const float scale = a.gamma[c] * (float)invstd;
        sc[c] = scale;
        sh[c] = a.beta[c] - (float)mean * scale;
        if (blockIdx.x == 0) {
            a.save_mean[(int64_t)g * a.C + c] = (float)mean;
            a.save_invstd[(int64_t)g * a.C + c] = (float)invstd;
        }
    }
    if (blockIdx.x == 0 && g == 0 && a.running_mean) {
        for (int c = tid; c < a.C; c += 256) {
            float rm = a.running_mean[c], rv = a.running_var[c];
            for (int gg = 0; gg < a.G; ++gg) {  // one update per group, in order (= separate forward calls)
                const double s1 = a.stats[((int64_t)gg * 2 + 0) * a.C + c], s2 = a.stats[((int64_t)gg * 2 + 1) * a.C + c];
                const double mean = s1 / M;
                double var = s2 / M - mean * mean;
                if (var < 0.0) var = 0.0;
                const double unb = M > 1.0 ? var * M / (M - 1.0) : var;
                rm = a.momentum * (float)mean + (1.f - a.momentum) * rm;
                rv = a.momentum * (float)unb + (1.f - a.momentum) * rv;
            }
            a.running_mean[c] = rm;
            a.running_var[c] = rv;
        }
        if (tid == 0 && a.nbt) *a.nbt += a.G;
    }
    __syncthreads();
    const int C4 = a.C >> 2;
    const int64_t units = a.m_per_group * C4;
    const float4* y4 = (const float4*)a.y + (int64_t)g * units;
    const float4* r4 = a.res ? (const float4*)a.res + (int64_t)g * units : nullptr;
    float4* z4 = (float4*)a.z + (int64_t)g * units;
    for (int64_t u = (int64_t)blockIdx.x * 256 + tid; u < units; u += (int64_t)gridDim.x * 256) {
        const int c = (int)(u % C4) * 4;
        float4 v = y4[u];
        v.x = fmaf(v.x, sc[c], sh[c]);
        v.y = fmaf(v.y, sc[c + 1], sh[c + 1]);
        v.z = fmaf(v.z, sc[c + 2], sh[c + 2]);
        v.w = fmaf(v.w, sc[c + 3], sh[c + 3]);
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
        for (int64_t p = pbeg + pl; p < pend; p += PT) {
            const int64_t e = ((int64_t)g * M + p) * C4 + c4;
            float4 d = ((const float4*)a.dz)[e];
            if (a.z) {
                const float4 zz = ((const float4*)a.z)[e];
                d.x = zz.x > 0.f ? d.x : 0.f; d.y = zz.y > 0.f ? d.y : 0.f;
                d.z = zz.z > 0.f ? d.z : 0.f; d.w = zz.w > 0.f ? d.w : 0.f;
            }
#pragma unroll
            for (int k = 0; k < 2; ++k)
                if (k < a.nsets) {
                    const float4 y = ((const float4*)a.y[k])[e];
                    sd[k].x += d.x; sd[k].y += d.y; sd[k].z += d.z; sd[k].w += d.w;
                    sx[k].x = fmaf(d.x, (y.x - mean[k].x) * istd[k].x, sx[k].x);
                    sx[k].y = fmaf(d.y, (y.y - mean[k].y) * istd[k].y, sx[k].y);
                    sx[k].z = fmaf(d.z, (y.z - mean[k].z) * istd[k].z, sx[k].z);
                    sx[k].w = fmaf(d.w, (y.w - mean[k].w) * istd[k].w, sx[k].w);
                }
        }
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
        atomicAdd(&a.sums[(((int64_t)k * a.G + g) * 2 + which) * a.C + c], t);
    }
}

__global__ void __launch_bounds__(256) bn_bwd_apply_kernel(const BnBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    // per set: k1[C] (mean dpre), k2[C] (mean dpre*xhat), scale[C], mean[C], invstd[C]
    const int g = blockIdx.y, tid = threadIdx.x;
    const double Md = (double)a.m_per_group;
    for (int j = tid; j < a.nsets * a.C; j += 256) {
        const int c = j % a.C, k = j / a.C;
        const double sdy = a.sums[(((int64_t)k * a.G + g) * 2 + 0) * a.C + c];
        const double sdx = a.sums[(((int64_t)k * a.G + g) * 2 + 1) * a.C + c];
        float* s = sm + (size_t)k * 5 * a.C;
        const float istd = a.invstd[k][(int64_t)g * a.C + c];
        s[c] = (float)(sdy / Md);
        s[a.C + c] = (float)(sdx / Md);
        s[2 * a.C + c] = a.gamma[k][c] * istd;
        s[3 * a.C + c] = a.mean[k][(int64_t)g * a.C + c];
        s[4 * a.C + c] = istd;
        if (blockIdx.x == 0 && g == 0) {
            double dg = 0.0, db = 0.0;
            for (int gg = 0; gg < a.G; ++gg) {
                db += a.sums[(((int64_t)k * a.G + gg) * 2 + 0) * a.C + c];
                dg += a.sums[(((int64_t)k * a.G + gg) * 2 + 1) * a.C + c];
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
        }
#pragma unroll
        for (int k = 0; k < 2; ++k)
            if (k < a.nsets) {
                const float* s = sm + (size_t)k * 5 * a.C;
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

int launch_bn_bwd(const BnBwdArgs& a, hipStream_t s) {
    OCL_REQUIRE(a.nsets == 1 || a.nsets == 2, "bn_bwd: nsets=%d", a.nsets);
    const int C4 = a.C / 4, PT = 256 / C4;
    const int64_t per_block_pixels = (int64_t)PT * 32;
    const int bx = (int)std::max<int64_t>(1, std::min<int64_t>(512, (a.m_per_group + per_block_pixels - 1) / per_block_pixels));
    ProfScope ps(PROF_BN, s);
    const size_t sm1 = (size_t)a.nsets * 2 * PT * a.C * 4;
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(bx, a.G), dim3(256), sm1, s, a);
    OCL_LAUNCH_CHECK();
    const int64_t units = a.m_per_group * C4;
    const int bx2 = (int)std::max<int64_t>(1, std::min<int64_t>(1024, (units + 1023) / 1024));
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(bx2, a.G), dim3(256), (size_t)a.nsets * 5 * a.C * 4, s, a);
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
                                                        int d) {
    const int n = blockIdx.x, lane = threadIdx.x;
    const float* p = v + (int64_t)n * d;
    float ss = 0.f;
    for (int j = lane; j < d; j += 64) ss = fmaf(p[j], p[j], ss);
    ss = wave_sum(ss);
    const float nrm = fmaxf(sqrtf(ss), 1e-12f);  // F.normalize eps
    if (lane == 0) norms[n] = nrm;
    for (int j = lane; j < d; j += 64) out[(int64_t)n * d + j] = p[j] / nrm;
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
int launch_l2norm_fwd(const float* v, float* out, float* norms, int n, int d, hipStream_t s) {
    ProfScope ps(PROF_HEAD, s);
    hipLaunchKernelGGL(l2norm_fwd_kernel, dim3(n), dim3(64), 0, s, v, out, norms, d);
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

__global__ void __launch_bounds__(256) colsum_kernel(const float* __restrict__ m, int rows, int cols, float* __restrict__ out,
                                                     int accumulate) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    float s = 0.f;
    for (int r = 0; r < rows; ++r) s += m[(int64_t)r * cols + c];
    out[c] = accumulate ? out[c] + s : s;
}
int launch_colsum(const float* m, int rows, int cols, float* out, int accumulate, hipStream_t s) {
    ProfScope ps(PROF_HEAD, s);
    hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(cols, 256)), dim3(256), 0, s, m, rows, cols, out, accumulate);
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
    for (int m = 1; m <= 2; ++m)
        for (int n = 1; n <= 5; ++n)
            OCL_HIP(hipFuncSetAttribute((const void*)conv_fn(m, n), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit));
    for (int m = 1; m <= 4; ++m)
        for (int n = 1; n <= 5; ++n)
            OCL_HIP(hipFuncSetAttribute((const void*)wgrad_fn(m, n), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit));
    done = true;
    return OCL_OK;
}

}  // namespace ocl
