// conv_w_kernel: the implicit-GEMM convolution of conv_t_kernel (same K-grouped operands, same D[channel][pixel] register tiles, same register
// epilogue) restructured around what the round-5 profiles showed: conv_t_kernel / conv_q_kernel run ONE dependent phase chain per workgroup
// (barrier -> patch store -> barrier -> K loop -> epilogue) with one wave per SIMD, or two that start together and stay in step, so every
// phase that is not the K loop -- 20 - 40 % of a workgroup's lifetime (profiles/r6_conv_phase_trace.txt) -- leaves the MFMA pipe idle.
//
// Here the workgroup is eight waves (two per SIMD) that share ONLY the weights (resident in LDS, copied once by LDS-DMA):
//  * a wave owns its pixel tiles (16 NT consecutive lattice pixels x 16 MT channels) from staging to stores: it stages the tile's input
//    patch (with halo) into its PRIVATE LDS region -- by buffer_load ... lds straight from global memory where no input transform applies
//    (no staging registers, nothing to commit), through registers where the producer's BatchNorm + ReLU is folded in (ConvArgs::xf) --
//    runs the K loop over it and stores from registers;
//  * no workgroup barrier exists between the prologue and the statistics flush: the two waves of a SIMD drift apart by themselves (the
//    older wave wins the arbitration) and one's staging / epilogue runs under the other's MFMAs;
//  * the next tile's patch is requested right after the K loop, under the epilogue's stores;
//  * tiles are dealt per SIMD pair (waves w, w + 4) in contiguous ranges: neighbouring tiles share halo rows in L1 / L2.
// Replaces, where its planner takes a launch, the ATen sequence behind models/resnet.py:10-12,32-37 (conv + train-mode BatchNorm statistics /
// folded eval-mode BatchNorm + residual + ReLU) and the data-gradient half of its autograd, exactly as conv_t_kernel does.
#include "conv_stats_dev.h"
#include <string.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#include <type_traits>

namespace ocl {

constexpr int kWNU = 16;       // 1-KiB staging pieces (64 lanes x 16 bytes) of a wave's patch, at most
constexpr int kWWaves = 4;     // waves per workgroup: one per SIMD
constexpr int kWOob = (int)0x80010000;   // staging offset of a unit that must read zeros: patch origin (> -64 KB, < 1 GB) + this is past every descriptor made by make_rsrc

// unit word of the staging table (one 16-byte unit of the patch per lane and piece): global offset from the patch origin in 16-byte units
// (13 bits) | channel quad (6) << 13 | patch row (4) << 19 | patch column (6) << 23 | image in tile (2) << 29; -1: padding slot (zeros)
__host__ __device__ constexpr int wunit_pack(int go16, int c4, int pr, int pc, int il) { return go16 | (c4 << 13) | (pr << 19) | (pc << 23) | (il << 29); }

// 16 bytes per lane global -> LDS without registers (buffer_load_dwordx4 ... lds: the LDS address is lds_dst + lane * 16, lds_dst wave-uniform).
// A plain device function on purpose: with the builtin written inside a (generic) lambda of a kernel template hipcc dropped the whole kernel
// from the host object -- no diagnostic, undefined symbols at link time.
__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t rs, float* lds_dst, int voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds_dst, 16, voff, 0, 0, 0);
}

#ifndef OCL_CW_FAKE_B
#define OCL_CW_FAKE_B 0   // measurement only: 1 = every round reads the B operand at the pixel's origin (no table look-up, no address arithmetic in the K loop; WRONG results)
#endif
#ifndef OCL_CW_SGB
#define OCL_CW_SGB 1   // K loop: operand reads of the next round interleaved one per MFMA of the current round (0: reads in one block in front of the MFMAs)
#endif

// What the round-6 measurements say about two waves on a SIMD (profiles/r6_convw_two_waves.txt): while one wave streams MFMAs its partner's
// other instructions issue at about one per MFMA (25 - 32 cycles each) and every one of them costs the streaming wave ~13 cycles: a staging
// or epilogue phase of 300 instructions that takes ~1 k cycles alone takes ~9 k beside a K loop, and the K loop beside it runs at 60 - 70 %.
// Work that is not an MFMA is therefore hidden in the shadow of the SAME wave's MFMAs (its own instructions between two of its MFMAs
// are free up to about five per gap), not under another wave's:
//  * one wave per SIMD (four per workgroup, one workgroup per CU), every wave owns its pixel tiles from staging to stores, no workgroup
//    barrier between the prologue and the statistics flush;
//  * the wave's patch region is TWO buffers: the patch of item j + 2 is requested (buffer_load ... lds, no registers) right after the K loop
//    of item j into the buffer that loop has just released, so a request has a whole K loop to land and the wave never waits for memory:
//    the s_waitcnt in front of the request finds everything older complete;
//  * an item is (pixel tile, channel chunk): with more than one chunk the accumulators persist over a tile's chunks (layer 4: 160 input
//    channels in chunks of 40 keep weights of a channel split + eight patch buffers inside the LDS).
// XF: instantiated for the input transform (ConvArgs::xf set: staging through registers, one item in flight in registers, written to LDS
// after the next K loop); the other instantiations stage by LDS-DMA only and carry no staging registers.
// TRACE (measurement build, kbench KBENCH_TRACE): s_memtime stamps of lane 0 of every wave, 32 slots per wave: start | requests issued |
// tables built | barrier passed | per item: K loop done, request + epilogue issued | ... | slot 31: end
template <int MT, int NT, bool BNB, bool XF, bool TRACE = false>
__global__ void __launch_bounds__(256, 1) conv_w_kernel(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    constexpr int COPW = 16 * MT;
    constexpr int NACC = (MT * NT == 1) ? 2 : 1;   // a single register tile: two accumulators (k-steps alternate) break the dependent MFMA chain
    const int nch = a.Cin / a.KC;                                // channel chunks (wave-uniform)
    int* qoff = (int*)lds_raw;                                   // [Qpad] patch offset (floats) of group q relative to a pixel's origin
    float* wl = (float*)(qoff + a.Qpad);                         // [nch][Qpad][COPW][4] resident weights of this channel split
    float* xft = wl + (size_t)nch * a.Qpad * COPW * 4;           // input transform: [groups][Cin/4][2][4] scale quads / shift quads
    const float* bnt = xft + (a.bnb_lds > 0 ? a.bnb_lds : 0);    // EPI_BNB: [groups][Cout/4][3][4]
    float* patch0 = (float*)(lds_raw + a.qstat_off);             // [4 waves][2][patch_floats]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, g = lane >> 4;
    float* patch = patch0 + (size_t)wave * 2 * a.patch_floats;
    const int n0 = blockIdx.y * COPW;
    const int flags = BNB ? a.flags : (a.flags & ~EPI_BNB);
    const int* __restrict__ blob = a.blob;
    const int nu = a.nstage;
    int tr_n = 0;
    auto stamp = [&]() __attribute__((always_inline)) {
        if constexpr (TRACE) {
            if (lane == 0 && tr_n < 31) a.trace[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * kWWaves + wave) * 32 + tr_n] = __builtin_amdgcn_s_memtime();
            ++tr_n;
        }
    };
    stamp();
    // ---- this wave's tile range: contiguous (neighbouring tiles share halo rows in L1 / L2) --------------------------------------------------
    const int T = (a.N / a.imgs) * a.tiles_per_img;
    int bx = blockIdx.x;
    if ((gridDim.x & 7) == 0) bx = (bx & 7) * (gridDim.x >> 3) + (bx >> 3);   // (block b runs on XCD b % 8: neighbouring ranges on one L2)
    const int nwv = gridDim.x * kWWaves, wv = bx * kWWaves + wave;
    const int t_begin = (int)(((int64_t)wv * T) / nwv), t_end = (int)(((int64_t)(wv + 1) * T) / nwv);
    const int n_items = (t_end - t_begin) * nch;
    // ---- per-lane constants: staging units, output pixels ------------------------------------------------------------------------------------
    int uw[kWNU];
#pragma unroll
    for (int i = 0; i < kWNU; ++i) uw[i] = i < nu ? blob[a.off_pu + i * 64 + lane] : -1;
    int loc_p[NT], loc_o[NT], loc_il[NT];
    {
        const int* lc = blob + a.off_loc + r16;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            loc_p[nt] = lc[(3 * nt + 0) * 16];
            loc_o[nt] = lc[(3 * nt + 1) * 16];
            loc_il[nt] = lc[(3 * nt + 2) * 16];
        }
    }
    const int qtab = tid < a.Qpad ? blob[16 + tid] : 0;
    const __amdgpu_buffer_rsrc_t rs_in = make_rsrc(a.in), rs_w = make_rsrc(a.wT);
    // ---- tile geometry (wave-uniform) ------------------------------------------------------------------------------------------------------
    int in_base = 0, iy0 = 0, ix0 = 0, obase = 0, nimg = 0, grp = 0;
    auto decode = [&](int t) __attribute__((always_inline)) {
        int tp, lx0;
        const int ti = mdiv(t, a.m_tpi, a.tiles_per_img, tp);
        const int img0 = ti * a.imgs;
        const int ly0 = mdiv(tp * a.ppi, a.m_lw, a.LW, lx0);
        iy0 = ly0 * a.is + a.min_dy;
        ix0 = lx0 * a.is + a.min_dx;
        in_base = (((img0 * a.Hin + iy0) * a.Win + ix0) * a.Cin) * 4;
        obase = ((img0 * a.Hout + ly0 * a.os + a.oy0) * a.Wout + lx0 * a.os + a.ox0) * a.Cout;
        nimg = min(a.imgs, a.N - img0);
        int rem;
        grp = mdiv(img0, a.m_tpg, a.group_size, rem);
    };
    // ---- staging --------------------------------------------------------------------------------------------------------------------------------
    // the request cursor runs two items ahead of the K loops (three with XF: one item sits in registers)
    int rq_t = t_begin, rq_c = 0, rq_n = 0;
    auto unit_off = [&](int w, int base, bool& ok) __attribute__((always_inline)) -> int {
        const int upr = (w >> 19) & 15, upc = (w >> 23) & 63, uil = (w >> 29) & 3;
        ok = (w >= 0) & ((unsigned)(iy0 + upr) < (unsigned)a.Hin) & ((unsigned)(ix0 + upc) < (unsigned)a.Win) & (uil < nimg);
        return ok ? base + ((w & 0x1fff) << 4) : kOob;
    };
    // tiles whose patch rows all lie inside the image (and whose columns are the plan's: a.aligned == 1, tiles of whole lattice rows / whole
    // images): a unit's offset is the tile's origin + a per-lane constant, the out-of-image columns and the padding slots folded into it
    int gv[kWNU];
#pragma unroll
    for (int i = 0; i < kWNU; ++i) {
        const int w = uw[i];
        const bool cok = (w >= 0) & ((unsigned)(a.min_dx + ((w >> 23) & 63)) < (unsigned)a.Win);
        gv[i] = cok ? ((w & 0x1fff) << 4) : kWOob;
    }
    auto rq_advance = [&]() __attribute__((always_inline)) {
        ++rq_n;
        if (++rq_c >= nch) { rq_c = 0; ++rq_t; }
    };
    // one LDS-DMA instruction per 1-KiB piece, out-of-image / padding units read zeros
    auto request_dma = [&]() __attribute__((always_inline)) {
        decode(rq_t);
        const int base = in_base + rq_c * a.KC * 4;
        float* dst = patch + (size_t)(rq_n & 1) * a.patch_floats;
        const bool inner = (a.aligned == 1) & (iy0 >= 0) & (iy0 + a.PR <= a.Hin) & (nimg == a.imgs);   // wave-uniform
        if (inner) {
#pragma unroll
            for (int i = 0; i < kWNU; ++i)
                if (i < nu) lds_dma16(rs_in, dst + i * 256, base + gv[i]);
        } else {
#pragma unroll
            for (int i = 0; i < kWNU; ++i)
                if (i < nu) {
                    bool ok;
                    const int off = unit_off(uw[i], base, ok);
                    lds_dma16(rs_in, dst + i * 256, off);
                }
        }
        rq_advance();
    };
    // staging through registers (input transform): loads now, transform + LDS store after the next K loop
    float4 pv[XF ? kWNU : 1];
    unsigned okm = 0;
    int pv_grp = 0, pv_c4 = 0, pv_buf = 0;
    auto xf_load = [&]() __attribute__((always_inline)) {
        decode(rq_t);
        const int base = in_base + rq_c * a.KC * 4;
        okm = 0;
        pv_grp = grp;
        pv_c4 = rq_c * (a.KC >> 2);
        pv_buf = rq_n & 1;
#pragma unroll
        for (int i = 0; i < kWNU; ++i)
            if (i < nu) {
                bool ok;
                const int off = unit_off(uw[i], base, ok);
                pv[XF ? i : 0] = buf_load16(rs_in, off);
                okm |= ok ? (1u << i) : 0u;
            }
        rq_advance();
    };
    auto xf_store = [&]() __attribute__((always_inline)) {
        const float* tb = xft + (size_t)(pv_grp * a.C4tot + pv_c4) * 8;
        float* dst = patch + (size_t)pv_buf * a.patch_floats;
#pragma unroll
        for (int i = 0; i < kWNU; ++i)
            if (i < nu) {
                const float* t = tb + ((uw[i] >> 13) & 63) * 8;
                const float4 sc = *(const float4*)t, sh = *(const float4*)(t + 4);
                float4 v = pv[XF ? i : 0];
                v.x = fmaxf(__fmaf_rn(v.x, sc.x, sh.x), 0.f); v.y = fmaxf(__fmaf_rn(v.y, sc.y, sh.y), 0.f);
                v.z = fmaxf(__fmaf_rn(v.z, sc.z, sh.z), 0.f); v.w = fmaxf(__fmaf_rn(v.w, sc.w, sh.w), 0.f);
                if (!((okm >> i) & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f);
                *(float4*)(dst + (size_t)(i * 64 + lane) * 4) = v;
            }
    };
    // ---- first requests: item 0 (and, by DMA, item 1) in front of the weights -- the K loop needs them first -----------------------------------
    if (XF) {
        if (rq_n < n_items) xf_load();
    } else {
        if (rq_n < n_items) request_dma();
        if (rq_n < n_items) request_dma();
    }
    // ---- weights: global -> LDS without registers, every wave its share.  The pack row of group q is arithmetic (tap of the group -> its index
    // in the pack), not a table look-up: no load stands between the kernel's start and the DMA requests
    {
        const int wcol_ok = a.WPT - n0;
        const int kc4 = a.KC >> 2;
        const int upc = a.Qpad * COPW;              // units per chunk (a multiple of 64)
        const int npieces = (nch * upc) >> 6;
        for (int pc0 = wave; pc0 < npieces; pc0 += kWWaves) {
            const int u0 = pc0 * 64;
            const int u = u0 + lane;
            const int ch = u / upc, ul = u - ch * upc;
            const int q = ul / COPW, c = ul - q * COPW;
            int c4;
            const int tq = mdiv(q, a.m_kc4, kc4, c4);
            // (a.d_c4 / a.d_pc: the pack index of every tap as nibbles -- tap_sel(a.tw, tq) with a per-lane tq became an indexed LOAD from the
            // kernel-argument segment followed by s_waitcnt vmcnt(0): every DMA request waited for the one before it)
            const int twq = tq < 8 ? (int)(((unsigned)a.d_c4 >> (4 * tq)) & 15u) : a.d_pc;
            const int row = twq * a.C4tot + ch * kc4 + c4;
            const int off = (q < a.Qc && c < wcol_ok) ? ((row * a.WPT + n0 + c) * 4) * 4 : kOob;
            lds_dma16(rs_w, wl + (size_t)u0 * 4, off);
        }
    }
    stamp();   // requests issued
    if (XF) {   // the producer's BatchNorm folded into scale / shift per (group, channel); see ConvArgs::xf (as conv_t_kernel)
        const int C = a.Cin;
        const double M = (double)a.xf_m_per_group;
        const bool lead = blockIdx.x == 0 && blockIdx.y == 0;
        for (int j = tid; j < a.groups * C; j += 256) {
            const int gq = j / C, c = j - gq * C;
            double mean, var;
            bn_batch_moments(a.xf_stats, a.xf_rep_stride, gq, c, C, M, a.xf_eps, mean, var);
            const double xv = var + (double)a.xf_eps;
            double invstd = (double)rsqrtf((float)xv);
            invstd = invstd * (1.5 - 0.5 * xv * invstd * invstd);
            invstd = invstd * (1.5 - 0.5 * xv * invstd * invstd);
            float sc, sh;
            bn_scale_shift(a.xf_gamma[c], a.xf_beta[c], (float)mean, (float)invstd, sc, sh);
            float* t = xft + (size_t)(gq * (C >> 2) + (c >> 2)) * 8 + (c & 3);
            t[0] = sc;
            t[4] = sh;
            if (lead) {
                a.xf_save_mean[j] = (float)mean;
                a.xf_save_invstd[j] = (float)invstd;
            }
        }
        if (lead && a.xf_running_mean)
            bn_running_update(a.xf_stats, a.xf_rep_stride, a.groups, C, M, a.xf_momentum, a.xf_eps, a.xf_running_mean, a.xf_running_var, a.xf_nbt, tid, 256);
    }
    if (BNB && (flags & EPI_BNB)) bnb_table(a, const_cast<float*>(bnt), tid, 256);
    if (tid < a.Qpad) qoff[tid] = qtab;
    stamp();   // tables built
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // own weight DMA (and the first patches by DMA) landed
    __syncthreads();                                   // weights, group table, transform tables visible to every wave
    stamp();   // barrier passed
    if (XF) {   // items 0 and 1 into the two buffers (the only exposed staging of the wave), item 2 into the registers
        if (n_items > 0) xf_store();
        if (rq_n < n_items) { xf_load(); xf_store(); }
        if (rq_n < n_items) xf_load();
    }

    float s1[MT][4], s2[MT][4];   // BatchNorm partial sums of this lane's channels over this wave's tiles
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int e = 0; e < 4; ++e) s1[mt][e] = s2[mt][e] = 0.f;
    int run_grp = -1;
    // a wave whose range crosses into the next BatchNorm group (at most one wave per group boundary and launch) adds the finished group's
    // sums straight to the accumulators; the group a wave ends in goes through the workgroup's flush below
    auto flush_direct = [&]() __attribute__((always_inline)) {
        StatCell* st_ = a.stats + (int64_t)((blockIdx.x * kWWaves + wave) % kStatReps) * a.stat_rep_stride;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x = row16_sum(s1[mt][e]), y = row16_sum(s2[mt][e]);
                const int co = n0 + mt * 16 + 4 * g + e;
                if (r16 == 0 && co < a.Cout) {
                    fx_add(&st_[((int64_t)run_grp * 2 + 0) * a.Cout + co], (double)x);
                    fx_add(&st_[((int64_t)run_grp * 2 + 1) * a.Cout + co], (double)y);
                }
                s1[mt][e] = s2[mt][e] = 0.f;
            }
    };

    const int nr = a.Qpad >> 2;
    f32x4 acc[NACC][MT][NT];
    int cur_t = t_begin, cur_c = 0;
    for (int j = 0; j < n_items; ++j) {
        const float* pb = patch + (size_t)(j & 1) * a.patch_floats;
        const float* wb = wl + (size_t)cur_c * a.Qpad * COPW * 4 + (size_t)(g * COPW + r16) * 4;
        if (cur_c == 0) {
#pragma unroll
            for (int s = 0; s < NACC; ++s)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[s][mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        int pbase[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) pbase[nt] = loc_p[nt];   // (pixels of images past the batch's end read a stale patch region: their results are never stored)
        {   // K loop: operands of round rho + 1 are read while the MFMAs of round rho issue (two register sets), as conv_t_kernel's resident form
            float4 bv[2][NT], av[2][MT];
            int fR = 0;
            int po = qoff[g], po1 = qoff[4 * min(1, nr - 1) + g];
            auto fetch = [&](int set) __attribute__((always_inline)) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bv[set][nt] = *(const float4*)(pb + pbase[nt] + (OCL_CW_FAKE_B ? 0 : po));
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) av[set][mt] = *(const float4*)(wb + (size_t)fR * 4 * COPW * 4 + mt * 64);
                ++fR;
                if (!OCL_CW_FAKE_B) {
                    po = po1;
                    po1 = qoff[4 * min(fR + 1, nr - 1) + g];
                }
            };
            auto fma4 = [&](int set) __attribute__((always_inline)) {
#define OCL_KSTEP(E, J)                                                                                                           \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                           \
        acc[J % NACC][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[set][mt].E, bv[set][nt].E, acc[J % NACC][mt][nt], 0, 0, 0);
                OCL_KSTEP(x, 0) OCL_KSTEP(y, 1) OCL_KSTEP(z, 0) OCL_KSTEP(w, 1)   // (a single register tile: consecutive MFMAs alternate between two accumulators)
#undef OCL_KSTEP
            };
            // one other instruction (operand read, table read, address add) behind every MFMA: they issue in the MFMA's shadow instead of
            // as a block in front of the round
            auto spread = [&]() __attribute__((always_inline)) {
#pragma unroll
                for (int i = 0; i < 4 * MT * NT; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                       // MFMA
                    __builtin_amdgcn_sched_group_barrier(0x002 | 0x004 | 0x100, 1, 0);       // VALU | SALU | DS read
                }
            };
            fetch(0);
            int rho = 0;
            for (; rho + 2 <= nr; rho += 2) {
                fetch(1);
                if (!OCL_CW_SGB) __builtin_amdgcn_sched_barrier(0);
                fma4(0);
                if (OCL_CW_SGB) spread();
                __builtin_amdgcn_sched_barrier(0);
                fetch(0);
                if (!OCL_CW_SGB) __builtin_amdgcn_sched_barrier(0);
                fma4(1);
                if (OCL_CW_SGB) spread();
                __builtin_amdgcn_sched_barrier(0);
            }
            if (rho < nr) fma4(0);
        }
        stamp();   // K loop done
        // ---- the patch of item j + 2 into the buffer this K loop has released.  Everything older than this point -- the request of item
        // j + 1 (a whole K loop ago) and the stores of the epilogue before it -- has long completed: the wait is a formality that makes the
        // guarantee explicit (the next K loop reads buffer (j + 1) & 1 without another wait)
        if (XF) {
            if (j + 2 < n_items) xf_store();          // item j + 2 (in registers since the last iteration) -> buffer j & 1
            if (rq_n < n_items) xf_load();            // item j + 3 -> registers
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (rq_n < n_items) request_dma();        // item j + 2 -> buffer j & 1
        }
        const bool last_chunk = cur_c + 1 >= nch;   // wave-uniform
        if (last_chunk) {
            decode(cur_t);
            const int t_grp = grp;
            if ((flags & (EPI_STATS | EPI_BNB)) && t_grp != run_grp) {   // wave-uniform
                if (run_grp >= 0) flush_direct();
                run_grp = t_grp;
            }
            int ooff[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) ooff[nt] = loc_il[nt] < nimg ? obase + loc_o[nt] : -1;
            if (NACC == 2) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[0][mt][nt] += acc[NACC - 1][mt][nt];
            }
            // ---- epilogue from registers: lane (r16 = pixel, g) holds channels n0 + mt*16 + 4g .. +3 of its NT pixels (as conv_t_kernel) -------------
            if (flags == EPI_STATS || flags == 0) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const bool pv_ok = ooff[nt] >= 0;
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const int co = n0 + mt * 16 + 4 * g;
                        if (pv_ok && co < a.Cout) {
                            const float4 v = make_float4(acc[0][mt][nt][0], acc[0][mt][nt][1], acc[0][mt][nt][2], acc[0][mt][nt][3]);
                            s1[mt][0] += v.x; s1[mt][1] += v.y; s1[mt][2] += v.z; s1[mt][3] += v.w;
                            s2[mt][0] = fmaf(v.x, v.x, s2[mt][0]); s2[mt][1] = fmaf(v.y, v.y, s2[mt][1]);
                            s2[mt][2] = fmaf(v.z, v.z, s2[mt][2]); s2[mt][3] = fmaf(v.w, v.w, s2[mt][3]);
                            *(float4*)(a.out + (int64_t)ooff[nt] + co) = v;
                        }
                    }
                }
            } else
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const bool pv_ok = ooff[nt] >= 0;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int co = n0 + mt * 16 + 4 * g;
                    if (!pv_ok || co >= a.Cout) continue;
                    float4 v = make_float4(acc[0][mt][nt][0], acc[0][mt][nt][1], acc[0][mt][nt][2], acc[0][mt][nt][3]);
                    if (flags & EPI_STATS) {
                        s1[mt][0] += v.x; s1[mt][1] += v.y; s1[mt][2] += v.z; s1[mt][3] += v.w;
                        s2[mt][0] = fmaf(v.x, v.x, s2[mt][0]); s2[mt][1] = fmaf(v.y, v.y, s2[mt][1]);
                        s2[mt][2] = fmaf(v.z, v.z, s2[mt][2]); s2[mt][3] = fmaf(v.w, v.w, s2[mt][3]);
                    }
                    float* op = a.out + (int64_t)ooff[nt] + co;
                    if (flags & EPI_AFFINE) {
                        const float4 sc = *(const float4*)(a.scale + co), sh = *(const float4*)(a.shift + co);
                        v.x = fmaf(v.x, sc.x, sh.x); v.y = fmaf(v.y, sc.y, sh.y); v.z = fmaf(v.z, sc.z, sh.z); v.w = fmaf(v.w, sc.w, sh.w);
                    }
                    if (flags & EPI_RES) {
                        const float4 r = *(const float4*)(a.res + (int64_t)ooff[nt] + co);
                        v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
                    }
                    if (flags & EPI_RESMASK) {
                        const float4 r = *(const float4*)(a.res + (int64_t)ooff[nt] + co);
                        const float4 m = *(const float4*)(a.resmask + (int64_t)ooff[nt] + co);
                        v.x += m.x > 0.f ? r.x : 0.f; v.y += m.y > 0.f ? r.y : 0.f; v.z += m.z > 0.f ? r.z : 0.f; v.w += m.w > 0.f ? r.w : 0.f;
                    }
                    if (flags & EPI_ACCUM) {
                        const float4 o = *(const float4*)op;
                        v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                    }
                    if (BNB && (flags & EPI_BNB)) {   // ReLU mask + the two batch sums of the BatchNorm this gradient enters (ConvArgs::bnb_*)
                        const int64_t eo = (int64_t)ooff[nt] + co;
                        const float* tq = bnt + (size_t)(t_grp * (a.Cout >> 2) + (co >> 2)) * 12;
                        bnb_apply(a, *(const float4*)tq, *(const float4*)(tq + 4), *(const float4*)(tq + 8), eo, v, s1[mt], s2[mt]);
                    }
                    if (flags & EPI_RELU) {
                        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                    }
                    *(float4*)op = v;
                }
            }
            cur_c = 0;
            ++cur_t;
        } else {
            ++cur_c;
        }
        stamp();   // request + epilogue issued
    }
    // ---- statistics: every wave leaves the sums of the group it ended in at the start of its own patch region; after the workgroup's only
    // other barrier, one thread per (sum, channel) adds the four waves' values in the order of their tile ranges (fixed: deterministic per
    // workgroup) and issues one accumulation per group present
    if (flags & (EPI_STATS | EPI_BNB)) {
        double* slot = (double*)patch;   // [2][COPW] doubles, then the group id
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x = row16_sum(s1[mt][e]), y = row16_sum(s2[mt][e]);
                if (r16 == 0) {
                    slot[mt * 16 + 4 * g + e] = (double)x;
                    slot[COPW + mt * 16 + 4 * g + e] = (double)y;
                }
            }
        if (lane == 0) *(int*)(slot + 2 * COPW) = run_grp;
        __syncthreads();
        if (tid < 2 * COPW) {
            const int which = tid / COPW, c = tid - which * COPW;
            const int co = n0 + c;
            if (co < a.Cout) {
                StatCell* st_ = a.stats + (int64_t)(blockIdx.x % kStatReps) * a.stat_rep_stride;
                double accd = 0.0;
                int cur = -1;
#pragma unroll
                for (int w = 0; w < kWWaves; ++w) {   // ranges in ascending order
                    const double* sl = (const double*)(patch0 + (size_t)w * 2 * a.patch_floats);
                    const int gw = *(const int*)(sl + 2 * COPW);
                    if (gw < 0) continue;
                    if (gw != cur) {
                        if (cur >= 0) fx_add(&st_[((int64_t)cur * 2 + which) * a.Cout + co], accd);
                        accd = 0.0;
                        cur = gw;
                    }
                    accd += sl[which * COPW + c];
                }
                if (cur >= 0) fx_add(&st_[((int64_t)cur * 2 + which) * a.Cout + co], accd);
            }
        }
    }
    if constexpr (TRACE) {
        if (lane == 0) a.trace[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * kWWaves + wave) * 32 + 31] = __builtin_amdgcn_s_memtime();
    }
}

// =====================================================================================================================================================
// conv_wx_kernel: conv_w_kernel specialised on the number of K rounds (NR) and of staging pieces (NU), for the launches that carry a
// training pass.  What the specialisation buys (profiles/r6_mfma_shadow_probe.txt, r6_convw_phases.txt: v_mfma_f32_16x16x4_f32 runs on the
// f32 vector ALUs -- a VALU instruction between two MFMAs of the same wave costs the stream 13 - 19 cycles, an s_add nothing, a
// ds_read_b128 nothing up to one per four MFMAs; a buffer_load ... lds costs the issuing wave 160 - 200 cycles wherever it stands):
//  * the K loop is NR unrolled rounds without a single VALU instruction of its own: the B operand's LDS address of every (buffer, pixel tile,
//    round) is a register computed once per launch (it does not depend on the tile: every tile of a plan has the same shape), the A
//    operand's is one register + an immediate;
//  * the patch of item j + 1 is requested during the K loop of item j, one piece every few rounds, into the LDS buffer that loop does not
//    read (buffer_load ... lds; a burst of requests runs into the CU's fill rate -- ~14 - 20 bytes per cycle and CU,
//    profiles/r6_mfma_shadow_probe.txt -- and stalls the in-order wave, MFMAs included); voffset = a per-lane constant + the tile's shift
//    (one VALU add per piece), the rows outside the image clipped by a per-IMAGE buffer descriptor (negative and past-the-end offsets read
//    zeros).  With the input transform (XF) the pieces travel through registers, two items ahead: at its round piece i of item j + 1
//    (loaded one K loop ago) is transformed and stored, then its registers take piece i of item j + 2;
//  * then the epilogue's stores; then the next K loop starts without a wait (its patch was written by this wave's own, ordered, LDS stores).
// The weights go the same way (registers, all waves, batches of 24 pieces) in the prologue.
// Requirements (planner): tiles of whole lattice rows of ONE image (imgs == 1, aligned == 1), Qpad == 4 NR, nstage == NU.
__host__ __device__ constexpr int wx_slot_round(int i, int NR, int NU) { return 1 + (i * (NR - 2)) / NU; }   // piece i's round: evenly over the K loop
template <int MT, int NT, int NR, int NU, bool BNB, bool XF, bool TRACE = false>
__global__ void __launch_bounds__(256, 1) conv_wx_kernel(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    typedef __attribute__((address_space(3))) const f32x4* lds_f4;   // (a native vector: loads through an address-space pointer)
    constexpr int COPW = 16 * MT;
    constexpr int NACC = (MT * NT == 1) ? 2 : 1;
    static_assert(wx_slot_round(NU - 1, NR, NU) < NR, "every piece has its round");
    const int nch = a.Cin / a.KC;
    int* qoff = (int*)lds_raw;
    float* wl = (float*)(qoff + a.Qpad);
    float* xft = wl + (size_t)nch * a.Qpad * COPW * 4;
    const float* bnt = xft + (a.bnb_lds > 0 ? a.bnb_lds : 0);
    float* patch0 = (float*)(lds_raw + a.qstat_off);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, g = lane >> 4;
    float* patch = patch0 + (size_t)wave * 2 * a.patch_floats;
    const int n0 = blockIdx.y * COPW;
    const int flags = BNB ? a.flags : (a.flags & ~EPI_BNB);
    const int* __restrict__ blob = a.blob;
    int tr_n = 0;
    auto stamp = [&]() __attribute__((always_inline)) {
        if constexpr (TRACE) {
            if (lane == 0 && tr_n < 31) a.trace[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * kWWaves + wave) * 32 + tr_n] = __builtin_amdgcn_s_memtime();
            ++tr_n;
        }
    };
    stamp();
    const int T = a.N * a.tiles_per_img;
    int bx = blockIdx.x;
    if ((gridDim.x & 7) == 0) bx = (bx & 7) * (gridDim.x >> 3) + (bx >> 3);
    const int nwv = gridDim.x * kWWaves, wv = bx * kWWaves + wave;
    const int t_begin = (int)(((int64_t)wv * T) / nwv), t_end = (int)(((int64_t)(wv + 1) * T) / nwv);
    const int n_items = (t_end - t_begin) * nch;
    // ---- per-lane constants ---------------------------------------------------------------------------------------------------------------------
    int gv[NU];   // staging offset of the lane's unit of piece i from the patch origin; padding slots and out-of-image columns: past every descriptor
#pragma unroll
    for (int i = 0; i < NU; ++i) {
        const int w = blob[a.off_pu + i * 64 + lane];
        const bool cok = (w >= 0) & ((unsigned)(a.min_dx + ((w >> 23) & 63)) < (unsigned)a.Win);
        gv[i] = cok ? ((w & 0x1fff) << 4) : kWOob;
    }
    int c4u[XF ? NU : 1];   // (input transform: the unit's channel quad)
    if (XF) {
#pragma unroll
        for (int i = 0; i < NU; ++i) c4u[XF ? i : 0] = (blob[a.off_pu + i * 64 + lane] >> 13) & 63;
    }
    int loc_p[NT], loc_o[NT];
    {
        const int* lc = blob + a.off_loc + r16;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            loc_p[nt] = lc[(3 * nt + 0) * 16];
            loc_o[nt] = lc[(3 * nt + 1) * 16];
        }
    }
    const int qtab = tid < a.Qpad ? blob[16 + tid] : 0;
    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(a.wT);
    const int img_bytes = a.Hin * a.Win * a.Cin * 4;
    // ---- the request of an item: geometry (wave-uniform), then NU pieces --------------------------------------------------------------------------
    int rq_t = t_begin, rq_c = 0, rq_n = 0;
    int rq_shift = 0x40000000;                    // (no item: every unit out of range)
    int rq_img = 0;                               // image of the request: its descriptor is rebuilt from scalars where it is used
    int rq_grp = 0;
    auto rq_prepare = [&]() __attribute__((always_inline)) {   // the item the cursor points at (or the all-zero request past the last item)
        // (straight-line selects: conditional stores to the captured variables were turned into a dynamically indexed stack slot -- scratch
        // traffic and an s_waitcnt vmcnt(0) in front of every request)
        const bool has = rq_n < n_items;
        int tp, lxr, rem;
        const int img = mdiv(has ? rq_t : t_begin, a.m_tpi, a.tiles_per_img, tp);
        const int ly0 = mdiv(tp * a.ppi, a.m_lw, a.LW, lxr);   // (whole rows: the remainder is 0)
        const int iy0 = ly0 * a.is + a.min_dy;
        const int shift = ((iy0 * a.Win + a.min_dx) * a.Cin + rq_c * a.KC) * 4;
        rq_shift = has ? shift : 0x40000000;
        rq_img = has ? img : 0;
        rq_grp = mdiv(rq_img, a.m_tpg, a.group_size, rem);
    };
    auto rq_advance = [&]() __attribute__((always_inline)) {
        ++rq_n;
        if (++rq_c >= nch) { rq_c = 0; ++rq_t; }
    };
    float4 pv[XF ? NU : 1];
    int pv_grp = 0, pv_c4 = 0;   // (input transform) BatchNorm group / first channel quad of the item whose pieces are being STORED
    unsigned okm = 0;            // (input transform) bit i: piece i in the registers lies inside the image
    auto piece_load = [&](int i) __attribute__((always_inline)) {   // piece i of the prepared item -> registers
        const __amdgpu_buffer_rsrc_t rq_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in) + (size_t)rq_img * (size_t)(img_bytes >> 2), 0, img_bytes, 0x00020000);
        const int voff = gv[i] + rq_shift;
        pv[XF ? i : 0] = buf_load16(rq_rs, voff);
        // (input transform: a unit outside the image must stay zero through it: its offset is out of the descriptor's range)
        if (XF) okm = (okm & ~(1u << i)) | (((unsigned)voff < (unsigned)img_bytes) ? (1u << i) : 0u);
    };
    // piece I of the prepared item -> buffer DST, without registers
#define OCL_PIECE_DMA(I, DST)                                                                                                                         \
    lds_dma16(__builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in) + (size_t)rq_img * (size_t)(img_bytes >> 2), 0, img_bytes, 0x00020000),     \
              (DST) + (I) * 256, gv[I] + rq_shift)
    auto piece_store = [&](int i, float* dst) __attribute__((always_inline)) {   // registers -> buffer dst (XF: through the producer's BatchNorm + ReLU)
        float4 v = pv[XF ? i : 0];
        if (XF) {
            const float* t = xft + (size_t)(pv_grp * a.C4tot + pv_c4 + c4u[XF ? i : 0]) * 8;
            const float4 sc = *(const float4*)t, sh = *(const float4*)(t + 4);
            v.x = fmaxf(__fmaf_rn(v.x, sc.x, sh.x), 0.f); v.y = fmaxf(__fmaf_rn(v.y, sc.y, sh.y), 0.f);
            v.z = fmaxf(__fmaf_rn(v.z, sc.z, sh.z), 0.f); v.w = fmaxf(__fmaf_rn(v.w, sc.w, sh.w), 0.f);
            if (!((okm >> i) & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        *(float4*)(dst + (size_t)(i * 64 + lane) * 4) = v;
    };
    // ---- item 0's patch, then the weights: all through registers (every wave its share of the weights, 24 pieces per batch) ------------------------
    rq_prepare();
    pv_grp = rq_grp; pv_c4 = rq_c * (a.KC >> 2);
#pragma unroll
    for (int i = 0; i < NU; ++i) {
        if (XF) piece_load(i);
        else OCL_PIECE_DMA(i, patch);
    }
    rq_advance();
    {   // weights: global -> LDS by DMA, every wave its share.  a.d_row == 1 (one chunk, the taps' pack order = their K order: every forward and
        // stride-1 data gradient): the pack row of group q is q, two multiplies per piece; else the general arithmetic
        const int wcol_ok = a.WPT - n0;
        const int kc4 = a.KC >> 2;
        const int upc = a.Qpad * COPW;
        const int npieces = (nch * upc) >> 6;
        if (a.d_row == 1) {
            for (int pc0 = wave; pc0 < npieces; pc0 += kWWaves) {
                const int u = pc0 * 64 + lane;
                const int q = u / COPW, c = u - q * COPW;
                const int off = (q < a.Qc && c < wcol_ok) ? (q * a.WPT + n0 + c) * 16 : kOob;
                lds_dma16(rs_w, wl + (size_t)pc0 * 256, off);
            }
        } else {
            for (int pc0 = wave; pc0 < npieces; pc0 += kWWaves) {
                const int u = pc0 * 64 + lane;
                const int ch = u / upc, ul = u - ch * upc;
                const int q = ul / COPW, c = ul - q * COPW;
                int c4;
                const int tq = mdiv(q, a.m_kc4, kc4, c4);
                const int twq = tq < 8 ? (int)(((unsigned)a.d_c4 >> (4 * tq)) & 15u) : a.d_pc;
                const int row = twq * a.C4tot + ch * kc4 + c4;
                const int off = (q < a.Qc && c < wcol_ok) ? ((row * a.WPT + n0 + c) * 4) * 4 : kOob;
                lds_dma16(rs_w, wl + (size_t)pc0 * 256, off);
            }
        }
    }
    stamp();   // requests issued
    if (XF) {
        const int C = a.Cin;
        const double M = (double)a.xf_m_per_group;
        const bool lead = blockIdx.x == 0 && blockIdx.y == 0;
        for (int j = tid; j < a.groups * C; j += 256) {
            const int gq = j / C, c = j - gq * C;
            double mean, var;
            bn_batch_moments(a.xf_stats, a.xf_rep_stride, gq, c, C, M, a.xf_eps, mean, var);
            const double xv = var + (double)a.xf_eps;
            double invstd = (double)rsqrtf((float)xv);
            invstd = invstd * (1.5 - 0.5 * xv * invstd * invstd);
            invstd = invstd * (1.5 - 0.5 * xv * invstd * invstd);
            float sc, sh;
            bn_scale_shift(a.xf_gamma[c], a.xf_beta[c], (float)mean, (float)invstd, sc, sh);
            float* t = xft + (size_t)(gq * (C >> 2) + (c >> 2)) * 8 + (c & 3);
            t[0] = sc;
            t[4] = sh;
            if (lead) {
                a.xf_save_mean[j] = (float)mean;
                a.xf_save_invstd[j] = (float)invstd;
            }
        }
        if (lead && a.xf_running_mean)
            bn_running_update(a.xf_stats, a.xf_rep_stride, a.groups, C, M, a.xf_momentum, a.xf_eps, a.xf_running_mean, a.xf_running_var, a.xf_nbt, tid, 256);
    }
    if (BNB && (flags & EPI_BNB)) bnb_table(a, const_cast<float*>(bnt), tid, 256);
    if (tid < a.Qpad) qoff[tid] = qtab;
    stamp();   // tables built
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    stamp();   // barrier passed
    int nx_grp = 0, nx_c4 = 0;
    if (XF) {   // input transform: patches through registers, two items ahead (item 0 -> buffer 0 now: the wave's only exposed staging; item 1 -> registers)
#pragma unroll
        for (int i = 0; i < NU; ++i) piece_store(i, patch);
        rq_prepare();
        nx_grp = rq_grp; nx_c4 = rq_c * (a.KC >> 2);
#pragma unroll
        for (int i = 0; i < NU; ++i) piece_load(i);
        rq_advance();
    }
    // ---- the B operand's LDS byte address of every (buffer, pixel tile, round): register constants of the launch -------------------------------------
    unsigned ba[2][NT][NR];
    {
        const unsigned pb0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)patch;
#pragma unroll
        for (int rho = 0; rho < NR; ++rho) {
            const int po = qoff[4 * rho + g];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                ba[0][nt][rho] = pb0 + (unsigned)(loc_p[nt] + po) * 4u;
                ba[1][nt][rho] = ba[0][nt][rho] + (unsigned)a.patch_floats * 4u;
            }
        }
    }
    float s1[MT][4], s2[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int e = 0; e < 4; ++e) s1[mt][e] = s2[mt][e] = 0.f;
    int run_grp = -1;
    auto flush_direct = [&]() __attribute__((always_inline)) {
        StatCell* st_ = a.stats + (int64_t)((blockIdx.x * kWWaves + wave) % kStatReps) * a.stat_rep_stride;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x = row16_sum(s1[mt][e]), y = row16_sum(s2[mt][e]);
                const int co = n0 + mt * 16 + 4 * g + e;
                if (r16 == 0 && co < a.Cout) {
                    fx_add(&st_[((int64_t)run_grp * 2 + 0) * a.Cout + co], (double)x);
                    fx_add(&st_[((int64_t)run_grp * 2 + 1) * a.Cout + co], (double)y);
                }
                s1[mt][e] = s2[mt][e] = 0.f;
            }
    };

    f32x4 acc[NACC][MT][NT];
    int cur_t = t_begin, cur_c = 0;
    // one item: K loop over buffer B (compile-time), the next item's pieces requested in its rounds, the epilogue behind the last chunk
    auto item = [&](auto BT) __attribute__((always_inline)) {
        constexpr int B = decltype(BT)::value;
        float* nxt = patch + (size_t)(1 - B) * a.patch_floats;
        const float* wb = wl + (size_t)cur_c * a.Qpad * COPW * 4 + (size_t)(g * COPW + r16) * 4;
        if (cur_c == 0) {
#pragma unroll
            for (int s = 0; s < NACC; ++s)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[s][mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        pv_grp = nx_grp; pv_c4 = nx_c4;   // (XF) the item in the registers (j + 1)
        rq_prepare();                     // the item requested during this K loop: j + 1 by DMA, j + 2 through registers (wave-uniform arithmetic)
        nx_grp = rq_grp; nx_c4 = rq_c * (a.KC >> 2);
        {
            f32x4 bv[2][NT], av[2][MT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bv[0][nt] = *(lds_f4)(size_t)ba[B][nt][0];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) av[0][mt] = *(const f32x4*)(wb + mt * 64);
#pragma unroll
            for (int rho = 0; rho < NR; ++rho) {
                const int cs = rho & 1, ns = cs ^ 1;
                if (rho + 1 < NR) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) bv[ns][nt] = *(lds_f4)(size_t)ba[B][nt][rho + 1];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) av[ns][mt] = *(const f32x4*)(wb + (size_t)(rho + 1) * 4 * COPW * 4 + mt * 64);
                }
#pragma unroll
                for (int i = 0; i < NU; ++i)
                    if (wx_slot_round(i, NR, NU) == rho) {
                        if (XF) {
                            piece_store(i, nxt);   // item j + 1 -> the buffer this loop does not read
                            piece_load(i);         // item j + 2 -> the register just freed
                        } else {
                            OCL_PIECE_DMA(i, nxt);   // item j + 1 -> the buffer this loop does not read
                        }
                    }
                if constexpr (TRACE) { if (rho == NR / 3 + 1 || rho == NR - NR / 3 - 1) stamp(); }   // (thirds of the K loop)
#define OCL_KSTEP(E, J)                                                                                                           \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                           \
        acc[J % NACC][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cs][mt].E, bv[cs][nt].E, acc[J % NACC][mt][nt], 0, 0, 0);
                OCL_KSTEP(x, 0) OCL_KSTEP(y, 1) OCL_KSTEP(z, 0) OCL_KSTEP(w, 1)
#undef OCL_KSTEP
#pragma unroll
                for (int i = 0; i < 4 * MT * NT; ++i) {   // one other instruction behind every MFMA
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002 | 0x004 | 0x020 | 0x100 | 0x200, 1, 0);   // VALU | SALU | VMEM read | DS read | DS write
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        stamp();   // K loop done
        // DMA: everything outstanding (this loop's requests, the stores of the epilogue before it) is at least a few rounds old; the wait
        // makes the next K loop's reads of the other buffer legal
        if (!XF) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        rq_advance();
        const bool last_chunk = cur_c + 1 >= nch;
        if (last_chunk) {
            int tp, rem;
            const int img = mdiv(cur_t, a.m_tpi, a.tiles_per_img, tp);
            int lxr;
            const int ly0 = mdiv(tp * a.ppi, a.m_lw, a.LW, lxr);
            const int obase = ((img * a.Hout + ly0 * a.os + a.oy0) * a.Wout + a.ox0) * a.Cout;
            const int t_grp = mdiv(img, a.m_tpg, a.group_size, rem);
            if ((flags & (EPI_STATS | EPI_BNB)) && t_grp != run_grp) {
                if (run_grp >= 0) flush_direct();
                run_grp = t_grp;
            }
            int ooff[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) ooff[nt] = obase + loc_o[nt];
            if (NACC == 2) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[0][mt][nt] += acc[NACC - 1][mt][nt];
            }
            if (flags == EPI_STATS || flags == 0) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const int co = n0 + mt * 16 + 4 * g;
                        if (co < a.Cout) {
                            const float4 v = make_float4(acc[0][mt][nt][0], acc[0][mt][nt][1], acc[0][mt][nt][2], acc[0][mt][nt][3]);
                            s1[mt][0] += v.x; s1[mt][1] += v.y; s1[mt][2] += v.z; s1[mt][3] += v.w;
                            s2[mt][0] = fmaf(v.x, v.x, s2[mt][0]); s2[mt][1] = fmaf(v.y, v.y, s2[mt][1]);
                            s2[mt][2] = fmaf(v.z, v.z, s2[mt][2]); s2[mt][3] = fmaf(v.w, v.w, s2[mt][3]);
                            *(float4*)(a.out + (int64_t)ooff[nt] + co) = v;
                        }
                    }
                }
            } else
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int co = n0 + mt * 16 + 4 * g;
                    if (co >= a.Cout) continue;
                    float4 v = make_float4(acc[0][mt][nt][0], acc[0][mt][nt][1], acc[0][mt][nt][2], acc[0][mt][nt][3]);
                    if (flags & EPI_STATS) {
                        s1[mt][0] += v.x; s1[mt][1] += v.y; s1[mt][2] += v.z; s1[mt][3] += v.w;
                        s2[mt][0] = fmaf(v.x, v.x, s2[mt][0]); s2[mt][1] = fmaf(v.y, v.y, s2[mt][1]);
                        s2[mt][2] = fmaf(v.z, v.z, s2[mt][2]); s2[mt][3] = fmaf(v.w, v.w, s2[mt][3]);
                    }
                    float* op = a.out + (int64_t)ooff[nt] + co;
                    if (flags & EPI_AFFINE) {
                        const float4 sc = *(const float4*)(a.scale + co), sh = *(const float4*)(a.shift + co);
                        v.x = fmaf(v.x, sc.x, sh.x); v.y = fmaf(v.y, sc.y, sh.y); v.z = fmaf(v.z, sc.z, sh.z); v.w = fmaf(v.w, sc.w, sh.w);
                    }
                    if (flags & EPI_RES) {
                        const float4 r = *(const float4*)(a.res + (int64_t)ooff[nt] + co);
                        v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
                    }
                    if (flags & EPI_RESMASK) {
                        const float4 r = *(const float4*)(a.res + (int64_t)ooff[nt] + co);
                        const float4 m = *(const float4*)(a.resmask + (int64_t)ooff[nt] + co);
                        v.x += m.x > 0.f ? r.x : 0.f; v.y += m.y > 0.f ? r.y : 0.f; v.z += m.z > 0.f ? r.z : 0.f; v.w += m.w > 0.f ? r.w : 0.f;
                    }
                    if (flags & EPI_ACCUM) {
                        const float4 o = *(const float4*)op;
                        v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                    }
                    if (BNB && (flags & EPI_BNB)) {
                        const int64_t eo = (int64_t)ooff[nt] + co;
                        const float* tq = bnt + (size_t)(t_grp * (a.Cout >> 2) + (co >> 2)) * 12;
                        bnb_apply(a, *(const float4*)tq, *(const float4*)(tq + 4), *(const float4*)(tq + 8), eo, v, s1[mt], s2[mt]);
                    }
                    if (flags & EPI_RELU) {
                        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                    }
                    *(float4*)op = v;
                }
            }
            cur_c = 0;
            ++cur_t;
        } else {
            ++cur_c;
        }
        stamp();   // epilogue issued
    };
    for (int j = 0; j < n_items; j += 2) {
        item(std::integral_constant<int, 0>());
        if (j + 1 < n_items) item(std::integral_constant<int, 1>());
    }
    if (flags & (EPI_STATS | EPI_BNB)) {
        double* slot = (double*)patch;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x = row16_sum(s1[mt][e]), y = row16_sum(s2[mt][e]);
                if (r16 == 0) {
                    slot[mt * 16 + 4 * g + e] = (double)x;
                    slot[COPW + mt * 16 + 4 * g + e] = (double)y;
                }
            }
        if (lane == 0) *(int*)(slot + 2 * COPW) = run_grp;
        __syncthreads();
        if (tid < 2 * COPW) {
            const int which = tid / COPW, c = tid - which * COPW;
            const int co = n0 + c;
            if (co < a.Cout) {
                StatCell* st_ = a.stats + (int64_t)(blockIdx.x % kStatReps) * a.stat_rep_stride;
                double accd = 0.0;
                int cur = -1;
#pragma unroll
                for (int w = 0; w < kWWaves; ++w) {
                    const double* sl = (const double*)(patch0 + (size_t)w * 2 * a.patch_floats);
                    const int gw = *(const int*)(sl + 2 * COPW);
                    if (gw < 0) continue;
                    if (gw != cur) {
                        if (cur >= 0) fx_add(&st_[((int64_t)cur * 2 + which) * a.Cout + co], accd);
                        accd = 0.0;
                        cur = gw;
                    }
                    accd += sl[which * COPW + c];
                }
                if (cur >= 0) fx_add(&st_[((int64_t)cur * 2 + which) * a.Cout + co], accd);
            }
        }
    }
    if constexpr (TRACE) {
        if (lane == 0) a.trace[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * kWWaves + wave) * 32 + 31] = __builtin_amdgcn_s_memtime();
    }
}

#undef OCL_PIECE_DMA

typedef void (*convw_fn_t)(const ConvArgs);
static convw_fn_t convw_fn(int MT, int NT, int bnb, int xf) {   // (the EPI_BNB epilogue belongs to data gradients, the input transform to forwards)
    if (bnb && xf) return nullptr;
#define OCL_CASE(M, N)                                                                        \
    if (MT == M && NT == N) return bnb ? conv_w_kernel<M, N, true, false> : xf ? conv_w_kernel<M, N, false, true> : conv_w_kernel<M, N, false, false>;
    OCL_CASE(1, 1) OCL_CASE(2, 1) OCL_CASE(3, 1) OCL_CASE(1, 2) OCL_CASE(2, 2)
#undef OCL_CASE
    return nullptr;
}

// the specialised instantiations: (MT, NT, NR, NU) of the four 3x3 stride-1 layers of a 32x32 training pass (layer 4 in chunks of 40 channels)
#define OCL_WX_SHAPES(X) X(2, 2, 12, 8) X(3, 1, 23, 10) X(1, 1, 45, 14) X(1, 1, 23, 7)
static convw_fn_t convwx_fn(int MT, int NT, int NR, int NU, int bnb, int xf, int trace = 0) {
    if (bnb && xf) return nullptr;
#define OCL_CASE(M, N, R, U)                                                                                                                        \
    if (MT == M && NT == N && NR == R && NU == U)                                                                                                   \
        return trace ? conv_wx_kernel<M, N, R, U, false, false, true>                                                                               \
                     : bnb ? conv_wx_kernel<M, N, R, U, true, false> : xf ? conv_wx_kernel<M, N, R, U, false, true> : conv_wx_kernel<M, N, R, U, false, false>;
    OCL_WX_SHAPES(OCL_CASE)
#undef OCL_CASE
    return nullptr;
}

int convw_set_det(int on) {
    const int v = on ? 1 : 0;
    OCL_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_det_sums), &v, sizeof(int)));
    return OCL_OK;
}

int convw_kernels_init() {
    for (int m = 1; m <= 3; ++m)
        for (int n = 1; n <= 2; ++n)
            for (int v = 0; v < 3; ++v)
                if (convw_fn_t f = convw_fn(m, n, v == 1, v == 2)) OCL_HIP(hipFuncSetAttribute((const void*)f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit));
#define OCL_CASE(M, N, R, U)                                                                                                                  \
    for (int v = 0; v < 4; ++v)                                                                                                               \
        if (convw_fn_t f = convwx_fn(M, N, R, U, v == 1, v == 2, v == 3)) OCL_HIP(hipFuncSetAttribute((const void*)f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit));
    OCL_WX_SHAPES(OCL_CASE)
#undef OCL_CASE
    return OCL_OK;
}

// ---- planner ------------------------------------------------------------------------------------------------------------------------------------
// Fills the plan for (MT, NT); OCL_ERR_ARG where the form does not fit: output classes, tiles that are neither whole lattice rows, nor a part
// of one row, nor whole images; weights of one channel split + eight patch buffers beyond the LDS at every channel chunking; a patch of more
// than kWNU pieces.
static int plan_conv_w_mn(const ConvGeomDesc& g, ConvPlan* p, int MT, int NT) {
    ConvArgs& a = p->a;
    if (g.ncls > 1 || g.Cin % 4 || g.Cout % 4) return OCL_ERR_ARG;
    const int LP = g.LH * g.LW, TP = 16 * NT;
    const int nt16 = cdiv(g.Cout, 16);
    if (MT > nt16) return OCL_ERR_ARG;
    const int COPW = 16 * MT;
    a.group_size = g.N / g.groups;
    a.groups = g.groups;
    int rows_l, cols_l;
    if (LP >= TP) {
        const bool whole_rows = TP % g.LW == 0 && LP % TP == 0, part_row = g.LW % TP == 0;
        if (!whole_rows && !part_row) return OCL_ERR_ARG;
        a.imgs = 1; a.ppi = TP; a.tiles_per_img = LP / TP;
        rows_l = whole_rows ? TP / g.LW : 1;
        cols_l = whole_rows ? g.LW : TP;
        a.aligned = whole_rows ? 1 : 2;   // 1: every tile starts at column 0 of the lattice (the staging offsets of in-image tiles are per-lane constants)
    } else {
        if (TP % LP || a.group_size % (TP / LP)) return OCL_ERR_ARG;
        a.imgs = TP / LP; a.ppi = LP; a.tiles_per_img = 1;
        rows_l = g.LH; cols_l = g.LW;
        a.aligned = 1;
    }
    a.PR = (rows_l - 1) * g.is + (a.max_dy - a.min_dy) + 1;
    a.PC = (cols_l - 1) * g.is + (a.max_dx - a.min_dx) + 1;
    a.C4tot = g.Cin / 4;
    if (a.imgs > 4 || a.PR > 16 || a.PC > 64) return OCL_ERR_ARG;
    if ((((a.imgs - 1) * a.Hin + a.PR) * a.Win + a.PC) * (g.Cin / 4) >= 8192) return OCL_ERR_ARG;   // 13-bit unit offsets
    const size_t xf_b = g.xf ? (size_t)g.groups * g.Cin * 8 : 0;
    const size_t bnb_b = g.bnb ? (size_t)g.groups * g.Cout * 12 : 0;
    a.bnb_lds = g.bnb ? (int)(xf_b / 4) : -1;
    // channel chunk: all input channels where the weights of the split + eight patch buffers fit, else the largest divisor that does
    size_t lds = 0;
    bool fit = false;
    for (int KC = g.Cin; KC >= 4 && !fit; KC -= 4) {
        if (g.Cin % KC) continue;
        const int nch = g.Cin / KC;
        a.KC = KC;
        const int SL = (KC / 4) | 1;     // 16-byte slots per pixel, odd: the b128 reads of 16 pixels spread over all banks
        a.CP = SL * 4;
        a.Qc = g.ntaps * (KC / 4);
        a.Qpad = (int)round_up(a.Qc, 4);
        const int units = a.imgs * a.PR * a.PC * SL;
        a.nstage = cdiv(units, 64);
        if (a.nstage > kWNU || a.Qpad > 256 || KC / 4 > 63) continue;
        a.patch_floats = a.nstage * 256;
        if ((size_t)a.patch_floats * 4 < (size_t)(2 * COPW + 1) * 8) continue;   // (the statistics slot lives there at the end)
        lds = (size_t)a.Qpad * 4 + (size_t)nch * a.Qpad * COPW * 16 + xf_b + bnb_b;
        a.qstat_off = (int)round_up(lds, 1024);
        lds = (size_t)a.qstat_off + (size_t)kWWaves * 2 * a.patch_floats * 4;
        fit = lds <= kLdsLimit - 1024;
    }
    if (!fit) return OCL_ERR_ARG;
    a.wres = 1; a.pipe = 0; a.QS = a.Qpad;
    a.n_splits = cdiv(nt16, MT);
    a.CoutP = a.n_splits * COPW;
    a.tiles_per_group = (a.group_size / a.imgs) * a.tiles_per_img;
    const int T = (g.N / a.imgs) * a.tiles_per_img;
    a.cls_pack = 1 | (g.ntaps << 4);
    a.cls_oyx = 0;
    a.xf = 0;
    p->cw = 1; p->cs = 0; p->q4 = 0; p->MT = MT; p->NT = NT;
    {   // 2: conv_wx_kernel has an instantiation for this shape (OCL_CONV_WX=0: the generic kernel everywhere -- A/B reference)
        static const bool env_wx = [] { const char* e = getenv("OCL_CONV_WX"); return !(e && atoi(e) == 0); }();
        if (env_wx && a.imgs == 1 && a.aligned == 1 && convwx_fn(MT, NT, a.Qpad / 4, a.nstage, 0, 0)) p->cw = 2;
    }
    p->lds_bytes = lds;
    a.WPT = g.WPT > 0 ? g.WPT : a.CoutP;
    for (int t = 0; t < 9; ++t) a.tpo[t] = t < a.ntaps ? ((a.tdy[t] - a.min_dy) * a.PC + (a.tdx[t] - a.min_dx)) * a.CP : 0;
    {   // the taps' pack indices as nibbles (conv_w_kernel's weight copy)
        unsigned lo = 0;
        for (int t = 0; t < 8 && t < a.ntaps; ++t) lo |= (unsigned)(a.tw[t] & 15) << (4 * t);
        a.d_c4 = (int)lo;
        a.d_pc = a.ntaps > 8 ? a.tw[8] : 0;
        bool ident = true;
        for (int t = 0; t < a.ntaps; ++t) ident = ident && a.tw[t] == t;
        a.d_row = (ident && a.KC == a.Cin) ? 1 : 0;
    }
    {
        auto magic = [](int d) { return d <= 1 ? 0u : (unsigned)(((1ull << 32) + (unsigned)d - 1) / (unsigned)d); };
        a.m_tpi = magic(a.tiles_per_img); a.m_lw = magic(a.LW); a.m_tpg = magic(a.group_size);
        a.m_kc4 = magic(a.KC / 4);
        a.m_ppi = a.m_pc = a.m_pr = 0;
        const int64_t xmax = std::max<int64_t>(std::max<int64_t>(T, LP), std::max(g.N, a.Qpad));
        const int64_t dmax = std::max(std::max(a.tiles_per_img, a.LW), std::max(a.group_size, a.KC / 4));
        if (xmax * dmax >= (1ll << 32)) return OCL_ERR_ARG;
    }
    // one workgroup per CU, one wave per SIMD; never more waves than tiles
    p->grid_y = a.n_splits;
    p->grid_x = std::max(1, std::min(256 / a.n_splits, cdiv(T, kWWaves)));
    if (p->grid_x >= 8) p->grid_x &= ~7;   // (whole XCD rounds: the kernel's block -> range map keeps neighbouring ranges on one L2)
    a.off_tdesc = (int)round_up(16 + 2 * a.Qpad, 4);
    a.off_pu = a.off_tdesc;
    a.off_loc = a.off_pu + a.nstage * 64;
    a.blob_ints = a.off_loc + 3 * NT * 16;
    a.blob = nullptr;
    return OCL_OK;
}

int plan_conv_w(const ConvGeomDesc& g, ConvPlan* p) {
    const int nt16 = cdiv(g.Cout, 16);
    if (g.force_MT || g.force_NT) {
        ConvPlan q = *p;
        const int rc = plan_conv_w_mn(g, &q, g.force_MT ? g.force_MT : std::min(3, nt16), g.force_NT ? g.force_NT : 1);
        if (rc == OCL_OK && convw_fn(q.MT, q.NT, 0, 0)) { *p = q; return OCL_OK; }
        return OCL_ERR_ARG;
    }
    // channel tiles per workgroup: the fewest issued tiles first (80 channels: five splits of one tile, not three of two), then the widest
    // workgroup (the B operand is read once per pixel tile and split); two pixel tiles per wave where a lattice row has 32 pixels (a wave's
    // tile is then a whole row) and the patches still fit
    ConvPlan best;
    int best_score = 1 << 30;
    for (int MT = std::min(3, nt16); MT >= 1; --MT) {
        const int score = cdiv(nt16, MT) * MT;
        if (score >= best_score) continue;
        for (int NT = (g.LW % 32 == 0 ? 2 : 1); NT >= 1; --NT) {
            ConvPlan q = *p;
            if (convw_fn(MT, NT, 0, 0) && plan_conv_w_mn(g, &q, MT, NT) == OCL_OK) {
                best = q;
                best_score = score;
                break;
            }
        }
    }
    if (best_score == (1 << 30)) return OCL_ERR_ARG;
    *p = best;
    return OCL_OK;
}

void conv_w_tables(const ConvPlan& p, std::vector<int>* out) {
    const ConvArgs& a = p.a;
    const int NT = p.NT;
    const int kc4 = a.KC / 4, SL = a.CP / 4;
    std::vector<int>& b = *out;
    b.assign((size_t)a.blob_ints, 0);
    int* qoff = b.data() + 16;
    int* qrow = qoff + a.Qpad;
    for (int q = 0; q < a.Qpad; ++q) {
        const bool ok = q < a.Qc;
        const int t = q / kc4, c4 = q % kc4;
        qoff[q] = ok ? a.tpo[t] + 4 * c4 : 0;
        qrow[q] = ok ? a.tw[t] * a.C4tot + c4 : -1;
    }
    // staging units: unit u of the flat [image][patch row][patch column][slot] patch = lane (u % 64) of piece (u / 64)
    int* uw = b.data() + a.off_pu;
    const int units = a.imgs * a.PR * a.PC * SL;
    for (int u = 0; u < a.nstage * 64; ++u) {
        int w = -1;
        if (u < units) {
            const int s = u % SL, pix = u / SL;
            const int pc = pix % a.PC, row = pix / a.PC;
            const int pr = row % a.PR, il = row / a.PR;
            if (s < kc4) w = wunit_pack(((il * a.Hin + pr) * a.Win + pc) * a.C4tot + s, s, pr, pc, il);
        }
        uw[u] = w;
    }
    // per-lane output pixels relative to the tile origin (every wave holds tiles of the same shape)
    int* lc = b.data() + a.off_loc;
    for (int r16 = 0; r16 < 16; ++r16)
        for (int nt = 0; nt < NT; ++nt) {
            const int r = nt * 16 + r16;
            const int il = r / a.ppi, pl = r % a.ppi;
            // a tile inside one lattice row: pl is the column offset from the tile's first pixel; whole rows / images: (ly, lx) from the tile origin
            const int ly = pl / a.LW, lx = pl % a.LW;
            lc[(3 * nt + 0) * 16 + r16] = ((il * a.PR + ly * a.is) * a.PC + lx * a.is) * a.CP;
            lc[(3 * nt + 1) * 16 + r16] = ((il * a.Hout + ly * a.os) * a.Wout + lx * a.os) * a.Cout;
            lc[(3 * nt + 2) * 16 + r16] = il;
        }
}

int launch_conv_w(const ConvPlan& p, hipStream_t s) {
    convw_fn_t fn = convw_fn(p.MT, p.NT, (p.a.flags & EPI_BNB) ? 1 : 0, p.a.xf);
    if (p.cw == 2) fn = convwx_fn(p.MT, p.NT, p.a.Qpad / 4, p.a.nstage, (p.a.flags & EPI_BNB) ? 1 : 0, p.a.xf, p.a.trace && !(p.a.flags & EPI_BNB) && !p.a.xf);
    else if (p.a.trace && !(p.a.flags & EPI_BNB) && !p.a.xf) {   // measurement builds (kbench)
        convw_fn_t ft = p.MT == 3 && p.NT == 1 ? conv_w_kernel<3, 1, false, false, true> : p.MT == 1 && p.NT == 1 ? conv_w_kernel<1, 1, false, false, true> :
                        p.MT == 2 && p.NT == 2 ? conv_w_kernel<2, 2, false, false, true> : nullptr;
        if (ft) {
            fn = ft;
            OCL_HIP(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit));
        }
    }
    if (!fn || !p.a.blob) {
        set_error("launch_conv_w: no kernel for MT=%d NT=%d / plan without device tables", p.MT, p.NT);
        return OCL_ERR_STATE;
    }
    ProfScope ps(PROF_CONV, s);
    hipLaunchKernelGGL(fn, dim3(p.grid_x, p.grid_y), dim3(256), p.lds_bytes, s, p.a);
    OCL_LAUNCH_CHECK();
    return OCL_OK;
}

}  // namespace ocl
