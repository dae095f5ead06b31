// K3: weight gradient of the Reduced-ResNet18 convolutions for gfx950 (its own translation unit: the kernel has ~90 instantiations and
// compiles beside conv.hip instead of behind it).
//
//  conv_wgrad_kernel  weight gradient as a (tap,ci) x co GEMM reduced over pixels, split-K over pixel tiles,
//                     partials summed by wgrad_reduce_kernel straight into PyTorch's OIHW gradient.
//
// Replaces the autograd of F.conv2d w.r.t. its weight behind models/resnet.py:10-12,32-37,90-99.
#include "conv_dev.h"
#include <string.h>
#include <algorithm>
#include <type_traits>
#include <cmath>

namespace ocl {

// =====================================================================================================
// weight gradient
// =====================================================================================================
// Weight gradient: (tap, ci) x co GEMM reduced over pixels.  A workgroup owns one (channel chunk, row block, column
// block) output tile and a strided subset of the pixel tiles; like the forward kernel it is software-pipelined: the dy
// tile and the input patch of the NEXT pixel tile are fetched into registers while the MFMAs of the current one run.
struct WTile {
    int img0, p0, oy0, nrows;
};

//
// RGW > 0 selects the 4x4x1 form of the product for layers with at most 20 output channels and one channel chunk (layer 1's 3x3
// convolutions, see plan_wgrad).  The 16x16x4 tiles pad layer 1's 180 x 20 gradient to 192 x 32: 41 % of the issued
// MFMAs multiply zeros (and the slabs are 32 columns wide for 20 channels).  With v_mfma_f32_4x4x1_16b_f32 the
// sixteen blocks of an instruction are sixteen PIXELS (the reduction dimension), and nothing is padded beyond quads:
//   block b = pixel s0 + b of the tile;   A: lane 4b + i holds the 16-byte unit u = 4 * rowgroup + i = (tap, channel quad) of that
//   pixel's patch (one ds_read_b128 feeds four MFMAs, k = channel inside the quad);   B: lane 4b + j holds dy[pixel][4s + j];
//   D: register e of lane 4b + j accumulates  x[unit 4 * rowgroup + e][k] * dy[4s + j]  summed over the pixels b, b + 16, ...
// so a wave owns RGW row groups (16 rows each) x 5 column quads x 4 channels = 20 * RGW accumulators, fed by RGW + 5 operand reads
// per 16 pixels, and the sixteen per-block partial sums are combined once, at the end (two DPP row shifts, two cross-row shuffles),
// in a fixed order.  The slab format is the 16x16x4 form's: the reduction kernels do not know which form wrote it.
//
// Staging of a pixel tile: which pixel slot / channel quad / patch position a thread's units are does not change from tile to tile;
// only the tile's base addresses and its validity limits do.  The thread keeps per unit a constant byte offset, an LDS address and a
// packed (row, patch row, image) word, and a tile costs an add, two compares and a select per load (round 4; the form that re-derived
// everything per tile from packed positions spent ~1100 instructions per tile and wave around ~600 of the K loop and is gone:
// profiles/r4_wgrad_tab_ab.txt).  Units that lie outside the tile or the patch for good store into a 16-byte dummy slot in front of
// the pixel table instead of branching around the store.
// TRACE = 1 (measurement build, kbench `wgradtrace`): s_memtime stamps of thread 0 at the phase boundaries into WgradArgs::trace,
// 64 slots per workgroup: start | prologue done | per tile: passed barrier 1, tile stored, passed barrier 2, next tile's loads issued,
// K loop done | ... | slab written.
// (the body takes its workgroup id from the caller: conv_wgrad_kernel passes blockIdx, conv_wgrad_multi_kernel the id inside its layer)
template <int MTW, int NTW, int PF, int RGW = 0, int TRACE = 0>
__device__ __forceinline__ void conv_wgrad_body(const WgradArgs& a, const int bid_x, const int bid_y) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    int* pixoff = (int*)lds_raw + 4;                // [KP]   (in front of it: the staging's dummy slot)
    float* dyt = (float*)(pixoff + a.KP);           // [KP][DP]
    float* patch = dyt + (size_t)a.KP * a.DP;       // [imgs][PR][PC][CP]
    float* xft = patch + (((size_t)a.imgs * a.PR * a.PC * a.CP + 3) & ~(size_t)3);   // input transform (WgradArgs::xf): [groups][Cin/4][2][4] scale / shift quads
    constexpr int BNW = RGW > 0 ? 4 * kQBlocks : 16 * NTW;
    constexpr int Q = BNW / 4;
    constexpr int DPF = (128 * Q + 255) / 256;      // dy prefetch registers (KP <= 128)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, g = lane >> 4;
    // (pixel split bx, output block by) of this workgroup.  xcd_by > 0 (a one-dimensional launch of S * by workgroups):
    // the `by` workgroups that read the SAME pixel tiles get linear ids that agree modulo 8 and lie within 8 * by of each other --
    // workgroup b is observed to run on XCD b % 8, so they share one L2 (4 MB per XCD, not coherent across XCDs) at about the same
    // time, instead of fetching every tile once per output block from memory.  The last S % 8 splits keep the plain order.
    int bx = bid_x, by = bid_y;
    if (a.xcd_by > 0) {
        const int id = bid_x, per = 8 * a.xcd_by, full = (a.S >> 3) * per;
        if (id < full) {
            const int grp = id / per, rem = id - grp * per;
            by = rem >> 3;
            bx = grp * 8 + (rem & 7);
        } else {
            const int r = a.S & 7, t = id - full;
            by = t / r;
            bx = (a.S & ~7) + (t - by * r);
        }
    }
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
    // 4x4x1 form: LDS offset of the lane's unit inside a pixel's patch (0 for the units past the last one: their rows are not written)
    constexpr int RG = RGW > 0 ? RGW : 1;
    int qoff[RG];
    f32x4 qacc[RG][kQBlocks][4];
    if constexpr (RGW > 0) {
        const int qk4 = a.KC >> 2, units = a.ntaps * qk4;
#pragma unroll
        for (int r = 0; r < RG; ++r) {
            const int u = ((by * 4 + wave) * RG + r) * 4 + (lane & 3);
            int c4;
            const int t = fdiv(min(u, units - 1), qk4, 1.0f / (float)qk4, c4);
            qoff[r] = u < units ? ((tap_sel(a.tdy, t) - a.min_dy) * a.PC + (tap_sel(a.tdx, t) - a.min_dx)) * a.CP + c4 * 4 : 0;
#pragma unroll
            for (int s = 0; s < kQBlocks; ++s)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    qacc[r][s][k] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    asm volatile("" : "+a"(qacc[r][s][k]));
                }
        }
    }

    const float inv_tpi = 1.0f / (float)a.tiles_per_img, inv_wo0 = 1.0f / (float)a.Wo;
    auto geom = [&](int tile) __attribute__((always_inline)) -> WTile {
        WTile t;
        int tp, rem;
        const int ti = fdiv(tile, a.tiles_per_img, inv_tpi, tp);
        t.img0 = ti * a.imgs;
        t.p0 = tp * a.ppi;
        t.oy0 = fdiv(t.p0, a.Wo, inv_wo0, rem);
        const int pend = min(t.p0 + a.ppi, LP);
        const int oy1 = fdiv(pend - 1, a.Wo, inv_wo0, rem);
        t.nrows = a.imgs > 1 ? min(a.imgs, a.N - t.img0) * a.PR : (oy1 - t.oy0) * a.stride + (a.max_dy - a.min_dy) + 1;
        return t;
    };

    // ---- per-thread unit bookkeeping (identical for every tile) ----------------------------------------------------
    const int kc4 = a.KC >> 2;
    const float inv_ppi = 1.0f / (float)a.ppi, inv_wo = 1.0f / (float)a.Wo;
    float4 dv[DPF], pv[PF];
    unsigned okm = 0;   // bit i: patch unit i of the tile in flight lies inside the image (input transform: the others stay zero)
    if (a.xf) {   // x is a raw convolution output: its BatchNorm + ReLU is applied while the patch is staged (first barrier of the tile loop publishes the table)
        const int C = a.Cin;
        for (int j = threadIdx.x; j < a.xf_groups * C; j += 256) {
            const int gq = j / C, c = j - gq * C;
            float sc, sh;
            bn_scale_shift(a.xf_gamma[c], a.xf_beta[c], a.xf_mean[j], a.xf_invstd[j], sc, sh);
            float* t = xft + (size_t)(gq * (C >> 2) + (c >> 2)) * 8 + (c & 3);
            t[0] = sc;
            t[4] = sh;
        }
    }
    const float inv_gs = a.xf ? 1.0f / (float)a.xf_group_size : 0.f;
    const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(a.x), rs_dy = make_rsrc(a.dy);
    // ---- per-unit constants ----------------------------------------------------------------------------------------------
    int d_pl[DPF], d_il[DPF], d_goff[DPF], d_lds[DPF];   // pixel inside its image (huge: never valid), image inside the tile, byte offset from the tile's dy base, LDS float offset of the quad (dummy slot when the unit is outside the tile)
    int px_pl = 0, px_il = 0, px_idx = -4, pxo = 0;          // the pixel-table entry of pixel slot `tid` (threads past the tile write the dummy slot)
    int p_word[PF], p_goff[PF], p_lds[PF];   // row | patch row << 16 | image << 24 (row 0xffff: never loaded);  byte offset from the tile's x base;  LDS byte offset | channel quad << 24
    {
        const int dummy_f = (int)((float*)lds_raw - dyt);   // float offset of the dummy slot relative to dyt (negative)
#pragma unroll
        for (int i = 0; i < DPF; ++i) {
            const int u = tid + i * 256;
            int c4, pl;
            const int q = fdiv(u, Q, 1.0f / (float)Q, c4);
            const int il = fdiv(q, a.ppi, inv_ppi, pl);
            const bool in_tile = q < a.KP;
            d_pl[i] = (in_tile && il < a.imgs && n0 + c4 * 4 < a.Cout) ? pl : 0x20000000;
            d_il[i] = il;
            d_goff[i] = ((il * LP + pl) * a.Cout + n0 + c4 * 4) * 4;
            d_lds[i] = in_tile ? q * a.DP + c4 * 4 : dummy_f;
        }
        {
            int pl;
            const int il = fdiv(tid, a.ppi, inv_ppi, pl);
            px_pl = (tid < a.KP && il < a.imgs) ? pl : 0x20000000;
            px_il = il;
            px_idx = tid < a.KP ? tid : -4;
        }
        const int n_units = a.imgs * a.PR * a.PC * kc4;
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            const int u = tid + i * 256;
            int c4, pc, pr;
            const int pix = fdiv(u, kc4, 1.0f / (float)kc4, c4);
            const int row = fdiv(pix, a.PC, 1.0f / (float)a.PC, pc);
            const int il = fdiv(row, a.PR, a.inv_PR, pr);
            const int ix = a.min_dx + pc;
            const bool inside = u < n_units;
            const bool x_ok = inside && ix >= 0 && ix < a.Win;
            p_word[i] = (x_ok ? row : 0xffff) | ((pr & 255) << 16) | ((il & 127) << 24);
            p_goff[i] = (((il * a.Hin + pr) * a.Win + pc) * a.Cin + c4 * 4) * 4;
            p_lds[i] = (inside ? (int)((patch - (float*)lds_raw) + (row * a.PC + pc) * a.CP + c4 * 4) * 4 : 0) | (c4 << 24);
        }
    }
    auto load_tile = [&](const WTile& t) __attribute__((always_inline)) {
        const int dbase = (t.img0 * LP + t.p0) * a.Cout * 4;
        const int ox0 = t.p0 - t.oy0 * a.Wo;
#pragma unroll
        for (int i = 0; i < DPF; ++i) {
            const int p = t.p0 + d_pl[i], n = t.img0 + d_il[i];
            const bool v = (p < LP) & (n < a.N);
            dv[i] = buf_load16(rs_dy, v ? dbase + d_goff[i] : kOob);   // zeros when masked
        }
        {   // LDS patch offset of pixel slot `tid`: (row, column) of the pixel relative to the tile's first output row (branch-free division)
            const int p = t.p0 + px_pl, n = t.img0 + px_il;
            const bool v = (p < LP) & (n < a.N);
            const int r = (ox0 + px_pl) & 0x3fffff;
            int dr = (int)((float)r * inv_wo), ox = r - dr * a.Wo;
            const int lo = ox < 0 ? 1 : 0, hi = ox >= a.Wo ? 1 : 0;
            dr += hi - lo;
            ox += (lo - hi) * a.Wo;
            pxo = v ? ((px_il * a.PR + dr * a.stride) * a.PC + ox * a.stride) * a.CP : 0;
        }
        const int iy0 = t.oy0 * a.stride + a.min_dy;
        const int base = (((t.img0 * a.Hin + iy0) * a.Win + a.min_dx) * a.Cin + c0) * 4;   // bytes; may be negative (halo)
        okm = 0;
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            const int row = p_word[i] & 0xffff, iy = iy0 + ((p_word[i] >> 16) & 255);
            const bool ok = (row < t.nrows) & (iy >= 0) & (iy < a.Hin);
            pv[i] = buf_load16(rs_x, ok ? base + p_goff[i] : kOob);
            okm |= ok ? (1u << i) : 0u;
        }
    };
    auto store_tile = [&](const WTile& t) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < DPF; ++i) *(float4*)(dyt + d_lds[i]) = dv[i];
        pixoff[px_idx] = pxo;
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            float4 v = pv[i];
            if (a.xf) {   // block-uniform
                int rem;
                const int gq = min(fdiv(t.img0 + (p_word[i] >> 24), a.xf_group_size, inv_gs, rem), a.xf_groups - 1);
                const float* tb = xft + (size_t)(gq * (a.Cin >> 2) + (c0 >> 2) + ((unsigned)p_lds[i] >> 24)) * 8;
                const float4 sc = *(const float4*)tb, sh = *(const float4*)(tb + 4);
                v.x = fmaxf(__fmaf_rn(v.x, sc.x, sh.x), 0.f); v.y = fmaxf(__fmaf_rn(v.y, sc.y, sh.y), 0.f);
                v.z = fmaxf(__fmaf_rn(v.z, sc.z, sh.z), 0.f); v.w = fmaxf(__fmaf_rn(v.w, sc.w, sh.w), 0.f);
                if (!((okm >> i) & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            float* d = (float*)(lds_raw + (p_lds[i] & 0xffffff));
            *(float2*)d = make_float2(v.x, v.y);
            *(float2*)(d + 2) = make_float2(v.z, v.w);
        }
    };
    int tr_n = 0;
    auto stamp = [&]() __attribute__((always_inline)) {
        if constexpr (TRACE) {
            if (tid == 0 && tr_n < 64) a.trace[(size_t)(by * a.S + bx) * 64 + tr_n++] = __builtin_amdgcn_s_memtime();
        }
    };
    // the K loop of the tile that sits in LDS
    auto compute_tile = [&]() __attribute__((always_inline)) {
        if constexpr (RGW > 0) {   // 16 pixels per step (KP is a multiple of 16 in this form); operands of step g + 1 are read while the MFMAs of step g issue
            const float* dq = dyt + (size_t)(lane >> 2) * a.DP + (lane & 3);
            const int* pq = pixoff + (lane >> 2);
            const int ng = a.KP >> 4;
            float bv[2][kQBlocks];
            float4 av[2][RG];
            int po_n = pq[0];   // (the pixel's patch offset is read one step ahead of the operand reads that depend on it)
            auto fetch = [&](int set, int g) __attribute__((always_inline)) {
                const int po = po_n;
                po_n = pq[min(g + 1, ng - 1) * 16];
#pragma unroll
                for (int r = 0; r < RG; ++r) av[set][r] = *(const float4*)(patch + po + qoff[r]);
#pragma unroll
                for (int s = 0; s < kQBlocks; ++s) bv[set][s] = dq[(size_t)g * 16 * a.DP + 4 * s];
            };
            auto fma = [&](int set) __attribute__((always_inline)) {
#pragma unroll
                for (int r = 0; r < RG; ++r)
#pragma unroll
                    for (int s = 0; s < kQBlocks; ++s) {
                        qacc[r][s][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[set][r].x, bv[set][s], qacc[r][s][0], 0, 0, 0);
                        qacc[r][s][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[set][r].y, bv[set][s], qacc[r][s][1], 0, 0, 0);
                        qacc[r][s][2] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[set][r].z, bv[set][s], qacc[r][s][2], 0, 0, 0);
                        qacc[r][s][3] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[set][r].w, bv[set][s], qacc[r][s][3], 0, 0, 0);
                    }
            };
            fetch(0, 0);
            int g = 0;
            for (; g + 2 <= ng; g += 2) {
                fetch(1, g + 1);
                fma(0);
                if (g + 2 < ng) fetch(0, g + 2);
                fma(1);
            }
            if (g < ng) fma(0);
            // (the accumulators are pinned to AccVGPRs across the tile loop: left alone, the register allocator keeps them in ArchVGPRs
            // outside the pixel loop and copies all 80 * RGW of them in and out per tile -- spilling at RGW = 3)
#pragma unroll
            for (int r = 0; r < RG; ++r)
#pragma unroll
                for (int s = 0; s < kQBlocks; ++s)
#pragma unroll
                    for (int k = 0; k < 4; ++k) asm volatile("" : "+a"(qacc[r][s][k]));
            stamp();
            return;
        }
        const float* pb = dyt + (size_t)g * a.DP + r16;
        // (A hand-pipelined form of this loop -- two operand register sets, the reads of iteration i + 1 issued in front of the MFMAs of
        // iteration i, table entries one iteration further ahead, pinned with sched_barriers: the conv kernel's recipe -- measured 5 - 8 %
        // SLOWER on every layer, profiles/r3_kbench_wgrad_pipelined_ab.txt: with two or three waves per SIMD the other waves already cover
        // the two LDS round trips of an iteration, and the second register set costs occupancy.  What bounds this loop is the number of
        // LDS instructions, one 4-byte read per MFMA; the remedy is K-grouped 16-byte operands, i.e. channel-major tiles.)
        auto ksteps = [&](int s, auto UC) __attribute__((always_inline)) {
            constexpr int U = decltype(UC)::value;
            int po[U];
            float av[U][MTW], bv[U][NTW];
#pragma unroll
            for (int u = 0; u < U; ++u) po[u] = pixoff[s + 4 * u + g];
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt) av[u][mt] = patch[po[u] + aoff[mt]];
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) bv[u][nt] = pb[(size_t)(s + 4 * u) * a.DP + nt * 16];
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][mt], bv[u][nt], acc[mt][nt], 0, 0, 0);
        };
        int s = 0;
        for (; s + 16 <= a.KP; s += 16) ksteps(s, std::integral_constant<int, 4>());
        for (; s < a.KP; s += 4) ksteps(s, std::integral_constant<int, 1>());
        stamp();
    
    };
    stamp();
    int tile = bx;
    WTile cur = geom(tile);
    if (tile < a.total_tiles) load_tile(cur);
    stamp();
    for (; tile < a.total_tiles; tile += a.S) {
        __syncthreads();  // previous tile consumed
        stamp();
        store_tile(cur);
        stamp();
        __syncthreads();
        stamp();
        const int next = tile + a.S;
        if (next < a.total_tiles) {
            cur = geom(next);
            load_tile(cur);
        }
        stamp();
        compute_tile();
    }
    // partial tile out: rows (chunk, mblock, m), cols co
    const int mrows_chunk = a.mblocks_per_chunk * 64 * MTW;
    float* dst = a.partial + (int64_t)bx * a.Mrows_total * a.CoutP;
    if constexpr (RGW > 0) {
        // Sum over the sixteen blocks (lanes 4b + j, b = 0..15), in a fixed order.  The four registers e of an accumulator are four rows
        // of the gradient, and the wave has four DPP rows: two v_permlane16_swap + one v_permlane32_swap (gfx950: exchanges of whole
        // 16- / 32-lane groups between two registers, no LDS) add the rows' partial sums so that DPP row R is left with register R's --
        // a reduce-scatter, 3 exchanges + 3 adds per accumulator instead of 8 shuffles + 8 adds -- then two row shifts add the four
        // blocks of the row, and lanes 12..15 of every row store (16 lanes, one instruction per accumulator).
        const int qk4 = a.KC >> 2, units = a.ntaps * qk4;
        int rowoff[RG];   // slab offset of (row of unit 4 * rowgroup + DPP row, channel 0) + column j, -1: past the last unit / not a storing lane
#pragma unroll
        for (int r = 0; r < RG; ++r) {
            const int u = ((by * 4 + wave) * RG + r) * 4 + (lane >> 4);
            int c4;
            const int t = fdiv(min(u, units - 1), qk4, 1.0f / (float)qk4, c4);
            rowoff[r] = (u < units && (lane & 12) == 12) ? (t * a.KC + c4 * 4) * a.CoutP + (lane & 3) : -1;
        }
        auto sw16 = [](float x, float y) __attribute__((always_inline)) -> float {   // rows: [x0 + x1, y0 + y1, x2 + x3, y2 + y3]
            const auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, y), false, false);
            return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
        };
        auto sw32 = [](float x, float y) __attribute__((always_inline)) -> float {   // halves: [x.lo + x.hi, y.lo + y.hi]
            const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, y), false, false);
            return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
        };
#pragma unroll
        for (int r = 0; r < RG; ++r)
#pragma unroll
            for (int s = 0; s < kQBlocks; ++s)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const f32x4 v = qacc[r][s][k];
                    float x = sw32(sw16(v[0], v[1]), sw16(v[2], v[3]));   // DPP row R: register R summed over the four rows
                    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x114, 0xf, 0xf, true));   // row_shr:4
                    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x118, 0xf, 0xf, true));   // row_shr:8
                    if (rowoff[r] >= 0 && 4 * s + (lane & 3) < a.Cout) dst[rowoff[r] + k * a.CoutP + 4 * s] = x;
                }
        stamp();
        return;
    }
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int ml = m0 + wave * 16 * MTW + mt * 16 + g * 4 + reg;   // row inside the chunk: (tap, channel)
            const int row = chunk * mrows_chunk + ml;
            if (ml >= a.Mchunk) continue;   // padding rows / columns are never read by the reduction: not written either
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
                if (n0 + nt * 16 + r16 < a.Cout) dst[(int64_t)row * a.CoutP + n0 + nt * 16 + r16] = acc[mt][nt][reg];
        }
    stamp();
}
template <int MTW, int NTW, int PF, int RGW = 0, int TRACE = 0>
__global__ void __launch_bounds__(256) conv_wgrad_kernel(const WgradArgs a) {
    conv_wgrad_body<MTW, NTW, PF, RGW, TRACE>(a, blockIdx.x, blockIdx.y);
}

// Every layer's weight gradient of a replay-sized pass in ONE launch (agents/exp_replay.py:36-89: 10 + 10 images per step).  Such a pass is
// bound by the NUMBER of dependent launches (~7 us each, 130 of them), and its 20 weight-gradient launches depend on nothing but their own
// layer's input and dL/dy: the backward keeps every dL/dy (ocl_net: one buffer per layer) and launches them together at its end.  The
// workgroup looks up its layer in the start table, reads that layer's WgradArgs from the device table (uniform address, constant address
// space: scalar loads, as the kernarg segment of the per-layer launch) and runs the same body: same slabs, same bits.  The forms a small
// pass plans: 16 x {32, 48} blocks at either prefetch depth, and the two 128-row forms of the 84 x 84 passes.
typedef const WgradMultiEntry __attribute__((address_space(4))) * WgradTabPtr;
template <int SET>   // 0: the 16-row forms (3 workgroups per CU by registers), 1: the 32- / 48-row forms
__global__ void __launch_bounds__(256) conv_wgrad_multi_kernel(const WgradMultiArgs m) {
    int l = 0;
    while (l + 1 < m.n && (int)blockIdx.x >= m.start[l + 1]) ++l;
    const WgradTabPtr e = (WgradTabPtr)m.tab + l;
    const WgradArgs& a = (const WgradArgs&)e->a;
    const int id = (int)blockIdx.x - m.start[l];
    int bx = id, by = 0;
    if (e->a.xcd_by <= 0) {
        by = id / e->grid_x;
        bx = id - by * e->grid_x;
    }
    if constexpr (SET == 0) {
        switch (e->variant) {
            case 0: conv_wgrad_body<1, 2, 4>(a, bx, by); break;
            case 1: conv_wgrad_body<1, 2, 8>(a, bx, by); break;
            case 2: conv_wgrad_body<1, 3, 4>(a, bx, by); break;
            default: conv_wgrad_body<1, 3, 8>(a, bx, by); break;
        }
    } else {
        if (e->variant == 4) conv_wgrad_body<2, 3, 8>(a, bx, by);
        else conv_wgrad_body<3, 2, 8>(a, bx, by);
    }
}

typedef void (*wgrad_fn_t)(const WgradArgs);
static wgrad_fn_t wgrad_fn(int M, int N, int PF) {
#define OCL_CASE(A, B)                                          \
    if (M == A && N == B) {                                     \
        if (PF == 4) return conv_wgrad_kernel<A, B, 4>;         \
        if (PF == 8) return conv_wgrad_kernel<A, B, 8>;         \
    }
    OCL_CASE(1, 1) OCL_CASE(1, 2) OCL_CASE(1, 3) OCL_CASE(1, 4) OCL_CASE(1, 5)
    OCL_CASE(2, 1) OCL_CASE(2, 2) OCL_CASE(2, 3) OCL_CASE(2, 4) OCL_CASE(2, 5)
    OCL_CASE(3, 1) OCL_CASE(3, 2) OCL_CASE(3, 3) OCL_CASE(3, 4) OCL_CASE(3, 5)
    OCL_CASE(4, 1) OCL_CASE(4, 2) OCL_CASE(4, 3) OCL_CASE(4, 4) OCL_CASE(4, 5)
#undef OCL_CASE
    return nullptr;
}
static wgrad_fn_t wgrad_q_fn(int rgw, int PF) {
#define OCL_CASE(R) \
    if (rgw == R) return PF == 4 ? conv_wgrad_kernel<1, 1, 4, R> : conv_wgrad_kernel<1, 1, 8, R>;
    OCL_CASE(1) OCL_CASE(2) OCL_CASE(3)
#undef OCL_CASE
    return nullptr;
}
// measurement builds (TRACE) of the forms the SCR pass runs most
static wgrad_fn_t wgrad_trace_fn(int M, int N, int PF, int rgw) {
    if (PF != 8) return nullptr;
    if (rgw == 3) return conv_wgrad_kernel<1, 1, 8, 3, 1>;
    if (rgw) return nullptr;
    if (M == 2 && N == 3) return conv_wgrad_kernel<2, 3, 8, 0, 1>;
    if (M == 3 && N == 2) return conv_wgrad_kernel<3, 2, 8, 0, 1>;
    if (M == 1 && N == 3) return conv_wgrad_kernel<1, 3, 8, 0, 1>;
    if (M == 1 && N == 2) return conv_wgrad_kernel<1, 2, 8, 0, 1>;
    return nullptr;
}
static int wgrad_pf_for(int units) { return units <= 1024 ? 4 : 8; }

// sums the split-K partials into the OIHW gradient: grad[co][ci][t] (+)= sum_s partial[s][(chunk,t,cc)][co].
// 32 consecutive outputs (co fastest: coalesced partial reads) x 8 split lanes per block; the 8 lane sums are combined
// through LDS in a fixed order, so the result does not depend on scheduling.
__device__ __forceinline__ void wgrad_reduce_body(const float* __restrict__ partial, int S, int Mrows_total, int CoutP, int mrows_chunk,
                                                  int KC, int ntaps, int CinReal, int Cout, float* __restrict__ grad, int accumulate,
                                                  int block, float (*red)[33]) {
    const int o = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int idx = block * 32 + o;  // (t, ci, co) with co fastest
    const int total = ntaps * CinReal * Cout;
    const bool valid = idx < total;
    const int co = idx % Cout;
    const int r = idx / Cout;
    const int ci = r % CinReal, t = r / CinReal;
    const int chunk = ci / KC, cc = ci - chunk * KC;
    const int row = chunk * mrows_chunk + t * KC + cc;
    const int64_t stride = (int64_t)Mrows_total * CoutP;
    const float* p = partial + (int64_t)row * CoutP + co;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (valid) {
        int s = sl;
        for (; s + 24 < S; s += 32) {
            s0 += p[(int64_t)s * stride];
            s1 += p[(int64_t)(s + 8) * stride];
            s2 += p[(int64_t)(s + 16) * stride];
            s3 += p[(int64_t)(s + 24) * stride];
        }
        for (; s < S; s += 8) s0 += p[(int64_t)s * stride];
    }
    red[sl][o] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (sl == 0 && valid) {
        float v = ((red[0][o] + red[1][o]) + (red[2][o] + red[3][o])) + ((red[4][o] + red[5][o]) + (red[6][o] + red[7][o]));
        float* gp = grad + ((int64_t)co * CinReal + ci) * ntaps + t;
        if (accumulate) v += *gp;
        *gp = v;
    }
}

// Few slabs (S <= 4: layer 4 of a replay-sized pass, two thirds of the network's weights): one thread per output sums its S slab
// entries itself, 256 outputs per block instead of 32 x 8 split lanes of which most idled.  Same arithmetic as the split form at
// S <= 8 -- lane sums of one term each, combined in the same tree -- hence the same bits (csrc/netcheck against the library before:
// 0 of 64 tensors differ).  20-image pass 832.5 -> 818.5 us, 13 images 807.7 -> 791.2; with the form taken up to S = 8 a 64-view pass
// got 20 us SLOWER (its layer 4 has 8 slabs: all split lanes busy, and eight times the threads in flight): profiles/r5_reduce_few.txt
__device__ __forceinline__ void wgrad_reduce_few(const float* __restrict__ partial, int S, int Mrows_total, int CoutP, int mrows_chunk,
                                                 int KC, int ntaps, int CinReal, int Cout, float* __restrict__ grad, int accumulate, int block) {
    const int idx = block * 256 + threadIdx.x;  // (t, ci, co) with co fastest
    if (idx >= ntaps * CinReal * Cout) return;
    const int co = idx % Cout;
    const int r = idx / Cout;
    const int ci = r % CinReal, t = r / CinReal;
    const int chunk = ci / KC, cc = ci - chunk * KC;
    const int row = chunk * mrows_chunk + t * KC + cc;
    const int64_t stride = (int64_t)Mrows_total * CoutP;
    const float* p = partial + (int64_t)row * CoutP + co;
    float l[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) l[k] = k < S ? p[(int64_t)k * stride] : 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) l[k] = (l[k] + 0.f) + (0.f + 0.f);   // the split form's (s0 + s1) + (s2 + s3) with one term per lane
    const float z = (0.f + 0.f) + (0.f + 0.f);                      // (lanes 4 .. 7 of the split form hold this)
    float v = ((l[0] + l[1]) + (l[2] + l[3])) + ((z + z) + (z + z));
    float* gp = grad + ((int64_t)co * CinReal + ci) * ntaps + t;
    if (accumulate) v += *gp;
    *gp = v;
}
constexpr int kReduceFewMaxS = 4;
static int reduce_blocks(int S, int total) { return S <= kReduceFewMaxS ? cdiv(total, 256) : cdiv(total, 32); }

__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float* __restrict__ partial, int S, int Mrows_total, int CoutP,
                                                           int mrows_chunk, int KC, int ntaps, int CinReal, int Cout,
                                                           float* __restrict__ grad, int accumulate) {
    __shared__ float red[8][33];
    if (S <= kReduceFewMaxS) {   // (uniform per launch)
        wgrad_reduce_few(partial, S, Mrows_total, CoutP, mrows_chunk, KC, ntaps, CinReal, Cout, grad, accumulate, blockIdx.x);
        return;
    }
    wgrad_reduce_body(partial, S, Mrows_total, CoutP, mrows_chunk, KC, ntaps, CinReal, Cout, grad, accumulate, blockIdx.x, red);
}

// the reductions of ALL layers of a backward pass in one launch (replay-sized batches run the whole backward on one stream and are
// bound by the number of dependent launches: 21 reductions -> 1); every layer keeps its own slab region until then
__global__ void __launch_bounds__(256) wgrad_reduce_multi_kernel(const WgradReduceMulti m) {
    __shared__ float red[8][33];
    int l = 0;
#pragma unroll 1
    while (l + 1 < m.n && (int)blockIdx.x >= m.L[l + 1].block0) ++l;
    const WgradReduceLayer& d = m.L[l];
    if (d.S <= kReduceFewMaxS) {   // (uniform per block)
        wgrad_reduce_few(m.partial + d.partial_off, d.S, d.Mrows_total, d.CoutP, d.mrows_chunk, d.KC, d.ntaps, d.CinReal, d.Cout,
                         m.grads + d.grad_off, m.accumulate, (int)blockIdx.x - d.block0);
        return;
    }
    wgrad_reduce_body(m.partial + d.partial_off, d.S, d.Mrows_total, d.CoutP, d.mrows_chunk, d.KC, d.ntaps, d.CinReal, d.Cout,
                      m.grads + d.grad_off, m.accumulate, (int)blockIdx.x - d.block0, red);
}

// LDS pixel stride of the wgrad input patch.  A reads (ds_read_b32, 32 banks, lanes 0-31 = 2 pixels x 16 channels)
// are conflict-free when stride*CP = 16 (mod 32); take the smallest even CP >= KC within 4 banks of that.
static int wg_cp(int kc, int stride) {
    for (int cp = kc;; cp += 2) {
        const int x = (stride * cp) & 31;
        const int d = std::min(x, 32 - x);
        if (16 - d <= 4) return cp;
    }
}

int plan_wgrad(int N, int Hin, int Win, int Cin, int Ho, int Wo, int Cout, int ksize, int stride, WgradPlan* p, int xf_groups, int wg_target) {
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
    // 48 output channels per workgroup; 80 (five 16-column blocks: layers 3 - 4 exactly, no padded columns -- 96 / 192 issued for 80 / 160
    // -- and 6 operand reads per 5 MFMAs) on passes of 48 images and more.  Alone (kbench) the wider block is faster at every size (220
    // views: 592 -> 546 us over the 20 launches, layer 4 39.2 -> 30.8 us; 20 x 84 x 84: layer 3 31.6 -> 23.6 us), but beside the dependent
    // chain of a 20-image pass the whole pass gets SLOWER (20 x 32 x 32: +8 us, 20 x 84 x 84: +16 us), while passes of 50 - 220 images gain
    // 22 - 75 us (profiles/r5_wgrad_ntw5.txt): the rule follows the pass times.
    const int ntmax = (ntile >= 5 && N >= 48) ? 5 : 3;
    int NTW = ntile <= ntmax ? ntile : ntmax;
    a.nblocks = cdiv(ntile, NTW);
    if (a.nblocks > 1) NTW = cdiv(ntile, a.nblocks);
    a.CoutP = a.nblocks * NTW * 16;
    int dp = NTW * 16;
    while ((dp & 31) != 16) dp += 16;
    a.DP = dp;
    const int LP = Ho * Wo;
    // pixel tile (KP output pixels, 128 / 64 / 32) and channel chunk KC: the largest tile whose patch + dy fit the LDS
    // target and the prefetch registers with a chunk of at least min(20, Cin) channels; else the best that fits at all.
    bool found = false;
    // largest pixel tile, workgroup target of the pixel split, smallest grid that stops the search (swept in round 4:
    // profiles/r4_kbench_wgrad_planner_sweep.txt, r4_wgrad_knobs_netcheck.txt -- these are the best column)
    // (wg_target > 0: the layer shares its launch with the other layers of the pass -- conv_wgrad_multi_kernel -- and need not fill the
    // machine alone: fewer pixel splits, i.e. fewer slabs to write and to reduce)
    constexpr int env_kp = 128;
    // (OCL_WGRAD_TARGET / OCL_WGRAD_ENOUGH: measurement overrides, profiles/r6_wgrad_mtw_ab.txt)
    static const int ov_target = [] { const char* e = getenv("OCL_WGRAD_TARGET"); return e ? atoi(e) : 0; }();
    static const int ov_enough = [] { const char* e = getenv("OCL_WGRAD_ENOUGH"); return e ? atoi(e) : 0; }();
    const int env_target = wg_target > 0 ? wg_target : (ov_target > 0 ? ov_target : 512);
    const int env_enough = wg_target > 0 ? std::max(1, wg_target * 3 / 4) : (ov_enough > 0 ? ov_enough : 384);
    for (int pass = 0; pass < 2 && !found; ++pass) {
        for (int KPmax = env_kp; KPmax >= 32 && !found; KPmax /= 2) {
            if (LP >= KPmax) {
                a.imgs = 1; a.ppi = KPmax; a.tiles_per_img = cdiv(LP, KPmax); a.KP = KPmax;
            } else {
                a.imgs = std::min(KPmax / LP, N); a.ppi = LP; a.tiles_per_img = 1; a.KP = (int)round_up((int64_t)a.imgs * LP, 4);
            }
            a.PC = (Wo - 1) * stride + (a.max_dx - a.min_dx) + 1;
            const int rows_l = (a.imgs == 1 && LP >= KPmax) ? std::min(Ho, (KPmax + Wo - 2) / Wo + 1) : Ho;
            a.PR = (rows_l - 1) * stride + (a.max_dy - a.min_dy) + 1;
            if (a.imgs > 127 || a.PR >= 256 || a.PC >= 256) continue;
            for (int KC = Cin; KC >= 4; KC -= 4) {
                if (Cin % KC) continue;
                if (pass == 0 && KC < std::min(20, Cin)) break;
                a.KC = KC; a.CP = wg_cp(KC, stride);
                const size_t bytes = 16 + (size_t)a.KP * 4 + (size_t)a.KP * a.DP * 4 + (size_t)a.imgs * a.PR * a.PC * a.CP * 4 +
                                     (size_t)xf_groups * Cin * 8 + (xf_groups ? 16 : 0);   // (dummy slot + ... + the input-transform table)
                const bool fits = (pass == 0 ? bytes <= kLdsTarget : bytes <= kLdsLimit - 1024) &&
                                  a.imgs * a.PR * a.PC * (KC / 4) <= 256 * kPatchPF;
                if (fits) { p->lds_bytes = bytes; found = true; break; }
            }
        }
    }
    if (!found) {
        set_error("plan_wgrad: no pixel tile fits the LDS (Hin=%d Win=%d Cin=%d Cout=%d)", Hin, Win, Cin, Cout);
        return OCL_ERR_ARG;
    }
    a.nchunks = Cin / a.KC;
    a.Mchunk = a.ntaps * a.KC;
    {   // patch prefetch walk: 256 units = d_row rows + d_pc pixels + d_c4 float4s
        const int kc4 = a.KC / 4;
        a.d_c4 = 256 % kc4;
        const int d_pix = 256 / kc4;
        a.d_pc = d_pix % a.PC;
        a.d_row = d_pix / a.PC;
        a.inv_PR = 1.0f / (float)a.PR;
    }
    const int mtiles = cdiv(a.Mchunk, 16);
    a.total_tiles = cdiv(N, a.imgs) * a.tiles_per_img;
    // Block tile (64*MTW rows) and pixel split S: aim at >= 384 workgroups (1.5 per CU) with the largest tile that
    // gets there, cap the split so that the fp32 partial slabs stay <= 12 MB (they are written and read once), and
    // balance the pixel tiles over the S slices.
    int MTW = 1, bestS = 1;
    int64_t best_blocks = -1;
    for (int m = wg_target > 0 ? 1 : std::min(4, cdiv(mtiles, 4)); m >= 1; --m) {   // (merged launch: the 64-row forms of its kernel)
        if (m * NTW > 20) continue;
        const int mb = cdiv(mtiles, 4 * m);
        const int by = a.nchunks * mb * a.nblocks;
        const int64_t slab = (int64_t)a.nchunks * mb * 64 * m * a.CoutP * 4;
        const int s_cap = (int)std::max<int64_t>(1, (12ll << 20) / slab);
        int S = std::max(1, std::min(std::min(a.total_tiles, s_cap), cdiv(env_target, by)));
        const int tpb = cdiv(a.total_tiles, S);
        S = cdiv(a.total_tiles, tpb);
        const int64_t blocks = (int64_t)by * S;
        if (blocks > best_blocks) { best_blocks = blocks; MTW = m; bestS = S; }
        if (blocks >= env_enough) break;
    }
    a.mblocks_per_chunk = cdiv(mtiles, 4 * MTW);
    a.Mrows_total = a.nchunks * a.mblocks_per_chunk * 64 * MTW;
    const int by = a.nchunks * a.mblocks_per_chunk * a.nblocks;
    a.S = bestS;
    p->MTW = MTW; p->NTW = NTW;
    p->grid_x = a.S; p->grid_y = by;
    p->partial_floats = (size_t)a.S * a.Mrows_total * a.CoutP;
    // The 4x4x1 form for <= 20 output channels and a single channel chunk: layer 1's 3x3 convolutions (round 5: the default where the
    // gate below plans it -- whole GPU suite green with it, -13 .. -33 us per SCR pass; OCL_WGRAD_Q=0 keeps the 16x16x4 form everywhere,
    // which tests/test_gpu_netcheck.py uses as the reference of its A/B).  Row groups (16 gradient rows) per wave: the smallest count
    // that covers the rows with one row block, i.e. the patch is staged once per pixel tile; the pixel split aims at 256 workgroups (the
    // block sums of the epilogue cost about one pixel tile's MFMAs, so fewer, longer workgroups than the 16x16x4 form).
    static const int env_q = [] { const char* e = getenv("OCL_WGRAD_Q"); return e ? atoi(e) : 1; }();
    // (the stem's 9 units fill 3 of 4 waves: measured slower; the block sums of the epilogue cost about 1.7 pixel tiles, and the form
    // runs one workgroup per CU: it pays from ~6 tiles of 128 pixels per workgroup at 256 workgroups -- SCR's 220 views yes,
    // 20 images of 84 x 84 no (+40 us); OCL_WGRAD_Q=2 lifts that limit for the planner test)
    if (env_q && Cout <= 4 * kQBlocks && Cin >= 8 && a.nchunks == 1 && a.CP % 4 == 0 && a.KP % 16 == 0 &&
        ((int64_t)a.total_tiles * a.KP >= 6 * 128 * 256 || env_q >= 2)) {
        constexpr int env_qtarget = 256;
        const int rg = cdiv(a.ntaps * (a.KC / 4), 4);
        const int rgw = rg <= 4 ? 1 : rg <= 8 ? 2 : 3;
        a.nblocks = 1;
        a.DP = a.CoutP = 4 * kQBlocks;
        const size_t bytes = 16 + (size_t)a.KP * 4 + (size_t)a.KP * a.DP * 4 + (((size_t)a.imgs * a.PR * a.PC * a.CP + 3) & ~(size_t)3) * 4 +
                             (size_t)xf_groups * Cin * 8 + (xf_groups ? 16 : 0);
        p->lds_bytes = bytes;
        a.mblocks_per_chunk = cdiv(a.Mchunk, 64);   // slab rows as the 16x16x4 form with MTW = 1 (the reduction reads this format)
        a.Mrows_total = a.mblocks_per_chunk * 64;
        const int qby = cdiv(rg, 4 * rgw);
        const int64_t slab = (int64_t)a.Mrows_total * a.CoutP * 4;
        const int s_cap = (int)std::max<int64_t>(1, (12ll << 20) / slab);
        int S = std::max(1, std::min(std::min(a.total_tiles, s_cap), cdiv(env_qtarget, qby)));
        S = cdiv(a.total_tiles, cdiv(a.total_tiles, S));
        a.S = S;
        p->MTW = 1; p->NTW = 1; p->q_rgw = rgw;
        p->grid_x = S; p->grid_y = qby;
        p->partial_floats = (size_t)S * a.Mrows_total * a.CoutP;
    }
    // XCD-aware order of the workgroups (conv_wgrad_kernel: bx / by) for launches with more than one output block and at least 8 pixel
    // splits: bit-identical, the 20 launches of the 220-view pass 622 -> 607 us, load wait per tile 1185 -> 581 ticks on layers 2 - 3
    // (profiles/r5_wgrad_switches.txt)
    a.xcd_by = (p->grid_y > 1 && a.S >= 8) ? p->grid_y : 0;
    return OCL_OK;
}

int launch_wgrad(const WgradPlan& p, hipStream_t s) {
    const int pf = wgrad_pf_for(p.a.imgs * p.a.PR * p.a.PC * (p.a.KC / 4));
    wgrad_fn_t fn = p.q_rgw ? wgrad_q_fn(p.q_rgw, pf) : wgrad_fn(p.MTW, p.NTW, pf);
    if (p.a.trace) {
        fn = wgrad_trace_fn(p.MTW, p.NTW, pf, p.q_rgw);
        if (!fn) {
            set_error("launch_wgrad: no trace build for MTW=%d NTW=%d PF=%d rgw=%d", p.MTW, p.NTW, pf, p.q_rgw);
            return OCL_ERR_UNSUPPORTED;
        }
        OCL_HIP(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit));
    }
    if (!fn) {
        set_error("launch_wgrad: no kernel for MTW=%d NTW=%d", p.MTW, p.NTW);
        return OCL_ERR_STATE;
    }
    ProfScope ps(PROF_WGRAD, s);
    hipLaunchKernelGGL(fn, p.a.xcd_by > 0 ? dim3(p.grid_x * p.grid_y, 1) : dim3(p.grid_x, p.grid_y), dim3(256), p.lds_bytes, s, p.a);
    OCL_LAUNCH_CHECK();
    return OCL_OK;
}

int wgrad_multi_variant(const WgradPlan& p) {
    if (p.q_rgw || p.a.trace) return -1;
    const int pf = wgrad_pf_for(p.a.imgs * p.a.PR * p.a.PC * (p.a.KC / 4));
    if (p.MTW == 1 && p.NTW == 2) return pf == 4 ? 0 : 1;
    if (p.MTW == 1 && p.NTW == 3) return pf == 4 ? 2 : 3;
    if (p.MTW == 2 && p.NTW == 3 && pf == 8) return 4;
    if (p.MTW == 3 && p.NTW == 2 && pf == 8) return 5;
    return -1;
}
// (the layers of one call share a form set: wgrad_multi_variant(p) / 4)
int launch_wgrad_multi(const WgradPlan* plans, int n, WgradMultiTable* t, hipStream_t s) {
    OCL_REQUIRE(n > 0 && n <= kMaxWgradMulti, "launch_wgrad_multi: %d layers", n);
    const int set = wgrad_multi_variant(plans[0]) / 4;
    std::vector<WgradMultiEntry> host((size_t)n);
    WgradMultiArgs m;
    memset(&m, 0, sizeof(m));
    size_t lds = 0;
    int blocks = 0;
    for (int i = 0; i < n; ++i) {
        memset(&host[i], 0, sizeof(WgradMultiEntry));
        host[i].a = plans[i].a;
        host[i].variant = wgrad_multi_variant(plans[i]);
        host[i].grid_x = plans[i].grid_x;
        OCL_REQUIRE(host[i].variant >= 0 && host[i].variant / 4 == set, "launch_wgrad_multi: layer %d has no form in this merged kernel (MTW=%d NTW=%d)", i,
                    plans[i].MTW, plans[i].NTW);
        m.start[i] = blocks;
        blocks += plans[i].grid_x * plans[i].grid_y;
        lds = std::max(lds, plans[i].lds_bytes);
    }
    m.start[n] = blocks;
    m.n = n;
    // the device table is rewritten only when it changes: in the steady state of a stream every pointer of the pass repeats
    const size_t bytes = host.size() * sizeof(WgradMultiEntry);
    if (!t->dev) OCL_HIP(hipMalloc(&t->dev, sizeof(WgradMultiEntry) * kMaxWgradMulti));
    if (t->host.size() != bytes || memcmp(t->host.data(), host.data(), bytes) != 0) {
        t->host.assign((const unsigned char*)host.data(), (const unsigned char*)host.data() + bytes);
        // (pageable source: the runtime stages it before the call returns, so the vector may change under a later call)
        OCL_HIP(hipMemcpyAsync(t->dev, t->host.data(), bytes, hipMemcpyHostToDevice, s));
        ++t->uploads;
    }
    m.tab = (const WgradMultiEntry*)t->dev;
    ProfScope ps(PROF_WGRAD, s);
    if (set == 0) hipLaunchKernelGGL(conv_wgrad_multi_kernel<0>, dim3(blocks), dim3(256), lds, s, m);
    else hipLaunchKernelGGL(conv_wgrad_multi_kernel<1>, dim3(blocks), dim3(256), lds, s, m);
    OCL_LAUNCH_CHECK();
    return OCL_OK;
}
void wgrad_multi_release(WgradMultiTable* t) {
    if (t->dev) (void)hipFree(t->dev);
    t->dev = nullptr;
    t->host.clear();
}

int launch_wgrad_reduce(const WgradPlan& p, float* grad_oihw, int accumulate, hipStream_t s) {
    const WgradArgs& a = p.a;
    const int cin_real = a.Cin == 4 ? 3 : a.Cin;  // the stem's NHWC4 input carries a zero 4th channel
    const int total = a.ntaps * cin_real * a.Cout;
    ProfScope ps(PROF_WGRAD, s);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(reduce_blocks(a.S, total)), dim3(256), 0, s, a.partial, a.S, a.Mrows_total, a.CoutP,
                       a.mblocks_per_chunk * 64 * p.MTW, a.KC, a.ntaps, cin_real, a.Cout, grad_oihw, accumulate);
    OCL_LAUNCH_CHECK();
    return OCL_OK;
}

void wgrad_reduce_layer(const WgradPlan& p, int64_t partial_off, int64_t grad_off, WgradReduceLayer* d) {
    const WgradArgs& a = p.a;
    d->partial_off = partial_off;
    d->grad_off = grad_off;
    d->S = a.S; d->Mrows_total = a.Mrows_total; d->CoutP = a.CoutP;
    d->mrows_chunk = a.mblocks_per_chunk * 64 * p.MTW;
    d->KC = a.KC; d->ntaps = a.ntaps;
    d->CinReal = a.Cin == 4 ? 3 : a.Cin;
    d->Cout = a.Cout;
    d->block0 = 0;
}

int launch_wgrad_reduce_multi(WgradReduceMulti m, hipStream_t s) {
    OCL_REQUIRE(m.n >= 1 && m.n <= kMaxReduceLayers, "wgrad_reduce_multi: %d layers", m.n);
    int blocks = 0;
    for (int i = 0; i < m.n; ++i) {
        m.L[i].block0 = blocks;
        blocks += reduce_blocks(m.L[i].S, m.L[i].ntaps * m.L[i].CinReal * m.L[i].Cout);
    }
    ProfScope ps(PROF_WGRAD, s);
    hipLaunchKernelGGL(wgrad_reduce_multi_kernel, dim3(blocks), dim3(256), 0, s, m);
    OCL_LAUNCH_CHECK();
    return OCL_OK;
}

// the kernels use up to 160 KB of dynamic LDS: opt in once per process (called by conv_kernels_init)
int wgrad_kernels_init() {
    for (int m = 1; m <= 4; ++m)
        for (int n = 1; n <= 5; ++n)
            for (int pf = 4; pf <= 8; pf += 4)
                OCL_HIP(hipFuncSetAttribute((const void*)wgrad_fn(m, n, pf), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit));
    for (int r = 1; r <= 3; ++r)
        for (int pf = 4; pf <= 8; pf += 4)
            OCL_HIP(hipFuncSetAttribute((const void*)wgrad_q_fn(r, pf), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit));
    OCL_HIP(hipFuncSetAttribute((const void*)conv_wgrad_multi_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit));
    OCL_HIP(hipFuncSetAttribute((const void*)conv_wgrad_multi_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit));
    return OCL_OK;
}

}  // namespace ocl
