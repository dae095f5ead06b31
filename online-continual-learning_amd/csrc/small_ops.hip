// K6-K13 + K9: buffer gather/scatter, SGD, CE, SupCon, kNN-Shapley, sort, NCM, MIR score, augmentation,
// small MFMA GEMM.  HBM/latency-bound integer+fp32 work: coalesced 16-B accesses, LDS sorts, one
// workgroup per row — none of this is reshaped into GEMMs.
#include "common.h"
#include <math.h>

using namespace ocl;

// =====================================================================================================
// K9 gather / scatter of replay-buffer rows
// =====================================================================================================
template <bool SCATTER>
__global__ void __launch_bounds__(256) rows_copy16(const uint4* __restrict__ src, const int64_t* __restrict__ idx,
                                                   uint4* __restrict__ dst, int64_t units) {
    const int64_t r = blockIdx.x;
    const int64_t row = idx[r];
    const uint4* s = SCATTER ? src + r * units : src + row * units;
    uint4* d = SCATTER ? dst + row * units : dst + r * units;
    for (int64_t u = (int64_t)blockIdx.y * blockDim.x + threadIdx.x; u < units; u += (int64_t)gridDim.y * blockDim.x)
        d[u] = s[u];
}
template <bool SCATTER>
__global__ void __launch_bounds__(256) rows_copy4(const uint32_t* __restrict__ src, const int64_t* __restrict__ idx,
                                                  uint32_t* __restrict__ dst, int64_t units) {
    const int64_t r = blockIdx.x;
    const int64_t row = idx[r];
    const uint32_t* s = SCATTER ? src + r * units : src + row * units;
    uint32_t* d = SCATTER ? dst + row * units : dst + r * units;
    for (int64_t u = (int64_t)blockIdx.y * blockDim.x + threadIdx.x; u < units; u += (int64_t)gridDim.y * blockDim.x)
        d[u] = s[u];
}

template <bool SCATTER>
static int rows_copy(const void* src, const int64_t* idx, int64_t n, int64_t row_bytes, void* dst, void* stream) {
    OCL_REQUIRE(n >= 0 && row_bytes > 0 && (row_bytes % 4) == 0, "rows_copy: n=%lld row_bytes=%lld (must be >0, %%4)",
                (long long)n, (long long)row_bytes);
    if (n == 0) return OCL_OK;
    OCL_REQUIRE(src && idx && dst, "rows_copy: null pointer");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(PROF_KNN, s);
    const bool v16 = (row_bytes % 16) == 0 && (((uintptr_t)src | (uintptr_t)dst) % 16) == 0;
    if (v16) {
        const int64_t units = row_bytes / 16;
        dim3 grid((unsigned)n, (unsigned)max((int64_t)1, min((int64_t)8, (units + 1023) / 1024)));
        hipLaunchKernelGGL(rows_copy16<SCATTER>, grid, dim3(256), 0, s, (const uint4*)src, idx, (uint4*)dst, units);
    } else {
        const int64_t units = row_bytes / 4;
        dim3 grid((unsigned)n, (unsigned)max((int64_t)1, min((int64_t)8, (units + 1023) / 1024)));
        hipLaunchKernelGGL(rows_copy4<SCATTER>, grid, dim3(256), 0, s, (const uint32_t*)src, idx, (uint32_t*)dst, units);
    }
    OCL_LAUNCH_CHECK();
    return OCL_OK;
}

// rows of TWO arrays by one index vector in one launch (replay-buffer images + their labels: every retrieval of the reference is the
// pair buffer_img[indices], buffer_label[indices]): a in 16-byte units over grid.y, b (a few bytes per row) by the first block of the row
__global__ void __launch_bounds__(256) rows_gather_pair(const uint4* __restrict__ a, uint4* __restrict__ da, int64_t units_a,
                                                        const uint32_t* __restrict__ b, uint32_t* __restrict__ db, int64_t units_b,
                                                        const int64_t* __restrict__ idx) {
    const int64_t r = blockIdx.x;
    const int64_t row = idx[r];
    for (int64_t u = (int64_t)blockIdx.y * blockDim.x + threadIdx.x; u < units_a; u += (int64_t)gridDim.y * blockDim.x)
        da[r * units_a + u] = a[row * units_a + u];
    if (blockIdx.y == 0)
        for (int64_t u = threadIdx.x; u < units_b; u += blockDim.x) db[r * units_b + u] = b[row * units_b + u];
}

__global__ void __launch_bounds__(256) gather_u8_hwc_f32_chw(const uint8_t* __restrict__ src, const int64_t* __restrict__ idx,
                                                             int h, int w, int c, float* __restrict__ dst) {
    const int64_t r = blockIdx.x;
    const int64_t hw = (int64_t)h * w, per = hw * c;
    const uint8_t* s = src + idx[r] * per;
    float* d = dst + r * per;
    // output-linear so stores coalesce; the u8 reads hit L1/L2 (3 KB / 21 KB per image)
    for (int64_t o = threadIdx.x + (int64_t)blockIdx.y * blockDim.x; o < per; o += (int64_t)blockDim.x * gridDim.y) {
        const int64_t ch = o / hw, p = o - ch * hw;
        d[o] = (float)s[p * c + ch] / 255.0f;  // ToTensor: .float().div(255)
    }
}

// =====================================================================================================
// K8 SGD (flat)
// =====================================================================================================
__global__ void __launch_bounds__(256) sgd_flat(float* __restrict__ p, const float* __restrict__ g, int64_t n, float lr,
                                                float wd, float gs, float* __restrict__ out) {
    const int64_t n4 = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    float4* o4 = (float4*)(out ? out : p);
    const float4* p4 = (const float4*)p;
    const float4* g4 = (const float4*)g;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 a = p4[i], b = g4[i];
        float4 r;
        r.x = fmaf(-lr, fmaf(wd, a.x, b.x * gs), a.x);
        r.y = fmaf(-lr, fmaf(wd, a.y, b.y * gs), a.y);
        r.z = fmaf(-lr, fmaf(wd, a.z, b.z * gs), a.z);
        r.w = fmaf(-lr, fmaf(wd, a.w, b.w * gs), a.w);
        o4[i] = r;
    }
    float* o = out ? out : p;
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        o[i] = fmaf(-lr, fmaf(wd, p[i], g[i] * gs), p[i]);
}

// =====================================================================================================
// GSS: max_i cos(mem_i, g) over k stored gradient vectors (utils/buffer/buffer_utils.py:51-56, gss_greedy_update.py:79,121)
// =====================================================================================================
// Pass 1: every workgroup owns a contiguous range of the vector; g's chunk stays in registers while the k rows stream past
// it; fp32 lane partials, fp64 from the wave reduction on, one fp64 partial per (workgroup, row) in the workspace
// [blocks][2k + 1] -- no atomics: pass 2 (one wave) sums the workgroups' partials in a fixed order, so the scores (which feed hard
// decisions: `batch_sim < 0`, multinomial weights, the stored buffer_score) are bit-reproducible run to run.
constexpr int kCosChunk = 4;   // float4 per thread and pass
// VEC: rows are 16-byte aligned (n % 4 == 0) -> float4 loads; otherwise scalar loads (any n)
template <bool VEC>
__global__ void __launch_bounds__(256) cosine_partial_kernel(const float* __restrict__ mem, int k, int64_t n,
                                                             const float* __restrict__ g, double* __restrict__ acc) {
    __shared__ double red[4];
    const int64_t units = VEC ? (n >> 2) : n;
    const int64_t per = (units + gridDim.x - 1) / gridDim.x;
    const int64_t beg = (int64_t)blockIdx.x * per, end = min(units, beg + per);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    auto block_add = [&](double v, double* dst) {
        v = wave_sum_d(v);
        __syncthreads();
        if (lane == 0) red[wid] = v;
        __syncthreads();
        if (threadIdx.x == 0) *dst = (red[0] + red[1]) + (red[2] + red[3]);
    };
    acc += (int64_t)blockIdx.x * (2 * k + 1);   // this workgroup's row of partials
    for (int r = -1; r < k; ++r) {   // r = -1: |g|^2
        const float* row = r < 0 ? g : mem + (int64_t)r * n;
        float sd = 0.f, sm = 0.f;
        if (VEC) {
            const float4* m4 = (const float4*)row;
            const float4* g4 = (const float4*)g;
            for (int64_t i = beg + threadIdx.x; i < end; i += 256) {
                const float4 a = m4[i], b = g4[i];
                sd = fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, fmaf(a.w, b.w, sd))));
                sm = fmaf(a.x, a.x, fmaf(a.y, a.y, fmaf(a.z, a.z, fmaf(a.w, a.w, sm))));
            }
        } else {
            for (int64_t i = beg + threadIdx.x; i < end; i += 256) {
                const float a = row[i], b = g[i];
                sd = fmaf(a, b, sd);
                sm = fmaf(a, a, sm);
            }
        }
        if (r < 0) {
            block_add((double)sm, acc + 2 * k);
        } else {
            block_add((double)sd, acc + 2 * r);
            block_add((double)sm, acc + 2 * r + 1);
        }
    }
}
__global__ void __launch_bounds__(64) cosine_finish_kernel(const double* __restrict__ acc, int nblocks, int k, float eps, float* __restrict__ out) {
    const int lane = threadIdx.x;
    float best = -INFINITY;
    const int W = 2 * k + 1;
    auto total = [&](int col) {   // workgroups in ascending order
        double t = 0.0;
        for (int b = 0; b < nblocks; ++b) t += acc[(int64_t)b * W + col];
        return t;
    };
    const float ng = sqrtf((float)total(2 * k));
    for (int r = lane; r < k; r += 64) {
        const float w = fmaxf(sqrtf((float)total(2 * r + 1)) * ng, eps);
        best = fmaxf(best, (float)total(2 * r) / w);
    }
    best = wave_max(best);
    if (lane == 0) out[0] = best;
}

// =====================================================================================================
// K6 cross-entropy: single workgroup, one wave per row, deterministic mean
// =====================================================================================================
__device__ __forceinline__ float ce_row(const float* __restrict__ x, int c, int64_t y, int lane, float* __restrict__ dx,
                                        float scale) {
    float m = -INFINITY;
    for (int j = lane; j < c; j += 64) m = fmaxf(m, x[j]);
    m = wave_max(m);
    float s = 0.f;
    for (int j = lane; j < c; j += 64) s += expf(x[j] - m);
    s = wave_sum(s);
    const float lse = logf(s) + m;
    const float xy = x[y];
    if (dx) {
        const float inv = 1.0f / s;
        for (int j = lane; j < c; j += 64) {
            float p = expf(x[j] - m) * inv;
            dx[j] = (p - (j == (int)y ? 1.f : 0.f)) * scale;
        }
    }
    return lse - xy;
}

__global__ void __launch_bounds__(256) ce_kernel(const float* __restrict__ logits, const int64_t* __restrict__ y, int n, int c,
                                                 int reduction, float* __restrict__ loss_out, float* __restrict__ dlogits) {
    __shared__ float part[4];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const float scale = reduction == 1 ? 1.0f / (float)n : 1.0f;
    float acc = 0.f;
    for (int r = wid; r < n; r += 4) {
        float l = ce_row(logits + (int64_t)r * c, c, y[r], lane, dlogits ? dlogits + (int64_t)r * c : nullptr, scale);
        if (reduction == 0) {
            if (lane == 0) loss_out[r] = l;
        } else {
            acc += l;
        }
    }
    if (reduction == 1) {
        if (lane == 0) part[wid] = acc;
        __syncthreads();
        if (threadIdx.x == 0) loss_out[0] = (part[0] + part[1] + part[2] + part[3]) / (float)n;
    }
}

// Cross-entropy over a column segment (agents/base.py:96-108, the labels trick and the separated softmax): seg[j] in {-1, 0, 1, ...}
// assigns every logit column to a segment (-1: takes no part); row r is a softmax over the columns of its label's segment.
__device__ __forceinline__ float ce_row_seg(const float* __restrict__ x, const int* __restrict__ seg, int c, int64_t y, int lane,
                                            float* __restrict__ dx, float scale) {
    const int sid = seg[y];
    float m = -INFINITY;
    for (int j = lane; j < c; j += 64)
        if (seg[j] == sid) m = fmaxf(m, x[j]);
    m = wave_max(m);
    float s = 0.f;
    for (int j = lane; j < c; j += 64)
        if (seg[j] == sid) s += expf(x[j] - m);
    s = wave_sum(s);
    const float lse = logf(s) + m;
    const float xy = x[y];
    if (dx) {
        const float inv = 1.0f / s;
        for (int j = lane; j < c; j += 64) {
            const float p = seg[j] == sid ? expf(x[j] - m) * inv : 0.f;
            dx[j] = (p - (j == (int)y ? 1.f : 0.f)) * scale;
        }
    }
    return lse - xy;
}

__global__ void __launch_bounds__(256) ce_seg_kernel(const float* __restrict__ logits, const int64_t* __restrict__ y,
                                                     const int* __restrict__ seg, int n, int c, float* __restrict__ loss_out,
                                                     float* __restrict__ dlogits) {
    __shared__ float part[4];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const float scale = 1.0f / (float)n;
    float acc = 0.f;
    for (int r = wid; r < n; r += 4)
        acc += ce_row_seg(logits + (int64_t)r * c, seg, c, y[r], lane, dlogits ? dlogits + (int64_t)r * c : nullptr, scale);
    if (lane == 0) part[wid] = acc;
    __syncthreads();
    if (threadIdx.x == 0) loss_out[0] = (part[0] + part[1] + part[2] + part[3]) / (float)n;
}

// Knowledge-distillation loss (utils/kd_manager.py:6-11): mean_r( -sum_j softmax(t_r/T)_j * log_softmax(s_r/T)_j ) * T^2, and its
// gradient w.r.t. the student scores, (softmax(s/T) - softmax(t/T)) * T / n.  One wave per row.
__global__ void __launch_bounds__(256) kd_kernel(const float* __restrict__ scores, const float* __restrict__ target, int n, int c,
                                                 float T, float* __restrict__ loss_out, float* __restrict__ dscores) {
    __shared__ float part[4];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const float invT = 1.0f / T;
    float acc = 0.f;
    for (int r = wid; r < n; r += 4) {
        const float* s = scores + (int64_t)r * c;
        const float* t = target + (int64_t)r * c;
        float ms = -INFINITY, mt = -INFINITY;
        for (int j = lane; j < c; j += 64) {
            ms = fmaxf(ms, s[j] * invT);
            mt = fmaxf(mt, t[j] * invT);
        }
        ms = wave_max(ms);
        mt = wave_max(mt);
        float ss = 0.f, st = 0.f;
        for (int j = lane; j < c; j += 64) {
            ss += expf(s[j] * invT - ms);
            st += expf(t[j] * invT - mt);
        }
        ss = wave_sum(ss);
        st = wave_sum(st);
        const float lse = logf(ss) + ms, inv_s = 1.0f / ss, inv_t = 1.0f / st;
        float row = 0.f;
        for (int j = lane; j < c; j += 64) {
            const float pt = expf(t[j] * invT - mt) * inv_t;
            row -= pt * (s[j] * invT - lse);
            if (dscores) dscores[(int64_t)r * c + j] = (expf(s[j] * invT - ms) * inv_s - pt) * T / (float)n;
        }
        acc += wave_sum(row);
    }
    if (lane == 0) part[wid] = acc;
    __syncthreads();
    if (threadIdx.x == 0) loss_out[0] = (part[0] + part[1] + part[2] + part[3]) / (float)n * T * T;
}

// K12 MIR: post CE - pre CE per sample
__global__ void __launch_bounds__(256) mir_kernel(const float* __restrict__ pre, const float* __restrict__ post,
                                                  const int64_t* __restrict__ y, int n, int c, float* __restrict__ out) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int r = blockIdx.x * 4 + wid;
    if (r >= n) return;
    const float a = ce_row(pre + (int64_t)r * c, c, y[r], lane, nullptr, 1.f);
    const float b = ce_row(post + (int64_t)r * c, c, y[r], lane, nullptr, 1.f);
    if (lane == 0) out[r] = b - a;
}

// =====================================================================================================
// K7 SupCon (utils/loss.py:19-96), contrast_mode 'all'
// =====================================================================================================
// pass 1: one workgroup per anchor i. G[i][j] = d(mean loss)/d(logit_ij); rowloss[i] = loss_i.
__global__ void __launch_bounds__(256) supcon_rows(const float* __restrict__ feat, const int64_t* __restrict__ y, int bsz, int A,
                                                   int dim, float T, float* __restrict__ G, float* __restrict__ rowloss) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* fi = sm;            // dim
    float* lg = sm + dim;      // A logits
    float* red = lg + A;       // 16
    const int i = blockIdx.x;
    for (int d = threadIdx.x; d < dim; d += blockDim.x) fi[d] = feat[(int64_t)i * dim + d];
    __syncthreads();
    float m = -INFINITY;
    const bool vec = (dim & 3) == 0 && (((uintptr_t)feat) & 15) == 0;
    for (int j = threadIdx.x; j < A; j += blockDim.x) {
        const float* fj = feat + (int64_t)j * dim;
        float dot = 0.f;
        if (vec) {   // 16-byte loads, four independent chains (one chain of `dim` FMAs behind 4-byte loads was most of this kernel's 10 us)
            const float4* fj4 = (const float4*)fj;
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll 8
            for (int d4 = 0; d4 < (dim >> 2); ++d4) {
                const float4 v = fj4[d4];
                const float4 e = *(const float4*)(fi + 4 * d4);
                a0 = fmaf(e.x, v.x, a0); a1 = fmaf(e.y, v.y, a1); a2 = fmaf(e.z, v.z, a2); a3 = fmaf(e.w, v.w, a3);
            }
            dot = (a0 + a1) + (a2 + a3);
        } else {
            for (int d = 0; d < dim; ++d) dot = fmaf(fi[d], fj[d], dot);
        }
        const float l = dot / T;
        lg[j] = l;
        m = fmaxf(m, l);  // reference takes the max over the full row, diagonal included (loss.py:71)
    }
    m = block_max(m, red);
    float se = 0.f;
    for (int j = threadIdx.x; j < A; j += blockDim.x)
        if (j != i) se += expf(lg[j] - m);
    se = block_sum(se, red);
    const float logden = logf(se);
    const int64_t yi = y[i % bsz];
    float sp = 0.f, np = 0.f;
    for (int j = threadIdx.x; j < A; j += blockDim.x) {
        if (j != i && y[j % bsz] == yi) {
            sp += (lg[j] - m) - logden;
            np += 1.f;
        }
    }
    sp = block_sum(sp, red);
    np = block_sum(np, red);
    if (threadIdx.x == 0) rowloss[i] = -(sp / np);  // 0/0 -> NaN like the reference (loss.py:90)
    if (G) {
        const float invA = 1.0f / (float)A, invden = 1.0f / se, invnp = 1.0f / np;
        for (int j = threadIdx.x; j < A; j += blockDim.x) {
            float g = 0.f;
            if (j != i) {
                const float p = expf(lg[j] - m) * invden;
                const float pos = (y[j % bsz] == yi) ? invnp : 0.f;
                g = (p - pos) * invA;
            }
            G[(int64_t)i * A + j] = g;
        }
    }
}
// pass 2: dfeat_i = (1/T) * sum_j (G[i][j] + G[j][i]) f_j ; block 0 also reduces the loss.
// The symmetrised coefficient row is staged in LDS first (the column read G[j][i] is strided: done once, in parallel),
// then threads run over the feature dimension with coalesced reads of f_j.
__global__ void __launch_bounds__(128) supcon_grad(const float* __restrict__ feat, int A, int dim, float T,
                                                   const float* __restrict__ G, const float* __restrict__ rowloss,
                                                   float* __restrict__ loss_out, float* __restrict__ dfeat) {
    extern __shared__ __attribute__((aligned(16))) float cf[];   // A coefficients
    const int i = blockIdx.x;
    if (dfeat) {
        for (int j = threadIdx.x; j < A; j += blockDim.x) cf[j] = G[(int64_t)i * A + j] + G[(int64_t)j * A + i];
        __syncthreads();
        for (int d = threadIdx.x; d < dim; d += blockDim.x) {
            // sixteen independent chains, sixteen loads in flight (a single chain of A = 220 dependent FMAs behind their loads was 18 us of
            // the SCR step's chain; four chains = 55 dependent L2 round trips: 17 us)
            float acc[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[k] = 0.f;
            int j = 0;
            for (; j + 16 <= A; j += 16) {
                float v[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) v[k] = feat[(int64_t)(j + k) * dim + d];
#pragma unroll
                for (int k = 0; k < 16; ++k) acc[k] = fmaf(cf[j + k], v[k], acc[k]);
            }
            for (; j < A; ++j) acc[j & 15] = fmaf(cf[j], feat[(int64_t)j * dim + d], acc[j & 15]);
#pragma unroll
            for (int st = 8; st > 0; st >>= 1)
#pragma unroll
                for (int k = 0; k < st; ++k) acc[k] += acc[k + st];
            dfeat[(int64_t)i * dim + d] = acc[0] / T;
        }
    }
    if (i == 0) {   // mean of the row losses: every thread a strided share, then a tree over the workgroup (was one thread, A dependent steps)
        __syncthreads();   // (cf is free again)
        float t = 0.f;
        for (int j = threadIdx.x; j < A; j += blockDim.x) t += rowloss[j];
        cf[threadIdx.x] = t;
        __syncthreads();
        for (int st = blockDim.x >> 1; st > 0; st >>= 1) {
            if ((int)threadIdx.x < st) cf[threadIdx.x] += cf[threadIdx.x + st];
            __syncthreads();
        }
        if (threadIdx.x == 0) loss_out[0] = cf[0] / (float)A;
    }
}

// =====================================================================================================
// K10 kNN Shapley (utils/buffer/aser_utils.py:7-61,94-116)
// =====================================================================================================
__device__ __forceinline__ bool key_less(float ka, int ia, float kb, int ib) {
    return (ka < kb) || (ka == kb && ia < ib);
}

// in-LDS bitonic sort of P (pow2) (key, idx) pairs, ascending by (key, idx)
__device__ __forceinline__ void bitonic_sort_lds(float* key, int* idx, int P) {
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            __syncthreads();
            for (int t = threadIdx.x; t < (P >> 1); t += blockDim.x) {
                const int lo = ((t / j) * (j << 1)) + (t % j);
                const int hi = lo + j;
                const bool up = ((lo & k) == 0);
                const float ka = key[lo], kb = key[hi];
                const int ia = idx[lo], ib = idx[hi];
                const bool lt = key_less(kb, ib, ka, ia);  // hi < lo
                if (lt == up) {
                    key[lo] = kb; key[hi] = ka;
                    idx[lo] = ib; idx[hi] = ia;
                }
            }
        }
    }
    __syncthreads();
}

__global__ void __launch_bounds__(256) knn_sv_kernel(const float* __restrict__ eval_f, const int64_t* __restrict__ eval_y,
                                                     const float* __restrict__ cand_f, const int64_t* __restrict__ cand_y,
                                                     int n_cand, int dim, int k, int P, float* __restrict__ sv_out,
                                                     int64_t* __restrict__ sorted_idx) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smraw[];
    double* scan = (double*)smraw;              // P doubles (first: 8-B aligned)
    float* key = (float*)(scan + P);            // P
    int* idx = (int*)(key + P);                 // P
    float* ef = (float*)(idx + P);              // dim
    const int e = blockIdx.x;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int d = threadIdx.x; d < dim; d += blockDim.x) ef[d] = eval_f[(int64_t)e * dim + d];
    for (int c = n_cand + threadIdx.x; c < P; c += blockDim.x) {
        key[c] = INFINITY;
        idx[c] = 0x7fffffff;
    }
    __syncthreads();
    // squared Euclidean distance sum((u-v)^2) (utils/utils.py:93-95): one THREAD per candidate, its feature row streamed with four
    // independent 16-byte loads in flight (rows are L2-resident: every workgroup reads the same candidates), the evaluation row
    // broadcast from LDS.  (One wave per candidate with a shuffle reduction made every step a dependent ~0.5 us L2 round trip.)
    if ((dim & 3) == 0 && (((uintptr_t)cand_f) & 15) == 0) {
        const int d4n = dim >> 2;
        for (int c = threadIdx.x; c < n_cand; c += blockDim.x) {
            const float4* cf = (const float4*)(cand_f + (int64_t)c * dim);
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
            int d4 = 0;
            for (; d4 + 4 <= d4n; d4 += 4) {
                const float4 v0 = cf[d4], v1 = cf[d4 + 1], v2 = cf[d4 + 2], v3 = cf[d4 + 3];
                const float4 e0 = *(const float4*)(ef + 4 * d4), e1 = *(const float4*)(ef + 4 * d4 + 4);
                const float4 e2 = *(const float4*)(ef + 4 * d4 + 8), e3 = *(const float4*)(ef + 4 * d4 + 12);
                float t;
                t = e0.x - v0.x; s0 = fmaf(t, t, s0); t = e0.y - v0.y; s1 = fmaf(t, t, s1); t = e0.z - v0.z; s2 = fmaf(t, t, s2); t = e0.w - v0.w; s3 = fmaf(t, t, s3);
                t = e1.x - v1.x; s0 = fmaf(t, t, s0); t = e1.y - v1.y; s1 = fmaf(t, t, s1); t = e1.z - v1.z; s2 = fmaf(t, t, s2); t = e1.w - v1.w; s3 = fmaf(t, t, s3);
                t = e2.x - v2.x; s0 = fmaf(t, t, s0); t = e2.y - v2.y; s1 = fmaf(t, t, s1); t = e2.z - v2.z; s2 = fmaf(t, t, s2); t = e2.w - v2.w; s3 = fmaf(t, t, s3);
                t = e3.x - v3.x; s0 = fmaf(t, t, s0); t = e3.y - v3.y; s1 = fmaf(t, t, s1); t = e3.z - v3.z; s2 = fmaf(t, t, s2); t = e3.w - v3.w; s3 = fmaf(t, t, s3);
            }
            for (; d4 < d4n; ++d4) {
                const float4 v0 = cf[d4];
                const float4 e0 = *(const float4*)(ef + 4 * d4);
                float t;
                t = e0.x - v0.x; s0 = fmaf(t, t, s0); t = e0.y - v0.y; s1 = fmaf(t, t, s1); t = e0.z - v0.z; s2 = fmaf(t, t, s2); t = e0.w - v0.w; s3 = fmaf(t, t, s3);
            }
            key[c] = (s0 + s1) + (s2 + s3);
            idx[c] = c;
        }
    } else {
        for (int c = wid; c < n_cand; c += nw) {
            const float* cf = cand_f + (int64_t)c * dim;
            float s = 0.f;
            for (int d = lane; d < dim; d += 64) {
                const float t = ef[d] - cf[d];
                s = fmaf(t, t, s);
            }
            s = wave_sum(s);
            if (lane == 0) {
                key[c] = s;
                idx[c] = c;
            }
        }
    }
    bitonic_sort_lds(key, idx, P);
    // indicator difference x factor (aser_utils.py:33-50), then reverse cumulative sum (:51-52)
    const int64_t ye = eval_y[e];
    const int N = n_cand;
    for (int j = threadIdx.x; j < P; j += blockDim.x) {
        double v = 0.0;
        if (j < N) {
            const float ind = (cand_y[idx[j]] == ye) ? 1.f : 0.f;
            const float nxt = (j + 1 < N) ? ((cand_y[idx[j + 1]] == ye) ? 1.f : 0.f) : 0.f;
            float numer = (float)(j + 1), denom = (float)(j + 1);
            if (j < N - 1) denom = denom * (float)k;
            if (j >= k && j < N - 1) numer = (float)k;
            if (j == N - 1) numer = 1.f;
            const float factor = numer / denom;
            v = (double)((ind - nxt) * factor);
        }
        scan[j] = v;
    }
    __syncthreads();
    // inclusive suffix scan in double (torch's CPU cumsum accumulates float in double)
    for (int off = 1; off < P; off <<= 1) {
        double add[8];  // P <= 2048, 256 threads: statically indexed so it stays in registers
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int j = threadIdx.x + q * 256;
            add[q] = (j < P && j + off < P) ? scan[j + off] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int j = threadIdx.x + q * 256;
            if (j < P) scan[j] += add[q];
        }
        __syncthreads();
    }
    for (int j = threadIdx.x; j < N; j += blockDim.x) {
        sv_out[(int64_t)e * N + idx[j]] = (float)scan[j];
        if (sorted_idx) sorted_idx[(int64_t)e * N + j] = (int64_t)idx[j];
    }
}

// column reductions over evaluation rows: 32 columns x 8 row lanes per workgroup; the fp64 lane sums are combined in a
// fixed order (deterministic)
__global__ void __launch_bounds__(256) col_reduce_kernel(const float* __restrict__ m, int rows, int cols, int mode,
                                                         float* __restrict__ out) {
    __shared__ double red[8][33];
    const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    double s = 0.0;
    float v = mode == 2 ? -INFINITY : INFINITY;
    if (c < cols)
        for (int r = rl; r < rows; r += 8) {
            const float t = m[(int64_t)r * cols + c];
            s += (double)t;
            v = mode == 2 ? fmaxf(v, t) : fminf(v, t);
        }
    red[rl][cl] = mode <= 1 ? s : (double)v;
    __syncthreads();
    if (rl == 0 && c < cols) {
        if (mode <= 1) {
            const double t = ((red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl])) + ((red[4][cl] + red[5][cl]) + (red[6][cl] + red[7][cl]));
            out[c] = mode == 1 ? (float)(t / (double)rows) : (float)t;
        } else {
            double t = red[0][cl];
            for (int k = 1; k < 8; ++k) t = mode == 2 ? fmax(t, red[k][cl]) : fmin(t, red[k][cl]);
            out[c] = (float)t;
        }
    }
}

__global__ void __launch_bounds__(256) aser_score_kernel(const float* __restrict__ adv, int n_adv, const float* __restrict__ coop,
                                                         int n_coop, int n_cand, int type, float* __restrict__ out) {
    __shared__ double ra[8][33], rc[8][33];
    const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    double sa = 0.0, sc = 0.0;
    float mn = INFINITY, mx = -INFINITY;
    if (c < n_cand) {
        for (int r = rl; r < n_adv; r += 8) {
            const float t = adv[(int64_t)r * n_cand + c];
            sa += (double)t;
            mn = fminf(mn, t);
        }
        if (type != 2)
            for (int r = rl; r < n_coop; r += 8) {
                const float t = coop[(int64_t)r * n_cand + c];
                sc += (double)t;
                mx = fmaxf(mx, t);
            }
    }
    ra[rl][cl] = type == 1 ? (double)mn : sa;
    rc[rl][cl] = type == 1 ? (double)mx : sc;
    __syncthreads();
    if (rl == 0 && c < n_cand) {
        if (type == 1) {
            double a = ra[0][cl], b = rc[0][cl];
            for (int k = 1; k < 8; ++k) { a = fmin(a, ra[k][cl]); b = fmax(b, rc[k][cl]); }
            out[c] = (float)b - (float)a;
        } else {
            const double ta = ((ra[0][cl] + ra[1][cl]) + (ra[2][cl] + ra[3][cl])) + ((ra[4][cl] + ra[5][cl]) + (ra[6][cl] + ra[7][cl]));
            const double tc = ((rc[0][cl] + rc[1][cl]) + (rc[2][cl] + rc[3][cl])) + ((rc[4][cl] + rc[5][cl]) + (rc[6][cl] + rc[7][cl]));
            if (type == 0) out[c] = (float)(tc / (double)n_coop) - (float)(ta / (double)n_adv);
            else out[c] = (float)ta * -1.0f;
        }
    }
}

// descending argsort, single workgroup; ties keep ascending index
__global__ void __launch_bounds__(256) argsort_desc_kernel(const float* __restrict__ v, int n, int P, int64_t* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smraw2[];
    float* key = (float*)smraw2;
    int* idx = (int*)(key + P);
    for (int i = threadIdx.x; i < P; i += blockDim.x) {
        if (i < n) {
            float x = v[i];
            key[i] = isnan(x) ? -INFINITY : -x;  // NaN sorts first in torch's descending order
            idx[i] = i;
        } else {
            key[i] = INFINITY;
            idx[i] = 0x7fffffff;
        }
    }
    bitonic_sort_lds(key, idx, P);
    for (int i = threadIdx.x; i < n; i += blockDim.x) out[i] = (int64_t)idx[i];
}

// =====================================================================================================
// K11 NCM (agents/base.py:121-142, 159-176)
// =====================================================================================================
__global__ void __launch_bounds__(256) ncm_means_kernel(const float* __restrict__ feat, const int64_t* __restrict__ labels, int n,
                                                        int d, const int64_t* __restrict__ class_ids, float* __restrict__ means,
                                                        int32_t* __restrict__ counts) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smraw3[];
    double* acc = (double*)smraw3;      // d
    float* red = (float*)(acc + d);     // 16
    const int c = blockIdx.x;
    const int64_t cls = class_ids[c];
    for (int j = threadIdx.x; j < d; j += blockDim.x) acc[j] = 0.0;
    int cnt = 0;
    for (int i = 0; i < n; ++i) {
        if (labels[i] != cls) continue;  // uniform across the workgroup
        const float* f = feat + (int64_t)i * d;
        float ss = 0.f;
        for (int j = threadIdx.x; j < d; j += blockDim.x) ss = fmaf(f[j], f[j], ss);
        ss = block_sum(ss, red);
        const float nrm = sqrtf(ss);
        for (int j = threadIdx.x; j < d; j += blockDim.x) acc[j] += (double)(f[j] / nrm);
        ++cnt;
    }
    if (threadIdx.x == 0 && counts) counts[c] = cnt;
    if (cnt == 0) return;
    __syncthreads();
    float ss = 0.f;
    for (int j = threadIdx.x; j < d; j += blockDim.x) {
        const float mu = (float)(acc[j] / (double)cnt);
        ss = fmaf(mu, mu, ss);
    }
    ss = block_sum(ss, red);
    const float nrm = sqrtf(ss);
    for (int j = threadIdx.x; j < d; j += blockDim.x) means[(int64_t)c * d + j] = (float)(acc[j] / (double)cnt) / nrm;
}

__global__ void __launch_bounds__(256) ncm_predict_kernel(const float* __restrict__ feat, int d, const float* __restrict__ means,
                                                          int n_cls, int64_t* __restrict__ pred) {
    extern __shared__ __attribute__((aligned(16))) float smf[];
    float* fn = smf;            // d
    float* dist = smf + d;      // n_cls
    float* red = dist + n_cls;  // 16
    const int i = blockIdx.x;
    const float* f = feat + (int64_t)i * d;
    float ss = 0.f;
    for (int j = threadIdx.x; j < d; j += blockDim.x) ss = fmaf(f[j], f[j], ss);
    ss = block_sum(ss, red);
    const float nrm = sqrtf(ss);
    for (int j = threadIdx.x; j < d; j += blockDim.x) fn[j] = f[j] / nrm;
    __syncthreads();
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int c = wid; c < n_cls; c += nw) {
        const float* mu = means + (int64_t)c * d;
        float s = 0.f;
        for (int j = lane; j < d; j += 64) {
            const float t = fn[j] - mu[j];
            s = fmaf(t, t, s);
        }
        s = wave_sum(s);
        if (lane == 0) dist[c] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int best = 0;
        float bv = dist[0];
        for (int c = 1; c < n_cls; ++c)
            if (dist[c] < bv) {
                bv = dist[c];
                best = c;
            }
        pred[i] = best;
    }
}

// =====================================================================================================
// K13 SCR view augmentation (parameters drawn on the host)
// =====================================================================================================
__device__ __forceinline__ float clamp01(float v) { return fminf(fmaxf(v, 0.f), 1.f); }

__device__ __forceinline__ void rgb2hsv(float r, float g, float b, float& h, float& s, float& v) {
    const float mx = fmaxf(r, fmaxf(g, b)), mn = fminf(r, fminf(g, b));
    const float df = mx - mn;
    v = mx;
    s = mx > 0.f ? df / mx : 0.f;
    if (df <= 0.f) {
        h = 0.f;
    } else if (mx == r) {
        h = (g - b) / df;
        if (h < 0.f) h += 6.f;
    } else if (mx == g) {
        h = (b - r) / df + 2.f;
    } else {
        h = (r - g) / df + 4.f;
    }
    h *= (1.0f / 6.0f);  // [0,1)
}
__device__ __forceinline__ void hsv2rgb(float h, float s, float v, float& r, float& g, float& b) {
    const float h6 = (h - floorf(h)) * 6.f;
    const int i = (int)floorf(h6) % 6;
    const float f = h6 - floorf(h6);
    const float p = v * (1.f - s), q = v * (1.f - f * s), t = v * (1.f - (1.f - f) * s);
    switch (i) {
        case 0: r = v; g = t; b = p; break;
        case 1: r = q; g = v; b = p; break;
        case 2: r = p; g = v; b = t; break;
        case 3: r = p; g = q; b = v; break;
        case 4: r = t; g = p; b = v; break;
        default: r = v; g = p; b = q; break;
    }
}

// Per-image augmentation parameters from raw uniform draws (the arithmetic of ScrAugment.sample_params, agents/scr.py, one thread
// per image): 10 crop attempts of (area, log-ratio), first fit wins, centre-crop fallback; position, flip, jitter factors, order, gray.
struct AugCfg {
    float s0, ds, lr0, dlr;            // scale lower bound / width, log-ratio lower bound / width
    float b, c, s, hue, p_jit, p_gray;
    float fb_w, fb_h;                  // fallback crop (whole image with the aspect ratio clamped)
    float j0[4], j1[4];                // jitter factor = j0 + j1 * u  (1-b, 2b; 1-c, 2c; 1-s, 2s; -hue, 2 hue)
};
__global__ void __launch_bounds__(64) aug_params_kernel(const float* __restrict__ u, int n, int h, int w, AugCfg cfg,
                                                        float* __restrict__ params) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* ui = u + (int64_t)i * OCL_AUG_NUNIFORM;
    const float H = (float)h, W = (float)w;
    float cw = 0.f, ch = 0.f;
    bool found = false;
    for (int t = 0; t < 10; ++t) {
        const float area = __fmul_rn(__fmul_rn(__fadd_rn(cfg.s0, __fmul_rn(cfg.ds, ui[t])), H), W);
        const float r = expf(__fadd_rn(cfg.lr0, __fmul_rn(cfg.dlr, ui[10 + t])));
        const float cwt = rintf(sqrtf(__fmul_rn(area, r)));
        const float cht = rintf(sqrtf(area / r));
        const bool fit = cwt > 0.f && cwt <= W && cht > 0.f && cht <= H;
        if (fit && !found) {
            found = true;
            cw = cwt;
            ch = cht;
        }
    }
    const float* e = ui + 20;
    float y0, x0;
    if (found) {
        y0 = fminf(floorf(__fmul_rn(e[0], H - ch + 1.f)), H - 1.f);
        x0 = fminf(floorf(__fmul_rn(e[1], W - cw + 1.f)), W - 1.f);
    } else {
        cw = cfg.fb_w;
        ch = cfg.fb_h;
        y0 = floorf((H - ch) / 2.f);
        x0 = floorf((W - cw) / 2.f);
    }
    float* p = params + (int64_t)i * OCL_AUG_NPARAM;
    p[0] = y0; p[1] = x0; p[2] = ch; p[3] = cw;
    p[4] = e[2] < 0.5f ? 1.f : 0.f;
    p[5] = e[3] < cfg.p_jit ? 1.f : 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) p[6 + k] = __fadd_rn(cfg.j0[k], __fmul_rn(cfg.j1[k], e[4 + k]));
    p[10] = fminf(fmaxf(floorf(__fmul_rn(e[8], 24.f)), 0.f), 23.f);
    p[11] = e[9] < cfg.p_gray ? 1.f : 0.f;
}

__global__ void __launch_bounds__(256) augment_kernel(const float* __restrict__ x, float* __restrict__ out, int h, int w,
                                                      const float* __restrict__ params) {
    const int n = blockIdx.y;
    const int pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= h * w) return;
    const float* pr = params + (int64_t)n * OCL_AUG_NPARAM;
    const float y0 = pr[0], x0 = pr[1], ch = pr[2], cw = pr[3];
    const bool flip = pr[4] > 0.5f, jit = pr[5] > 0.5f, gray = pr[11] > 0.5f;
    const int oy = pix / w, oxr = pix - oy * w;
    const int ox = flip ? (w - 1 - oxr) : oxr;
    // bilinear crop+resize, half-pixel centres, edge clamp
    float sy = y0 + ((float)oy + 0.5f) * (ch / (float)h) - 0.5f;
    float sx = x0 + ((float)ox + 0.5f) * (cw / (float)w) - 0.5f;
    sy = fminf(fmaxf(sy, 0.f), (float)(h - 1));
    sx = fminf(fmaxf(sx, 0.f), (float)(w - 1));
    const int iy0 = (int)floorf(sy), ix0 = (int)floorf(sx);
    const int iy1 = min(iy0 + 1, h - 1), ix1 = min(ix0 + 1, w - 1);
    const float fy = sy - (float)iy0, fx = sx - (float)ix0;
    const float* img = x + (int64_t)n * 3 * h * w;
    float c[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float* p = img + (int64_t)k * h * w;
        const float a = p[iy0 * w + ix0], b = p[iy0 * w + ix1], cc = p[iy1 * w + ix0], dd = p[iy1 * w + ix1];
        const float top = a + (b - a) * fx, bot = cc + (dd - cc) * fx;
        c[k] = top + (bot - top) * fy;
    }
    if (jit) {
        // the 24 permutations of {0:brightness,1:contrast,2:saturation,3:hue} in lexicographic order
        int ord = (int)pr[10];
        unsigned avail = 0x3210u;   // the operations still to run, one nibble each, in ascending order (registers: an indexed array would live in scratch)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int f = s == 0 ? 6 : s == 1 ? 2 : 1;
            const int q = ord / f;
            ord -= q * f;
            const int op = (int)((avail >> (4 * q)) & 15u);
            avail = (avail & ((1u << (4 * q)) - 1u)) | ((avail >> (4 * (q + 1))) << (4 * q));   // drop nibble q
            if (op == 0) {  // additive brightness (kornia 0.4-style): x + (f - 1)
                const float d = pr[6] - 1.f;
                c[0] = clamp01(c[0] + d); c[1] = clamp01(c[1] + d); c[2] = clamp01(c[2] + d);
            } else if (op == 1) {  // multiplicative contrast
                const float f = pr[7];
                c[0] = clamp01(c[0] * f); c[1] = clamp01(c[1] * f); c[2] = clamp01(c[2] * f);
            } else if (op == 2) {
                float hh, ss, vv;
                rgb2hsv(c[0], c[1], c[2], hh, ss, vv);
                ss = clamp01(ss * pr[8]);
                hsv2rgb(hh, ss, vv, c[0], c[1], c[2]);
            } else {
                float hh, ss, vv;
                rgb2hsv(c[0], c[1], c[2], hh, ss, vv);
                hh = hh + pr[9];
                hsv2rgb(hh, ss, vv, c[0], c[1], c[2]);
            }
        }
    }
    if (gray) {
        const float g = 0.299f * c[0] + 0.587f * c[1] + 0.114f * c[2];
        c[0] = c[1] = c[2] = g;
    }
    float* o = out + (int64_t)n * 3 * h * w;
    o[oy * w + oxr] = c[0];
    o[(int64_t)h * w + oy * w + oxr] = c[1];
    o[(int64_t)2 * h * w + oy * w + oxr] = c[2];
}

// =====================================================================================================
// small exact-fp32 MFMA GEMM: one wave per 16x16 tile of C, arbitrary strides
// lane l: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15]; D: col=l&15, row=(l>>4)*4+reg  (v_mfma_f32_16x16x4_f32)
// =====================================================================================================
__global__ void __launch_bounds__(64) gemm_small_kernel(const float* __restrict__ a, int64_t a_rs, int64_t a_cs,
                                                        const float* __restrict__ b, int64_t b_rs, int64_t b_cs,
                                                        float* __restrict__ c, int64_t c_rs, int m, int n, int k,
                                                        const float* __restrict__ bias, int relu, int accumulate) {
    const int lane = threadIdx.x;
    const int r16 = lane & 15, g = lane >> 4;
    const int row0 = blockIdx.y * 16, col0 = blockIdx.x * 16;
    const int ar = row0 + r16, bc = col0 + r16;
    const bool av = ar < m, bv = bc < n;
    const float* ap = a + (int64_t)(av ? ar : 0) * a_rs;
    const float* bp = b + (int64_t)(bv ? bc : 0) * b_cs;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int k0 = 0;
    for (; k0 + 16 <= k; k0 += 16) {
        float av4[4], bv4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int kk = k0 + u * 4 + g;
            av4[u] = av ? ap[(int64_t)kk * a_cs] : 0.f;
            bv4[u] = bv ? bp[(int64_t)kk * b_rs] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av4[u], bv4[u], acc, 0, 0, 0);
    }
    for (; k0 < k; k0 += 4) {
        const int kk = k0 + g;
        const float x = (av && kk < k) ? ap[(int64_t)kk * a_cs] : 0.f;
        const float y = (bv && kk < k) ? bp[(int64_t)kk * b_rs] : 0.f;
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc, 0, 0, 0);
    }
    const int col = col0 + r16;
    if (col < n) {
        const float bs = bias ? bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = row0 + g * 4 + r;
            if (row < m) {
                float v = acc[r] + bs;
                float* cp = c + (int64_t)row * c_rs + col;
                if (accumulate) v += *cp;
                if (relu) v = fmaxf(v, 0.f);
                *cp = v;
            }
        }
    }
}

// =====================================================================================================
// C entry points
// =====================================================================================================
static int next_pow2(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

extern "C" {

int ocl_gather_rows(const void* src, const int64_t* idx, int64_t n, int64_t row_bytes, void* dst, void* stream) {
    return rows_copy<false>(src, idx, n, row_bytes, dst, stream);
}
int ocl_scatter_rows(void* dst, const int64_t* idx, int64_t n, int64_t row_bytes, const void* src, void* stream) {
    return rows_copy<true>(src, idx, n, row_bytes, dst, stream);
}

int ocl_gather_rows_pair(const void* src_a, int64_t row_bytes_a, void* dst_a, const void* src_b, int64_t row_bytes_b, void* dst_b,
                         const int64_t* idx_host, int64_t* idx_dev, int64_t n, void* stream) {
    OCL_REQUIRE(n >= 0 && row_bytes_a > 0 && row_bytes_b > 0 && (row_bytes_a % 4) == 0 && (row_bytes_b % 4) == 0,
                "gather_rows_pair: n=%lld row bytes %lld / %lld (must be > 0, %%4)", (long long)n, (long long)row_bytes_a, (long long)row_bytes_b);
    if (n == 0) return OCL_OK;
    OCL_REQUIRE(src_a && dst_a && src_b && dst_b && idx_dev, "gather_rows_pair: null pointer");
    hipStream_t s = (hipStream_t)stream;
    if (idx_host) {   // the index vector was built on the host (numpy / torch-CPU RNG draws): upload it here, through the staging ring
        int rc = ocl::upload_small(idx_host, (size_t)n * sizeof(int64_t), idx_dev, s);
        if (rc != OCL_OK) return rc;
    }
    const bool v16 = (row_bytes_a % 16) == 0 && (((uintptr_t)src_a | (uintptr_t)dst_a) % 16) == 0;
    if (!v16) {
        int rc = rows_copy<false>(src_a, idx_dev, n, row_bytes_a, dst_a, stream);
        if (rc != OCL_OK) return rc;
        return rows_copy<false>(src_b, idx_dev, n, row_bytes_b, dst_b, stream);
    }
    ProfScope ps(PROF_KNN, s);
    const int64_t units = row_bytes_a / 16;
    dim3 grid((unsigned)n, (unsigned)max((int64_t)1, min((int64_t)8, (units + 1023) / 1024)));
    hipLaunchKernelGGL(rows_gather_pair, grid, dim3(256), 0, s, (const uint4*)src_a, (uint4*)dst_a, units, (const uint32_t*)src_b,
                       (uint32_t*)dst_b, row_bytes_b / 4, idx_dev);
    OCL_LAUNCH_CHECK();
    return OCL_OK;
}

int ocl_gather_u8_hwc_to_f32_chw(const uint8_t* src, const int64_t* idx, int64_t n, int h, int w, int c, float* dst,
                                 void* stream) {
    OCL_REQUIRE(n >= 0 && h > 0 && w > 0 && c > 0, "gather_u8: bad shape");
    if (n == 0) return OCL_OK;
    OCL_REQUIRE(src && idx && dst, "gather_u8: null pointer");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(PROF_KNN, s);
    const int64_t per = (int64_t)h * w * c;
    dim3 grid((unsigned)n, (unsigned)max((int64_t)1, min((int64_t)16, (per + 2047) / 2048)));
    hipLaunchKernelGGL(gather_u8_hwc_f32_chw, grid, dim3(256), 0, s, src, idx, h, w, c, dst);
    OCL_LAUNCH_CHECK();
    return OCL_OK;
}

int ocl_sgd_step(float* params, const float* grads, int64_t n, float lr, float weight_decay, float grad_scale, float* out,
                 void* stream) {
    OCL_REQUIRE(params && grads && n > 0, "sgd: null pointer or n<=0");
    // a backward whose one-pass BatchNorm timed out has poisoned `grads` with NaN: refuse the step instead of applying it (the word
    // is read without synchronising; a time-out that lands after this check is caught by the next forward, before its step)
    if (int arc = ocl::check_async_error("sgd_step")) return arc;
    OCL_REQUIRE((((uintptr_t)params | (uintptr_t)grads | (uintptr_t)out) % 16) == 0, "sgd: pointers must be 16-B aligned");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(PROF_BN, s);
    const int blocks = (int)min((int64_t)2048, (n / 4 + 255) / 256 + 1);
    hipLaunchKernelGGL(sgd_flat, dim3(blocks), dim3(256), 0, s, params, grads, n, lr, weight_decay, grad_scale, out);
    OCL_LAUNCH_CHECK();
    return OCL_OK;
}

constexpr int kCosMaxBlocks = 512;
int64_t ocl_cosine_max_workspace_bytes(int k) { return (int64_t)kCosMaxBlocks * (2 * (int64_t)k + 1) * 8; }

int ocl_cosine_max(const float* mem, int k, int64_t n, const float* g, float eps, float* out, void* workspace, void* stream) {
    OCL_REQUIRE(mem && g && out && workspace && k > 0 && n > 0, "cosine_max: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(PROF_KNN, s);
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(kCosMaxBlocks, (n / 4 + 256 * kCosChunk - 1) / (256 * kCosChunk)));
    if (n % 4 == 0 && ((uintptr_t)mem % 16) == 0 && ((uintptr_t)g % 16) == 0)
        hipLaunchKernelGGL(cosine_partial_kernel<true>, dim3(blocks), dim3(256), 0, s, mem, k, n, g, (double*)workspace);
    else
        hipLaunchKernelGGL(cosine_partial_kernel<false>, dim3(blocks), dim3(256), 0, s, mem, k, n, g, (double*)workspace);
    OCL_LAUNCH_CHECK();
    hipLaunchKernelGGL(cosine_finish_kernel, dim3(1), dim3(64), 0, s, (const double*)workspace, blocks, k, eps, out);
    OCL_LAUNCH_CHECK();
    return OCL_OK;
}

int ocl_ce_fwd_bwd(const float* logits, const int64_t* y, int n, int c, int reduction, float* loss_out, float* dlogits,
                   void* stream) {
    OCL_REQUIRE(logits && y && loss_out && n > 0 && c > 0, "ce: bad arguments");
    OCL_REQUIRE(reduction == 0 || reduction == 1, "ce: reduction must be 0 (none) or 1 (mean)");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(PROF_HEAD, s);
    hipLaunchKernelGGL(ce_kernel, dim3(1), dim3(256), 0, s, logits, y, n, c, reduction, loss_out, dlogits);
    OCL_LAUNCH_CHECK();
    return OCL_OK;
}

int ocl_ce_segmented_fwd_bwd(const float* logits, const int64_t* y, const int32_t* seg, int n, int c, float* loss_out, float* dlogits,
                             void* stream) {
    OCL_REQUIRE(logits && y && seg && loss_out && n > 0 && c > 0, "ce_segmented: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(PROF_HEAD, s);
    hipLaunchKernelGGL(ce_seg_kernel, dim3(1), dim3(256), 0, s, logits, y, seg, n, c, loss_out, dlogits);
    OCL_LAUNCH_CHECK();
    return OCL_OK;
}

int ocl_kd_fwd_bwd(const float* scores, const float* target_scores, int n, int c, float T, float* loss_out, float* dscores,
                   void* stream) {
    OCL_REQUIRE(scores && target_scores && loss_out && n > 0 && c > 0 && T > 0.f, "kd: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(PROF_HEAD, s);
    hipLaunchKernelGGL(kd_kernel, dim3(1), dim3(256), 0, s, scores, target_scores, n, c, T, loss_out, dscores);
    OCL_LAUNCH_CHECK();
    return OCL_OK;
}

int ocl_mir_scores(const float* logits_pre, const float* logits_post, const int64_t* y, int n, int c, float* scores_out,
                   void* stream) {
    OCL_REQUIRE(logits_pre && logits_post && y && scores_out && n > 0 && c > 0, "mir_scores: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(PROF_KNN, s);
    hipLaunchKernelGGL(mir_kernel, dim3(cdiv(n, 4)), dim3(256), 0, s, logits_pre, logits_post, y, n, c, scores_out);
    OCL_LAUNCH_CHECK();
    return OCL_OK;
}

int64_t ocl_supcon_workspace_bytes(int n_anchor) {
    return ((int64_t)n_anchor * n_anchor + n_anchor + 64) * (int64_t)sizeof(float);
}

int ocl_supcon_fwd_bwd(const float* feat, const int64_t* y, int bsz, int n_views, int dim, float temperature, float* loss_out,
                       float* dfeat, void* workspace, void* stream) {
    OCL_REQUIRE(feat && y && loss_out && workspace, "supcon: null pointer");
    OCL_REQUIRE(bsz > 0 && n_views > 0 && dim > 0 && temperature > 0.f, "supcon: bad shape/temperature");
    const int A = bsz * n_views;
    OCL_REQUIRE(A <= 8192 && dim <= 4096, "supcon: A=%d dim=%d too large", A, dim);
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(PROF_HEAD, s);
    float* G = (float*)workspace;
    float* rowloss = G + (int64_t)A * A;
    const size_t sm = (size_t)(dim + A + 16) * sizeof(float);
    hipLaunchKernelGGL(supcon_rows, dim3(A), dim3(256), sm, s, feat, y, bsz, A, dim, temperature, G, rowloss);
    OCL_LAUNCH_CHECK();
    hipLaunchKernelGGL(supcon_grad, dim3(dfeat ? A : 1), dim3(128), (size_t)std::max(A, 128) * sizeof(float), s, feat, A, dim, temperature, G, rowloss,
                       loss_out, dfeat);
    OCL_LAUNCH_CHECK();
    return OCL_OK;
}

int ocl_knn_sv(const float* eval_f, const int64_t* eval_y, int n_eval, const float* cand_f, const int64_t* cand_y, int n_cand,
               int dim, int k, float* sv_out, int64_t* sorted_idx, void* stream) {
    OCL_REQUIRE(n_eval >= 0 && n_cand >= 0 && dim > 0 && k > 0, "knn_sv: bad sizes n_eval=%d n_cand=%d dim=%d k=%d", n_eval,
                n_cand, dim, k);
    if (n_eval == 0 || n_cand == 0) return OCL_OK;
    OCL_REQUIRE(eval_f && eval_y && cand_f && cand_y && sv_out, "knn_sv: null pointer");
    OCL_REQUIRE(n_cand <= OCL_KNN_MAX_CAND, "knn_sv: n_cand=%d exceeds OCL_KNN_MAX_CAND=%d", n_cand, OCL_KNN_MAX_CAND);
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(PROF_KNN, s);
    const int P = next_pow2(n_cand);
    const size_t sm = (size_t)P * (8 + 4 + 4) + (size_t)dim * 4;
    hipLaunchKernelGGL(knn_sv_kernel, dim3(n_eval), dim3(256), sm, s, eval_f, eval_y, cand_f, cand_y, n_cand, dim, k, P, sv_out,
                       sorted_idx);
    OCL_LAUNCH_CHECK();
    return OCL_OK;
}

int ocl_col_reduce(const float* m, int rows, int cols, int mode, float* out, void* stream) {
    OCL_REQUIRE(m && out && rows > 0 && cols > 0 && mode >= 0 && mode <= 3, "col_reduce: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(PROF_KNN, s);
    hipLaunchKernelGGL(col_reduce_kernel, dim3(cdiv(cols, 32)), dim3(256), 0, s, m, rows, cols, mode, out);
    OCL_LAUNCH_CHECK();
    return OCL_OK;
}

int ocl_aser_score(const float* sv_adv, int n_adv, const float* sv_coop, int n_coop, int n_cand, int type, float* out,
                   void* stream) {
    OCL_REQUIRE(sv_adv && out && n_adv > 0 && n_cand > 0 && type >= 0 && type <= 2, "aser_score: bad arguments");
    OCL_REQUIRE(type == 2 || (sv_coop && n_coop > 0), "aser_score: cooperative matrix required for type %d", type);
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(PROF_KNN, s);
    hipLaunchKernelGGL(aser_score_kernel, dim3(cdiv(n_cand, 32)), dim3(256), 0, s, sv_adv, n_adv, sv_coop, n_coop, n_cand, type,
                       out);
    OCL_LAUNCH_CHECK();
    return OCL_OK;
}

int ocl_argsort_desc(const float* v, int n, int64_t* idx_out, void* stream) {
    OCL_REQUIRE(n >= 0, "argsort: n<0");
    if (n == 0) return OCL_OK;
    OCL_REQUIRE(v && idx_out, "argsort: null pointer");
    OCL_REQUIRE(n <= OCL_SORT_MAX, "argsort: n=%d exceeds OCL_SORT_MAX=%d", n, OCL_SORT_MAX);
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(PROF_KNN, s);
    const int P = next_pow2(n);
    hipLaunchKernelGGL(argsort_desc_kernel, dim3(1), dim3(256), (size_t)P * 8, s, v, n, P, idx_out);
    OCL_LAUNCH_CHECK();
    return OCL_OK;
}

int ocl_ncm_class_means(const float* feat, const int64_t* labels, int n, int d, const int64_t* class_ids, int n_cls,
                        float* means_out, int32_t* counts_out, void* stream) {
    OCL_REQUIRE(feat && labels && class_ids && means_out && n >= 0 && d > 0 && n_cls > 0, "ncm_means: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(PROF_KNN, s);
    hipLaunchKernelGGL(ncm_means_kernel, dim3(n_cls), dim3(256), (size_t)d * 8 + 64, s, feat, labels, n, d, class_ids, means_out,
                       counts_out);
    OCL_LAUNCH_CHECK();
    return OCL_OK;
}

int ocl_ncm_predict(const float* feat, int n, int d, const float* means, int n_cls, int64_t* pred_out, void* stream) {
    OCL_REQUIRE(n >= 0 && d > 0 && n_cls > 0, "ncm_predict: bad sizes");
    if (n == 0) return OCL_OK;
    OCL_REQUIRE(feat && means && pred_out, "ncm_predict: null pointer");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(PROF_KNN, s);
    hipLaunchKernelGGL(ncm_predict_kernel, dim3(n), dim3(256), (size_t)(d + n_cls + 16) * 4, s, feat, d, means, n_cls, pred_out);
    OCL_LAUNCH_CHECK();
    return OCL_OK;
}

int ocl_scr_augment(const float* x, float* out, int n, int h, int w, const float* params, void* stream) {
    OCL_REQUIRE(n >= 0 && h > 0 && w > 0, "augment: bad shape");
    if (n == 0) return OCL_OK;
    OCL_REQUIRE(x && out && params && x != out, "augment: null pointer or in-place");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(PROF_BN, s);
    hipLaunchKernelGGL(augment_kernel, dim3(cdiv(h * w, 256), n), dim3(256), 0, s, x, out, h, w, params);
    OCL_LAUNCH_CHECK();
    return OCL_OK;
}

int ocl_scr_augment_uniform(const float* x, float* out, int n, int h, int w, const float* u, const double* cfg12, float* params,
                            void* stream) {
    OCL_REQUIRE(n >= 0 && h > 0 && w > 0, "augment: bad shape");
    if (n == 0) return OCL_OK;
    OCL_REQUIRE(x && out && u && cfg12 && params && x != out, "augment: null pointer or in-place");
    AugCfg cfg;
    // derived constants in double, rounded once (what the host arithmetic does with Python floats applied to fp32 tensors)
    cfg.s0 = (float)cfg12[0]; cfg.ds = (float)(cfg12[1] - cfg12[0]);
    cfg.lr0 = (float)log(cfg12[2]); cfg.dlr = (float)(log(cfg12[3]) - log(cfg12[2]));
    cfg.b = (float)cfg12[4]; cfg.c = (float)cfg12[5]; cfg.s = (float)cfg12[6]; cfg.hue = (float)cfg12[7];
    cfg.p_jit = (float)cfg12[8]; cfg.p_gray = (float)cfg12[9];
    cfg.fb_w = (float)cfg12[10]; cfg.fb_h = (float)cfg12[11];
    for (int k = 0; k < 3; ++k) {
        cfg.j0[k] = (float)(1.0 - cfg12[4 + k]);
        cfg.j1[k] = (float)(2.0 * cfg12[4 + k]);
    }
    cfg.j0[3] = (float)(-cfg12[7]);
    cfg.j1[3] = (float)(2.0 * cfg12[7]);
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(PROF_BN, s);
    hipLaunchKernelGGL(aug_params_kernel, dim3(cdiv(n, 64)), dim3(64), 0, s, u, n, h, w, cfg, params);
    OCL_LAUNCH_CHECK();
    hipLaunchKernelGGL(augment_kernel, dim3(cdiv(h * w, 256), n), dim3(256), 0, s, x, out, h, w, (const float*)params);
    OCL_LAUNCH_CHECK();
    return OCL_OK;
}

int ocl_gemm_small(const float* a, int64_t a_rs, int64_t a_cs, const float* b, int64_t b_rs, int64_t b_cs, float* c,
                   int64_t c_rs, int m, int n, int k, const float* bias, int relu, int accumulate, void* stream) {
    OCL_REQUIRE(a && b && c && m > 0 && n > 0 && k > 0, "gemm_small: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(PROF_HEAD, s);
    hipLaunchKernelGGL(gemm_small_kernel, dim3(cdiv(n, 16), cdiv(m, 16)), dim3(64), 0, s, a, a_rs, a_cs, b, b_rs, b_cs, c, c_rs, m,
                       n, k, bias, relu, accumulate);
    OCL_LAUNCH_CHECK();
    return OCL_OK;
}

}  // extern "C"
