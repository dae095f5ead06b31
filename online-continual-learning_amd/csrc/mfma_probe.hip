// mfma_probe -- measurement tool (not part of libocl_hip.so): what does an instruction issued BETWEEN the v_mfma_f32_16x16x4_f32 of ONE wave
// cost the MFMA stream?  One wave per SIMD (256-thread workgroups, one per CU), 16 MFMAs per loop iteration on four independent accumulators,
// and per iteration a fixed number of other instructions placed between them (inline asm: the order written is the order issued):
//   none | v_add_u32 x {4, 16} | s_add_u32 x 16 | ds_read_b128 x {2, 4, 8, 16} whose destination registers are waited for at the END of the
//   iteration (the convolution K loop's pattern: operands of the next round) | the same 4 reads as one block in front of the MFMAs |
//   4 ds_read_b128 + 4 v_add + 2 s_add (what conv_w_kernel issues per 16 MFMAs at MT x NT = 4)
// Prints cycles per MFMA (s_memtime of wave 0 of every workgroup, averaged) and the wall-clock rate.  profiles/r6_mfma_shadow_probe.txt
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MF(ACC) "v_mfma_f32_16x16x4_f32 %[" #ACC "], %[a], %[b], %[" #ACC "]\n\t"
#define M4 MF(c0) MF(c1) MF(c2) MF(c3)
#define VADD "v_add_u32 %[v], %[v], %[vi]\n\t"
#define SADD "s_add_u32 %[sc], %[sc], 1\n\t"
#define RD(D, O) "ds_read_b128 %[" #D "], %[la] offset:" #O "\n\t"
#define WAITL "s_waitcnt lgkmcnt(0)\n\t"
#define GL(D, O) "global_load_dwordx4 %[" #D "], %[ga], off offset:" #O "\n\t"
#define BL(D, O) "buffer_load_dwordx4 %[" #D "], %[bo], %[rs], 0 offen offset:" #O "\n\t"
#define DW(S, O) "ds_write_b128 %[la], %[" #S "] offset:" #O "\n\t"
#define WAITV "s_waitcnt vmcnt(0)\n\t"
#define BLS(D, O) "buffer_load_dwordx4 %[" #D "], %[bo], %[rs], %[so] offen offset:" #O "\n\t"
#define BLD(O) "s_mov_b32 m0, %[m0v]\n\tbuffer_load_dwordx4 %[bo], %[rs], %[so] offen offset:" #O " lds\n\t"
#define OPS : [c0] "+v"(c0), [c1] "+v"(c1), [c2] "+v"(c2), [c3] "+v"(c3), [n0] "+v"(n0), [n1] "+v"(n1), [n2] "+v"(n2), [n3] "+v"(n3), [v] "+v"(v), [sc] "+s"(sc) \
            : [a] "v"(ra), [b] "v"(rb), [vi] "v"(vinc), [la] "v"(lds_a), [ga] "v"(gptr), [bo] "v"(boff), [rs] "s"(rsrc), [so] "s"(soff), [m0v] "s"(m0v) : "memory"

template <int VARIANT>
__global__ void __launch_bounds__(256, 1) probe_kernel(float* out, unsigned long long* ticks, int iters, const float* gsrc) {
    __shared__ __attribute__((aligned(16))) float sm[8192 + 4 * 1024];
    for (int i = threadIdx.x; i < 8192; i += 256) sm[i] = (float)((i * 7) & 15) * 0.0625f;
    __syncthreads();
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    const int lane = threadIdx.x & 63;
    float ra = sm[lane], rb = sm[1024 + lane];
    f32x4 n0 = c0, n1 = c0, n2 = c0, n3 = c0;
    unsigned v = threadIdx.x, vinc = 3, sc = 0;
    const unsigned lds_a = (unsigned)(lane * 16);   // (sm is the kernel's only LDS object: it starts at LDS address 0)
    const float* gptr = gsrc + (size_t)blockIdx.x * 4096 + lane * 4;   // (16 KB per workgroup, L2-resident after the first pass)
    const unsigned boff = (unsigned)(blockIdx.x * 16384 + lane * 16);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gsrc), 0, 0x1ffffff0, 0x00020000);
    unsigned soff = 0;
    const unsigned wv_ = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned m0v = 32768u + wv_ * 4096u;   // (LDS destination of the DMA variant: past the operand area)
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (VARIANT >= 20) soff = (unsigned)(((unsigned)it * 4u * 1024u * 1024u + wv_ * 4096u) & 0x0fffffffu);   // a new 4 KB per wave and iteration, 4 MB apart per workgroup step
        if (VARIANT == 20) asm volatile(MF(c0) BLS(n0, 0) MF(c1) MF(c2) MF(c3) MF(c0) BLS(n1, 1024) MF(c1) MF(c2) MF(c3)
                                        MF(c0) BLS(n2, 2048) MF(c1) MF(c2) MF(c3) MF(c0) BLS(n3, 3072) MF(c1) MF(c2) MF(c3) "s_waitcnt vmcnt(0)\n\t" OPS);
        if (VARIANT == 21) asm volatile(MF(c0) BLS(n0, 0) MF(c1) MF(c2) MF(c3) M4 M4 M4 "s_waitcnt vmcnt(4)\n\t" OPS);   // one streaming load per 16 MFMAs, waited for 4 iterations later
        if (VARIANT == 22) asm volatile(MF(c0) BLS(n0, 0) MF(c1) MF(c2) MF(c3) MF(c0) BLS(n1, 1024) MF(c1) MF(c2) MF(c3)
                                        MF(c0) BLS(n2, 2048) MF(c1) MF(c2) MF(c3) MF(c0) BLS(n3, 3072) MF(c1) MF(c2) MF(c3) "s_waitcnt vmcnt(16)\n\t" OPS);   // four per 16, waited for 4 iterations later
        if (VARIANT == 23) asm volatile(MF(c0) BLD(0) MF(c1) MF(c2) MF(c3) MF(c0) BLD(1024) MF(c1) MF(c2) MF(c3)
                                        MF(c0) BLD(2048) MF(c1) MF(c2) MF(c3) MF(c0) BLD(3072) MF(c1) MF(c2) MF(c3) "s_waitcnt vmcnt(16)\n\t" OPS);   // the same by LDS-DMA
        if (VARIANT == 0) asm volatile(M4 M4 M4 M4 OPS);
        if (VARIANT == 1) asm volatile(M4 VADD M4 VADD M4 VADD M4 VADD OPS);
        if (VARIANT == 2) asm volatile(MF(c0) VADD MF(c1) VADD MF(c2) VADD MF(c3) VADD MF(c0) VADD MF(c1) VADD MF(c2) VADD MF(c3) VADD
                                       MF(c0) VADD MF(c1) VADD MF(c2) VADD MF(c3) VADD MF(c0) VADD MF(c1) VADD MF(c2) VADD MF(c3) VADD OPS);
        if (VARIANT == 3) asm volatile(MF(c0) SADD MF(c1) SADD MF(c2) SADD MF(c3) SADD MF(c0) SADD MF(c1) SADD MF(c2) SADD MF(c3) SADD
                                       MF(c0) SADD MF(c1) SADD MF(c2) SADD MF(c3) SADD MF(c0) SADD MF(c1) SADD MF(c2) SADD MF(c3) SADD OPS);
        if (VARIANT == 9) asm volatile(MF(c0) RD(n0, 0) MF(c1) MF(c2) MF(c3) M4 MF(c0) RD(n1, 4096) MF(c1) MF(c2) MF(c3) M4 WAITL OPS);
        if (VARIANT == 4) asm volatile(MF(c0) RD(n0, 0) MF(c1) MF(c2) MF(c3) MF(c0) RD(n1, 4096) MF(c1) MF(c2) MF(c3)
                                       MF(c0) RD(n2, 8192) MF(c1) MF(c2) MF(c3) MF(c0) RD(n3, 12288) MF(c1) MF(c2) MF(c3) WAITL OPS);
        if (VARIANT == 8) asm volatile(RD(n0, 0) RD(n1, 4096) RD(n2, 8192) RD(n3, 12288) M4 M4 M4 M4 WAITL OPS);
        if (VARIANT == 5) asm volatile(MF(c0) RD(n0, 0) MF(c1) MF(c2) RD(n1, 2048) MF(c3) MF(c0) RD(n2, 4096) MF(c1) MF(c2) RD(n3, 6144) MF(c3)
                                       MF(c0) RD(n0, 8192) MF(c1) MF(c2) RD(n1, 10240) MF(c3) MF(c0) RD(n2, 12288) MF(c1) MF(c2) RD(n3, 14336) MF(c3) WAITL OPS);
        if (VARIANT == 6) asm volatile(MF(c0) RD(n0, 0) MF(c1) RD(n1, 1024) MF(c2) RD(n2, 2048) MF(c3) RD(n3, 3072) MF(c0) RD(n0, 4096) MF(c1) RD(n1, 5120) MF(c2) RD(n2, 6144) MF(c3) RD(n3, 7168)
                                       MF(c0) RD(n0, 8192) MF(c1) RD(n1, 9216) MF(c2) RD(n2, 10240) MF(c3) RD(n3, 11264) MF(c0) RD(n0, 12288) MF(c1) RD(n1, 13312) MF(c2) RD(n2, 14336) MF(c3) RD(n3, 15360) WAITL OPS);
        if (VARIANT == 7) asm volatile(MF(c0) RD(n0, 0) MF(c1) VADD MF(c2) MF(c3) MF(c0) RD(n1, 4096) MF(c1) VADD MF(c2) SADD MF(c3)
                                       MF(c0) RD(n2, 8192) MF(c1) VADD MF(c2) MF(c3) MF(c0) RD(n3, 12288) MF(c1) VADD MF(c2) SADD MF(c3) WAITL OPS);
        if (VARIANT == 11) asm volatile(MF(c0) GL(n0, 0) MF(c1) MF(c2) MF(c3) MF(c0) GL(n1, 1024) MF(c1) MF(c2) MF(c3)
                                        MF(c0) GL(n2, 2048) MF(c1) MF(c2) MF(c3) MF(c0) GL(n3, 3072) MF(c1) MF(c2) MF(c3) WAITV OPS);
        if (VARIANT == 12) asm volatile(MF(c0) BL(n0, 0) MF(c1) MF(c2) MF(c3) MF(c0) BL(n1, 1024) MF(c1) MF(c2) MF(c3)
                                        MF(c0) BL(n2, 2048) MF(c1) MF(c2) MF(c3) MF(c0) BL(n3, 3072) MF(c1) MF(c2) MF(c3) WAITV OPS);
        if (VARIANT == 13) asm volatile(MF(c0) BL(n0, 0) MF(c1) MF(c2) MF(c3) M4 M4 M4 WAITV OPS);   // one load per 16 MFMAs
        if (VARIANT == 14) asm volatile(MF(c0) DW(n0, 0) MF(c1) MF(c2) MF(c3) MF(c0) DW(n1, 4096) MF(c1) MF(c2) MF(c3)
                                        MF(c0) DW(n2, 8192) MF(c1) MF(c2) MF(c3) MF(c0) DW(n3, 12288) MF(c1) MF(c2) MF(c3) WAITL OPS);
        if (VARIANT == 15) asm volatile(MF(c0) BL(n0, 0) MF(c1) MF(c2) MF(c3) M4 M4 M4 OPS);          // one load per 16 MFMAs, never waited for inside the loop
        if (VARIANT == 16) {   // one load per 16 MFMAs, waited for 4 iterations later (counted)
            asm volatile(MF(c0) BL(n0, 0) MF(c1) MF(c2) MF(c3) M4 M4 M4 "s_waitcnt vmcnt(3)\n\t" OPS);
        }
        if (VARIANT == 10) asm volatile(M4 M4 M4 M4 RD(n0, 0) RD(n1, 4096) RD(n2, 8192) RD(n3, 12288) WAITL OPS);   // reads and their wait exposed behind the MFMAs
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
    out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + n0[0] + n1[1] + n2[2] + n3[3] + (float)v + (float)sc;
}

template <int V>
static void run(const char* name, float* out, unsigned long long* ticks, int iters, const float* gsrc) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(probe_kernel<V>, dim3(256), dim3(256), 0, 0, out, ticks, iters, gsrc);
    CK(hipEventRecord(a, 0));
    hipLaunchKernelGGL(probe_kernel<V>, dim3(256), dim3(256), 0, 0, out, ticks, iters, gsrc);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    unsigned long long h[256];
    CK(hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost));
    double avg = 0;
    for (int i = 0; i < 256; ++i) avg += (double)h[i];
    avg /= 256;
    printf("%-62s %7.2f ticks per MFMA (s_memtime)   %7.1f us wall = %5.1f TF/s\n", name, avg / ((double)iters * 16), ms * 1e3,
           256.0 * 4 * iters * 16 * 2048.0 / (ms * 1e-3) * 1e-12);
}

int main() {
    float* out; unsigned long long* ticks; float* gsrc;
    CK(hipMalloc(&out, 256 * 256 * 4)); CK(hipMalloc(&ticks, 256 * 8)); CK(hipMalloc(&gsrc, (size_t)512 << 20)); CK(hipMemset(gsrc, 0, (size_t)512 << 20));
    const int iters = 2000;
    run<0>("16 MFMAs per iteration, nothing else", out, ticks, iters, gsrc);
    run<1>("+ 4 v_add_u32 (one per 4 MFMAs)", out, ticks, iters, gsrc);
    run<2>("+ 16 v_add_u32 (one per MFMA)", out, ticks, iters, gsrc);
    run<3>("+ 16 s_add_u32 (one per MFMA)", out, ticks, iters, gsrc);
    run<9>("+ 2 ds_read_b128 (one per 8 MFMAs), wait at the end", out, ticks, iters, gsrc);
    run<4>("+ 4 ds_read_b128 (one per 4 MFMAs), wait at the end", out, ticks, iters, gsrc);
    run<8>("+ 4 ds_read_b128 in one block in front, wait at the end", out, ticks, iters, gsrc);
    run<10>("+ 4 ds_read_b128 behind the MFMAs, waited for at once", out, ticks, iters, gsrc);
    run<5>("+ 8 ds_read_b128 (one per 2 MFMAs), wait at the end", out, ticks, iters, gsrc);
    run<6>("+ 16 ds_read_b128 (one per MFMA), wait at the end", out, ticks, iters, gsrc);
    run<7>("+ 4 ds_read_b128 + 4 v_add + 2 s_add, wait at the end", out, ticks, iters, gsrc);
    run<11>("+ 4 global_load_dwordx4 (one per 4 MFMAs), wait at the end", out, ticks, iters, gsrc);
    run<12>("+ 4 buffer_load_dwordx4 (one per 4 MFMAs), wait at the end", out, ticks, iters, gsrc);
    run<13>("+ 1 buffer_load_dwordx4 per 16 MFMAs, wait at the end", out, ticks, iters, gsrc);
    run<16>("+ 1 buffer_load_dwordx4 per 16 MFMAs, vmcnt(3)", out, ticks, iters, gsrc);
    run<15>("+ 1 buffer_load_dwordx4 per 16 MFMAs, no wait in the loop", out, ticks, iters, gsrc);
    run<14>("+ 4 ds_write_b128 (one per 4 MFMAs), wait at the end", out, ticks, iters, gsrc);
    run<20>("+ 4 STREAMING buffer_load_dwordx4 per 16 MFMAs, vmcnt(0) at the end", out, ticks, iters, gsrc);
    run<22>("+ 4 STREAMING buffer_load_dwordx4 per 16 MFMAs, waited 4 iterations later", out, ticks, iters, gsrc);
    run<21>("+ 1 STREAMING buffer_load_dwordx4 per 16 MFMAs, waited 4 iterations later", out, ticks, iters, gsrc);
    run<23>("+ 4 STREAMING buffer_load ... lds per 16 MFMAs, waited 4 iterations later", out, ticks, iters, gsrc);
    return 0;
}
