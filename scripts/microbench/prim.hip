// Primitive latencies on one CU (512 threads, one workgroup): what a barrier, an LDS hand-over, a
// scalar select chain, a DPP reduction, ... cost in cycles.  Build: hipcc --offload-arch=gfx950 -O3 prim.hip -o prim
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N 4000
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
    asm("s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 0" : "+v"(v));
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__global__ __launch_bounds__(512) void prim(unsigned long long *out, int test, int seed) {
    __shared__ double buf[1024];
    __shared__ int mail[64];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    double acc = tid * 1e-3;
    int iacc = seed;
    double T[16];
    for (int k = 0; k < 16; k++) T[k] = tid + k;
    buf[tid] = tid; buf[tid + 512] = 1.0;
    if (tid < 64) mail[tid] = tid;
    __syncthreads();
    unsigned long long t0 = clock64();
    if (test == 0) {  // empty loop
        for (int it = 0; it < N; it++) { iacc += it; asm volatile("" ::: "memory"); }
    } else if (test == 1) {  // barrier only
        for (int it = 0; it < N; it++) { __syncthreads(); }
    } else if (test == 2) {  // LDS hand-over: wave (it%8) lane 0 writes, barrier, everybody reads + readfirstlane
        for (int it = 0; it < N; it++) {
            if (wave == (it & 7) && lane == 0) mail[0] = iacc + it;
            __syncthreads();
            iacc += __builtin_amdgcn_readfirstlane(mail[0]);
            __syncthreads();
        }
    } else if (test == 3) {  // dependent LDS read chain (latency): idx = mail[idx]
        int idx = lane & 63;
        for (int it = 0; it < N; it++) idx = mail[idx & 63];
        iacc += idx;
    } else if (test == 4) {  // 16-way uniform select chain writing 4 LDS values (row extraction)
        for (int it = 0; it < N; it++) {
            const int rl = (iacc + it) & 15;
#pragma unroll
            for (int ii = 0; ii < 16; ii++) if (ii == rl) buf[lane] = T[ii];
            iacc += rl;
        }
    } else if (test == 5) {  // two u32 DPP reductions + ballot pick
        unsigned v = tid * 2654435761u;
        for (int it = 0; it < N; it++) {
            const unsigned hm = wave_max_u32(v);
            const unsigned lm = wave_max_u32(v == hm ? (unsigned)lane : 0u);
            const unsigned long long mask = __ballot(v == hm && (unsigned)lane == lm);
            iacc += __ffsll((long long)mask);
            v = v * 1664525u + 1013904223u + iacc;
        }
    } else if (test == 6) {  // 16 readlane_f64 + 16 writelane pairs
        for (int it = 0; it < N; it++) {
            int plo = 0, phi = 0;
            const int ql = (iacc + it) & 63;
#pragma unroll
            for (int ii = 0; ii < 16; ii++) {
                const int lo = __builtin_amdgcn_readlane(__double2loint(T[ii]), ql);
                const int hi = __builtin_amdgcn_readlane(__double2hiint(T[ii]), ql);
                asm("v_writelane_b32 %0, %1, %2" : "+v"(plo) : "s"(lo), "n"(ii));
                asm("v_writelane_b32 %0, %1, %2" : "+v"(phi) : "s"(hi), "n"(ii));
            }
            iacc += plo ^ phi;
        }
    } else if (test == 7) {  // f64 division chain (dependent)
        for (int it = 0; it < N; it++) acc = 1.0 / (acc + 1.5);
    } else if (test == 8) {  // 64 independent fma (the rank-1 sweep)
        double a[16];
        for (int k = 0; k < 16; k++) a[k] = k * 0.5;
        for (int it = 0; it < N; it++) {
#pragma unroll
            for (int k = 0; k < 16; k++) {
#pragma unroll
                for (int j = 0; j < 4; j++) T[k] = fma(-a[k], acc + j, T[k]);
            }
        }
        for (int k = 0; k < 16; k++) acc += T[k];
    } else if (test == 9) {  // LDS atomic publish + poll by wave 1 (sequence counter), plus barrier
        for (int it = 0; it < N; it++) {
            if (lane == 0) __hip_atomic_fetch_add(&mail[1], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (wave == 1) {
                while (__hip_atomic_load(&mail[1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < mail[1] * 0 + 8 * (it + 1) + 1)
                    __builtin_amdgcn_s_sleep(1);
            }
            __syncthreads();
        }
    } else if (test == 11) {  // 38 readlanes into SGPRs, consumed by 38 fma
        for (int it = 0; it < N; it++) {
            const int ql = (iacc + it) & 63;
            double a[16];
#pragma unroll
            for (int ii = 0; ii < 16; ii++) {
                const int lo = __builtin_amdgcn_readlane(__double2loint(T[ii]), ql);
                const int hi = __builtin_amdgcn_readlane(__double2hiint(T[ii]), ql);
                a[ii] = __hiloint2double(hi, lo);
            }
#pragma unroll
            for (int ii = 0; ii < 16; ii++) T[ii] = fma(a[ii], 1e-9, T[ii]);
        }
    } else if (test == 12) {  // one lane writes 16 doubles (ds_write_b128 x8), waitcnt, atomic add by lane 0
        for (int it = 0; it < N; it++) {
            const int ql = (iacc + it) & 63;
            if (lane == ql) {
#pragma unroll
                for (int ii = 0; ii < 16; ii++) buf[wave * 17 + ii] = T[ii];
            }
            if (lane == 0) __hip_atomic_fetch_add(&mail[1], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            iacc += it;
        }
    } else if (test == 13) {  // same, then drain (what a following s_waitcnt sees)
        for (int it = 0; it < N; it++) {
            const int ql = (iacc + it) & 63;
            if (lane == ql) {
#pragma unroll
                for (int ii = 0; ii < 16; ii++) buf[wave * 17 + ii] = T[ii];
            }
            if (lane == 0) __hip_atomic_fetch_add(&mail[1], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            iacc += it;
        }
    } else if (test == 14) {  // full-wave ds_write_b64 + waitcnt (for comparison)
        for (int it = 0; it < N; it++) {
            buf[tid] = T[it & 15] + it;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    } else if (test == 15) {  // 16 doubles broadcast read (uniform address), consumed
        for (int it = 0; it < N; it++) {
            const double *src = &buf[(wave * 17 + (it & 1)) & 1023];
            double a[16];
#pragma unroll
            for (int ii = 0; ii < 16; ii++) a[ii] = src[ii];
#pragma unroll
            for (int ii = 0; ii < 16; ii++) T[ii] = fma(a[ii], 1e-9, T[ii]);
        }
    } else if (test == 16) {  // 1-lane write of 16 doubles, then broadcast read-back by the wave, consumed
        for (int it = 0; it < N; it++) {
            const int ql = (iacc + it) & 63;
            double *dst = &buf[wave * 18];
            if (lane == ql) {
#pragma unroll
                for (int ii = 0; ii < 16; ii++) dst[ii] = T[ii];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            double a[16];
#pragma unroll
            for (int ii = 0; ii < 16; ii++) a[ii] = dst[ii];
#pragma unroll
            for (int ii = 0; ii < 16; ii++) T[ii] = fma(a[ii], 1e-9, T[ii]);
            iacc += it;
        }
    } else if (test == 10) {  // s_memtime pair (cost of one profiling mark)
        for (int it = 0; it < N; it++) { iacc += (int)clock64(); }
    }
    unsigned long long t1 = clock64();
    if (lane == 0) out[wave] = t1 - t0;
    if (acc == 12345.678 || iacc == 987654321) out[8] = (unsigned long long)acc + iacc;
    for (int k = 0; k < 16; k++) if (T[k] == 1e300) out[9] = k;
}
int main() {
    unsigned long long *d; hipMalloc(&d, 16 * 8);
    const char *names[] = {"empty loop", "barrier", "LDS hand-over (write,bar,read,bar)", "dependent LDS read", "16-way select chain + ds_write",
                           "2x u32 DPP reduce + ballot pick", "16x readlane_f64 + writelane", "dependent f64 division", "64 fma", "atomic publish + poll + barrier", "clock64", "32 readlane + 16 fma", "1-lane 16 doubles write + atomic", "same + drain", "full-wave ds_write_b64 + drain", "16 doubles broadcast read + 16 fma", "1-lane write 16 + read-back + 16 fma"};
    for (int t = 0; t <= 16; t++) {
        hipMemset(d, 0, 16 * 8);
        hipLaunchKernelGGL(prim, dim3(1), dim3(512), 0, 0, d, t, 1);
        hipDeviceSynchronize();
        unsigned long long h[16]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        printf("%-40s cycles/iter wave0 %.1f wave1 %.1f wave7 %.1f\n", names[t], (double)h[0] / N, (double)h[1] / N, (double)h[7] / N);
    }
    return 0;
}
