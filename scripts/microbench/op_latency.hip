// Dependent-chain latency and independent-chain throughput of the instructions K1's control wave is made of, for ONE
// wave per SIMD (the control wave's situation).  build: hipcc --offload-arch=gfx950 -O3 -o op_latency op_latency.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHAIN(NAME, DECL, OP)                                                             \
    __global__ __launch_bounds__(64) void NAME(long long *out, int n, double seed) {      \
        DECL;                                                                             \
        const long long t0 = clock64();                                                   \
        for (int i = 0; i < n; i++) { OP; OP; OP; OP; OP; OP; OP; OP; }                   \
        const long long t1 = clock64();                                                   \
        if (threadIdx.x == 0) out[0] = t1 - t0;                                           \
        if (a == 1.2345e300) out[1] = 1;                                                  \
    }
CHAIN(k_fma1, double a = seed + threadIdx.x, a = fma(a, 1.0000001, 0.5))
CHAIN(k_add1, double a = seed + threadIdx.x, a = a + 0.5)
CHAIN(k_mul1, double a = seed + threadIdx.x, a = a * 1.0000001)
CHAIN(k_max1, double a = seed + threadIdx.x, a = fmax(a * 1.0000001, 0.25))
CHAIN(k_div1, double a = seed + threadIdx.x, a = 1.5 / a)
CHAIN(k_fma32, float a = (float)seed + threadIdx.x, a = fmaf(a, 1.0000001f, 0.5f))
__global__ __launch_bounds__(64) void k_fma4(long long *out, int n, double seed) {
    double a = seed + threadIdx.x, b = a + 1, c = a + 2, d = a + 3;
    const long long t0 = clock64();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) { a = fma(a, 1.0000001, 0.5); b = fma(b, 1.0000001, 0.5); c = fma(c, 1.0000001, 0.5); d = fma(d, 1.0000001, 0.5); }
    }
    const long long t1 = clock64();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    if (a + b + c + d == 1.2345e300) out[1] = 1;
}
__global__ __launch_bounds__(64) void k_fma8(long long *out, int n, double seed) {
    double a[8];
    for (int k = 0; k < 8; k++) a[k] = seed + threadIdx.x + k;
    const long long t0 = clock64();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int r = 0; r < 8; r++)
#pragma unroll
            for (int k = 0; k < 8; k++) a[k] = fma(a[k], 1.0000001, 0.5);
    }
    const long long t1 = clock64();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    double sm = 0; for (int k = 0; k < 8; k++) sm += a[k];
    if (sm == 1.2345e300) out[1] = 1;
}
__global__ __launch_bounds__(64) void k_keep(long long *out, int n, double seed) {   // the keep_max pattern: compare + two selects
    double bk = seed, key = seed + threadIdx.x;
    int bp = 7, p = threadIdx.x;
    const long long t0 = clock64();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const bool better = key > bk || (key == bk && p < bp);
            bk = better ? key : bk; bp = better ? p : bp;
            key = key * 1.0000001; p += 1;
        }
    }
    const long long t1 = clock64();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    if (bk == 1.2345e300) out[1] = bp;
}
__global__ __launch_bounds__(64) void k_iadd(long long *out, int n, double seed) {
    int a = (int)seed + threadIdx.x;
    const long long t0 = clock64();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) a = (a ^ (a >> 3)) + 12345;
    }
    const long long t1 = clock64();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    if (a == 123456789) out[1] = 1;
}
__global__ __launch_bounds__(64) void k_lds(long long *out, int n, double seed) {   // pointer chase through LDS
    __shared__ int nxt[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) nxt[i] = (i * 7 + 64) & 1023;
    __syncthreads();
    int a = threadIdx.x;
    const long long t0 = clock64();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) a = nxt[a];
    }
    const long long t1 = clock64();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    if (a == 123456789) out[1] = 1;
}
__global__ __launch_bounds__(64) void k_dpp(long long *out, int n, double seed) {   // a DPP reduction level: move + add
    double a = seed + threadIdx.x;
    const long long t0 = clock64();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(a), 0x111, 0xf, 0xf, false);
            const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(a), 0x111, 0xf, 0xf, false);
            a = a + __hiloint2double(hi, lo);
        }
    }
    const long long t1 = clock64();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    if (a == 1.2345e300) out[1] = 1;
}
__global__ __launch_bounds__(64) void k_readlane(long long *out, int n, double seed) {   // VALU -> readlane -> VALU
    int a = (int)seed + threadIdx.x;
    const long long t0 = clock64();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) a = a + __builtin_amdgcn_readlane(a, 5);
    }
    const long long t1 = clock64();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    if (a == 123456789) out[1] = 1;
}
template <typename K> void run(const char *name, K kern, int ops, long long *d) {
    long long h[2];
    const int n = 4000;
    hipLaunchKernelGGL(kern, dim3(1024), dim3(64), 0, 0, d, n, 1.5);   // 1024 single-wave workgroups: one wave per SIMD
    hipDeviceSynchronize();
    hipLaunchKernelGGL(kern, dim3(1024), dim3(64), 0, 0, d, n, 1.5);
    hipDeviceSynchronize();
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("%-44s %7.1f ticks per op\n", name, (double)h[0] / ((double)n * ops));
}
int main() {
    long long *d;
    hipMalloc(&d, 16);
    run("dependent v_fma_f64", k_fma1, 8, d);
    run("dependent v_add_f64", k_add1, 8, d);
    run("dependent v_mul_f64", k_mul1, 8, d);
    run("dependent v_mul_f64 + v_max_f64", k_max1, 8, d);
    run("dependent f64 division", k_div1, 8, d);
    run("dependent v_fma_f32", k_fma32, 8, d);
    run("4 independent fma_f64 chains (per fma)", k_fma4, 32, d);
    run("8 independent fma_f64 chains (per fma)", k_fma8, 64, d);
    run("keep_max step (cmp f64 + selects + mul)", k_keep, 8, d);
    run("dependent integer xor/shift/add (3 ops)", k_iadd, 8, d);
    run("LDS pointer chase (per load)", k_lds, 8, d);
    run("DPP move x2 + v_add_f64 (per level)", k_dpp, 8, d);
    run("v_readlane + v_add (per round trip)", k_readlane, 8, d);
    return 0;
}
