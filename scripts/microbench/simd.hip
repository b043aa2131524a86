// Which waves of a 512-thread workgroup share a SIMD?  Waves selected by a mask run a block of
// dependent-free f64 fma (or an integer VALU chain); the rest wait at the barrier.
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 2000
__global__ __launch_bounds__(512) void k(unsigned long long *out, int mask, int kind, int pad) {
    __shared__ int sh[64];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    double T[16], acc = tid * 1e-3;
    int v = tid + pad;
    for (int k = 0; k < 16; k++) T[k] = tid + k;
    if (tid < 64) sh[tid] = tid;
    __syncthreads();
    unsigned long long t0 = clock64(), t1 = t0;
    if (mask & (1 << wave)) {
        if (kind == 0) {
            for (int it = 0; it < N; it++) {
#pragma unroll
                for (int k = 0; k < 16; k++) {
#pragma unroll
                    for (int j = 0; j < 4; j++) T[k] = fma(-1e-9 * k, acc + j, T[k]);
                }
            }
        } else {
            for (int it = 0; it < N; it++) {
#pragma unroll
                for (int k = 0; k < 64; k++) v = v * 3 + (v >> 2);   // dependent integer chain
            }
        }
        t1 = clock64();
    }
    __syncthreads();
    if (lane == 0) out[wave] = t1 - t0;
    for (int k = 0; k < 16; k++) if (T[k] == 1e300 || v == 123456789) out[9] = k;
}
int main() {
    unsigned long long *d; (void)hipMalloc(&d, 16 * 8);
    int masks[] = {0x01, 0x11, 0x03, 0x05, 0x09, 0x21, 0x0f, 0xf0, 0xff, 0x7f, 0x08, 0x88};
    for (int kind = 0; kind < 2; kind++)
        for (int mask : masks) {
            (void)hipMemset(d, 0, 16 * 8);
            hipLaunchKernelGGL(k, dim3(1), dim3(512), 0, 0, d, mask, kind, 1);
            (void)hipDeviceSynchronize();
            unsigned long long h[16]; (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
            printf("%s mask %02x:", kind == 0 ? "64 fma     " : "64 int dep ", mask);
            for (int w = 0; w < 8; w++) if (mask & (1 << w)) printf(" w%d %.0f", w, (double)h[w] / N);
            printf("\n");
        }
    return 0;
}
