// What does a workgroup barrier cost on gfx950 when one wave works and seven wait?  (K1's iteration has three.)
// build: hipcc --offload-arch=gfx950 -O3 -o barrier_cost barrier_cost.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void k_barrier(long long *out, int iters, int work, int mode) {
    __shared__ volatile int flag;
    __shared__ double sink;
    const int tid = threadIdx.x, wave = tid >> 6;
    if (tid == 0) flag = 0;
    __syncthreads();
    double acc = tid;
    const long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
        if (wave == 0) {   // the "control wave": a dependent chain of fmas
            for (int k = 0; k < work; k++) acc = fma(acc, 1.0000001, 0.5);
        }
        if (mode == 0) {
            __syncthreads();
        } else if (mode == 1) {   // LDS flag: wave 0 publishes, the others poll
            if (wave == 0) {
                if ((tid & 63) == 0) flag = it + 1;
            } else {
                while (flag < it + 1) __builtin_amdgcn_s_sleep(1);
            }
        } else if (mode == 2) {   // two barriers per iteration (hand-over there and back)
            __syncthreads();
            if (wave == 1) acc = fma(acc, 1.0000001, 0.5);
            __syncthreads();
        }
    }
    const long long t1 = clock64();
    if (tid == 0) { out[0] = t1 - t0; sink = acc; }
    if (acc == 12345.678) out[1] = (long long)sink;
}
int main() {
    long long *d, h[2];
    hipMalloc(&d, 16);
    const int iters = 20000;
    for (int mode = 0; mode < 3; mode++)
        for (int work : {0, 16, 64, 256}) {
            hipLaunchKernelGGL(k_barrier, dim3(256), dim3(512), 0, 0, d, iters, work, mode);
            hipDeviceSynchronize();
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            hipEventRecord(a);
            hipLaunchKernelGGL(k_barrier, dim3(256), dim3(512), 0, 0, d, iters, work, mode);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
            printf("mode %d (%s) work %3d fma: %.1f clock64 ticks / iteration, %.1f ns / iteration\n", mode,
                   mode == 0 ? "one barrier" : mode == 1 ? "LDS flag poll" : "two barriers", work, (double)h[0] / iters, ms * 1e6 / iters);
        }
    return 0;
}
