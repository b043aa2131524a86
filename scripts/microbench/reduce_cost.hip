// The pieces of a wave-wide selection on one wave per SIMD: cycles per dependent call.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "lp_kernel.hip.h"
using namespace mipx;
template <int MODE>
__global__ __launch_bounds__(64) void k(long long *out, int n, int seed, int *res) {
    const int lane = threadIdx.x;
    unsigned v = (unsigned)(seed + lane * 2654435761u);
    double d = 1.0 + lane * 0.37 + seed;
    int acc = 0;
    const long long t0 = clock64();
    for (int i = 0; i < n; i++) {
        if (MODE == 0) { v = wave_min_u32(v) + lane * 7 + i; }
        else if (MODE == 1) { const unsigned long long m = __ballot(v & 1); v = v * 3 + (unsigned)__ffsll((long long)m); }
        else if (MODE == 2) { const unsigned long long m = __ballot(v & 1); const int l = __ffsll((long long)(m | 1)) - 1; v = v * 3 + (unsigned)__builtin_amdgcn_readlane((int)v, l); }
        else if (MODE == 3) { double km; const int w = wave_argmin_pos(d, (lane << 16) | lane, km); d = d + (w == lane ? 2.5 : 0.0) + km * 1e-9; acc ^= w; }
        else if (MODE == 4) { d = wave_max_f64(d) * 0.999 + lane * 1e-3; }
    }
    const long long t1 = clock64();
    if (lane == 0) { out[0] = t1 - t0; res[0] = acc + (int)v + (int)d; }
}
int main() {
    long long *d, h[2]; int *r;
    hipMalloc(&d, 16); hipMalloc(&r, 8);
    const int n = 20000;
    const char *names[] = {"wave_min_u32 (6 DPP steps + readlane)", "ballot + ffs", "ballot + ffs + readlane", "wave_argmin_pos", "wave_max_f64"};
#define RUN(M) hipLaunchKernelGGL(k<M>, dim3(1024), dim3(64), 0, 0, d, n, 3, r); hipDeviceSynchronize(); hipLaunchKernelGGL(k<M>, dim3(1024), dim3(64), 0, 0, d, n, 3, r); hipDeviceSynchronize(); hipMemcpy(h, d, 16, hipMemcpyDeviceToHost); printf("%-44s %7.1f ticks\n", names[M], (double)h[0] / n);
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4)
    return 0;
}
