// How long does ONE workgroup per CU need to pull a 256 KiB block (K1's anchor: the set-up of a chain) into
// registers, depending on where the block lives?  512 threads, 32 x 16 B per thread, all loads issued before the
// first use; blocks at random 256 KiB-aligned offsets of a buffer of the given size (one pass over a 2 GiB
// buffer = HBM + cold TLB; a 64 MiB buffer read twice = the memory-side cache; 2 MiB = the L2).
// build: hipcc --offload-arch=gfx950 -O3 -o anchor_load anchor_load.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(512) void k_load(const char *buf, const unsigned *block, long long *cycles, double *sink, int nt) {
    const d2 *src = reinterpret_cast<const d2 *>(buf + (size_t)block[blockIdx.x] * 262144);
    const int tid = threadIdx.x;
    d2 v[32];
    const long long t0 = clock64();
#pragma unroll
    for (int k = 0; k < 32; k++) v[k] = nt ? __builtin_nontemporal_load(src + tid + 512 * k) : src[tid + 512 * k];
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < 32; k++) acc += v[k].x + v[k].y;
    __syncthreads();
    const long long t1 = clock64();
    if (tid == 0) cycles[blockIdx.x] = t1 - t0;
    if (acc == 1.2345e300) sink[0] = acc;
}
int main() {
    const size_t big = (size_t)2 << 30;
    char *buf; unsigned *dblk; long long *dcy; double *sink;
    hipMalloc(&buf, big); hipMemset(buf, 0, big);
    const int G = 8192;
    hipMalloc(&dblk, G * 4); hipMalloc(&dcy, G * 8); hipMalloc(&sink, 8);
    std::vector<unsigned> blk(G); std::vector<long long> cy(G);
    struct { const char *name; size_t bytes; int passes; } cfg[] = {
        {"2 GiB buffer (HBM, one block per workgroup)", big, 1}, {"256 MiB buffer, second pass", (size_t)256 << 20, 2},
        {"64 MiB buffer, second pass (memory-side cache)", (size_t)64 << 20, 2}, {"2 MiB buffer (L2)", (size_t)2 << 20, 2}};
    for (auto &c : cfg)
        for (int grid : {256, 8192}) {
            const unsigned nb = (unsigned)(c.bytes / 262144);
            unsigned x = 12345u;
            for (int i = 0; i < G; i++) { x = x * 1664525u + 1013904223u; blk[i] = (nb >= (unsigned)G) ? (unsigned)((i * 2654435761u) % nb) : (x >> 8) % nb; }
            if (nb >= (unsigned)G) for (int i = 0; i < G; i++) blk[i] = (unsigned)(((unsigned long long)i * 7919ull) % nb);   // distinct blocks
            hipMemcpy(dblk, blk.data(), G * 4, hipMemcpyHostToDevice);
            for (int p = 0; p < c.passes; p++) { hipLaunchKernelGGL(k_load, dim3(grid), dim3(512), 0, 0, buf, dblk, dcy, sink, 0); hipDeviceSynchronize(); }
            hipMemcpy(cy.data(), dcy, grid * 8, hipMemcpyDeviceToHost);
            double s = 0; for (int i = 0; i < grid; i++) s += cy[i];
            printf("%-52s grid %5d: %8.0f ticks per 256 KiB block (%.1f B/tick per CU)\n", c.name, grid, s / grid, 262144.0 / (s / grid));
        }
    return 0;
}
