// Probe: throughput of fully divergent (one cache line per lane) L1-resident gathers on gfx950, by access width.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <typename T> __global__ void __launch_bounds__(256) k(const T* __restrict__ tab, int mask, int iters, float* out) {
    unsigned x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    float acc = 0.f;
    for (int i = 0; i < iters; ++i) {
        x = x * 1664525u + 1013904223u;
        const T v = tab[(x >> 8) & mask];
        const float* f = reinterpret_cast<const float*>(&v);
        acc += f[0];
        if (sizeof(T) >= 8) acc += f[1];
        if (sizeof(T) >= 16) acc += f[2] + f[3];
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <typename T> void run(const char* name, int entries) {
    T* tab; float* out; hipMalloc(&tab, entries * sizeof(T)); hipMemset(tab, 0, entries * sizeof(T));
    const int blocks = 256 * 4, iters = 2000;
    hipMalloc(&out, blocks * 256 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL(k<T>, dim3(blocks), dim3(256), 0, 0, tab, entries - 1, iters, out);
        hipEventRecord(b); hipEventSynchronize(b);
    }
    float ms; hipEventElapsedTime(&ms, a, b);
    double waveloads_per_cu = (double)blocks * 4 * iters / 256.0;
    printf("%-10s table %6zu KB: %.3f ms, %.1f ns per wave-load per CU (~%.0f cycles @2.3GHz)\n", name, entries * sizeof(T) / 1024, ms,
           ms * 1e6 / waveloads_per_cu, ms * 1e6 / waveloads_per_cu * 2.3);
    hipFree(tab); hipFree(out);
}
int main() {
    for (int kb : {8, 1600}) {
        run<float>("dword", kb * 1024 / 4); run<float2>("dwordx2", kb * 1024 / 8); run<float4>("dwordx4", kb * 1024 / 16);
    }
    return 0;
}
