// Probe: cost of one wave-wide L1-resident load on gfx950 as a function of the lane address pattern and the access width
// (what bounds k_raymarch's 2x2x2 trilinear footprint: 4 x 16-byte loads per sample).
#include <hip/hip_runtime.h>
#include <cstdio>
template <typename T, int PAT> __global__ void __launch_bounds__(256) k(const char* __restrict__ tab, int iters, float* out) {
    const int lane = threadIdx.x & 63;
    unsigned off;
    unsigned x = threadIdx.x * 2654435761u + 12345u;
    if (PAT == 0) off = 0;                                           // broadcast
    else if (PAT == 1) off = lane * sizeof(T);                       // fully coalesced
    else if (PAT == 2) off = (lane & 7) * 8 + (lane >> 3) * 256;     // 8x8 tile, 1 texel per lane in x, rows of a 32-wide brick
    else if (PAT == 3) off = ((lane & 7) >> 1) * 8 + ((lane >> 3) >> 1) * 256;   // 2x2 lanes share a texel
    else if (PAT == 4) off = (lane & 7) * 8 + (lane >> 3) * 8192;    // 8 z-planes
    else if (PAT == 5) off = (lane & 7) * 16 + (lane >> 3) * 256;    // 2 texels per lane in x
    else if (PAT == 6) off = (lane & 7) * 8 + (lane >> 3) * 264;     // rows + 1 texel shear
    else off = 0;
    float acc = 0.f;
    for (int i = 0; i < iters; ++i) {
        unsigned o;
        if (PAT == 7) { x = x * 1664525u + 1013904223u; o = ((x >> 8) & 1023) * 16; }   // random 16-byte slots in 16 KB
        else o = off + ((i * 8) & 4095);
        const T v = *reinterpret_cast<const T*>(tab + o);
        const float* f = reinterpret_cast<const float*>(&v);
        acc += f[0];
        if (sizeof(T) >= 8) acc += f[1];
        if (sizeof(T) >= 16) acc += f[2] + f[3];
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <typename T, int PAT> void run(const char* name) {
    char* tab; float* out; hipMalloc(&tab, 1 << 20); hipMemset(tab, 0, 1 << 20);
    const int blocks = 256 * 4, iters = 4000;
    hipMalloc(&out, blocks * 256 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL((k<T, PAT>), dim3(blocks), dim3(256), 0, 0, tab, iters, out);
        hipEventRecord(b); hipEventSynchronize(b);
    }
    float ms; hipEventElapsedTime(&ms, a, b);
    double waveloads_per_cu = (double)blocks * 4 * iters / 256.0;
    printf("%-28s %2zu B/lane: %.3f ms, ~%.1f cycles per wave-load per CU @2.3GHz\n", name, sizeof(T), ms, ms * 1e6 / waveloads_per_cu * 2.3);
    hipFree(tab); hipFree(out);
}
#define ALL(P, N) run<float, P>(N); run<float2, P>(N); run<float4, P>(N);
int main() {
    ALL(0, "broadcast") ALL(1, "coalesced") ALL(2, "tile 8x8 rows") ALL(3, "tile 8x8 rows, 2x2 shared") ALL(4, "tile 8x8 z-planes")
    ALL(5, "tile 8x8 rows, 2 texel/lane") ALL(6, "tile 8x8 sheared") ALL(7, "random 16 KB")
    return 0;
}
