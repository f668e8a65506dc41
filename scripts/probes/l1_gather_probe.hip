// l1_gather_probe.hip -- what a wave-wide gather costs in the CU's L1 (TCP) when every line hits: W bytes per lane (4 / 8 / 16), the 64 lanes
// spread over L distinct 128-byte lines (1 .. 64), lanes of one line adjacent ("grouped") or interleaved (lane % L).  16 waves per CU, four
// loads in flight per wave (like k_raymarch's trilinear footprint), table = 16 KB per CU region (L1-resident).  Question behind it: is
// k_raymarch's ~26 L1 cycles per wave-load a per-LINE cost (then a blocked brick layout that packs a 2x2x2 footprint's neighbours into
// fewer lines would pay) or a per-lane/per-instruction cost (then it would not)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int W> struct Vec;
template <> struct Vec<4>  { using T = unsigned; };
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <> struct Vec<8>  { using T = u32x2; };
template <> struct Vec<16> { using T = u32x4; };

template <int W>
__device__ inline unsigned fold(const typename Vec<W>::T& v)
{
    if constexpr (W == 4) return v;
    else if constexpr (W == 8) return v.x ^ v.y;
    else return v.x ^ v.y ^ v.z ^ v.w;
}

template <int W, int INTERLEAVED>
__global__ void __launch_bounds__(1024) k_l1(const unsigned char* __restrict__ table, int L, int iters, long long* out, unsigned* check)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned char* base = table + (size_t)(blockIdx.x & 255) * 16384;
    const int per = 64 / L;                                   // lanes per line
    const int lsel = INTERLEAVED ? (lane % L) : (lane / per);
    const int slot = INTERLEAVED ? (lane / L) : (lane % per);
    const unsigned off_in_line = (unsigned)(slot * W) & 127u;
    unsigned acc = 0;
    // warm the L1
    for (int i = threadIdx.x; i < 16384 / 4; i += 1024) acc ^= ((const unsigned*)base)[i];
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        typename Vec<W>::T v[4];
        const unsigned char* p[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned line = (unsigned)(it * 7 + wave * 5 + k * 31 + lsel) & 127u;
            p[k] = base + line * 128u + off_in_line;
        }
        if constexpr (W == 4)
            asm volatile("global_load_dword %0, %4, off\n\tglobal_load_dword %1, %5, off\n\tglobal_load_dword %2, %6, off\n\tglobal_load_dword %3, %7, off\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]) : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]) : "memory");
        else if constexpr (W == 8)
            asm volatile("global_load_dwordx2 %0, %4, off\n\tglobal_load_dwordx2 %1, %5, off\n\tglobal_load_dwordx2 %2, %6, off\n\tglobal_load_dwordx2 %3, %7, off\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]) : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]) : "memory");
        else
            asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %5, off\n\tglobal_load_dwordx4 %2, %6, off\n\tglobal_load_dwordx4 %3, %7, off\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]) : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]) : "memory");
#pragma unroll
        for (int k = 0; k < 4; ++k) acc += fold<W>(v[k]);
    }
    const long long t1 = clock64();
    if (lane == 0) out[blockIdx.x * 16 + wave] = t1 - t0;
    if (acc == 0x12345u) check[0] = acc;
}

template <int W, int INTERLEAVED>
void run(const unsigned char* d_table, long long* d_out, unsigned* d_chk)
{
    const int iters = 2000, nblk = 256;
    printf("%2d B/lane, %s:", W, INTERLEAVED ? "interleaved" : "grouped    ");
    for (int L = 1; L <= 64; L *= 2) {
        hipLaunchKernelGGL((k_l1<W, INTERLEAVED>), dim3(nblk), dim3(1024), 0, 0, d_table, L, iters, d_out, d_chk);
        std::vector<long long> h(nblk * 16);
        (void)hipMemcpy(h.data(), d_out, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
        double mx = 0;
        for (size_t i = 0; i < h.size(); ++i) mx = mx > (double)h[i] ? mx : (double)h[i];
        printf("  L=%-2d %6.2f", L, mx / (16.0 * 4 * iters));
    }
    printf("   (cycles per wave-load per CU, 16 waves)\n");
}

int main()
{
    unsigned char* d_table; long long* d_out; unsigned* d_chk;
    (void)hipMalloc(&d_table, 256 * 16384); (void)hipMemset(d_table, 1, 256 * 16384);
    (void)hipMalloc(&d_out, 256 * 16 * sizeof(long long)); (void)hipMalloc(&d_chk, 16);
    run<4, 0>(d_table, d_out, d_chk);  run<4, 1>(d_table, d_out, d_chk);
    run<8, 0>(d_table, d_out, d_chk);  run<8, 1>(d_table, d_out, d_chk);
    run<16, 0>(d_table, d_out, d_chk); run<16, 1>(d_table, d_out, d_chk);
    return 0;
}
