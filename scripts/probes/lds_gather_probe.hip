// lds_gather_probe.hip -- cost of the cube-map footprint gather if the (R8) cube map lived in LDS: per wave, 4 x ds_read_u8_d16 (the
// bilinear quad: offsets 0, 1, pitch, pitch+1 off one address) with (a) the address pattern of an 8x8-voxel wave tile whose
// neighbouring voxels land ~5 texels apart, (b) fully random addresses, (c) one address for all lanes; 12 waves per CU (3 per SIMD),
// table = 6 x 130 x 130 bytes.  Also checks that ds_read_u8_d16 into a register preset to 0x4B000000 yields float(2^23 + byte).
// Reports LDS-pipe cycles per wave-gather at CU level (all 12 waves hammering) next to the ~65 cycles per wave-slice that k_fill spends today.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

#define PITCH 130
#define TABLE (6 * PITCH * PITCH)

template <int MODE>
__global__ void __launch_bounds__(768) k_lds(int iters, long long* out, unsigned* check)
{
    extern __shared__ unsigned char lds[];
    for (int i = threadIdx.x; i < TABLE; i += 768) lds[i] = (unsigned char)(i * 7 + 3);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned q0 = 0x4B000000u, q1 = 0x4B000000u, q2 = 0x4B000000u, q3 = 0x4B000000u;
    unsigned acc = 0;
    unsigned state = threadIdx.x * 2654435761u + 12345u;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        unsigned addr;
        if (MODE == 0) {          // wave tile 8x8, ~5 texels apart, patch origin moving with the iteration
            const unsigned ox = (it * 3 + wave * 11) % 80, oy = (it * 5 + wave * 7) % 80, face = (it + wave) % 6;
            addr = face * PITCH * PITCH + (oy + (lane >> 3) * 5) * PITCH + ox + (lane & 7) * 5;
        } else if (MODE == 1) {   // random
            state = state * 1664525u + 1013904223u;
            addr = (state >> 8) % (TABLE - PITCH - 2);
        } else {
            addr = (it * 131) % (TABLE - PITCH - 2);
        }
        asm volatile("ds_read_u8_d16 %0, %4\n\tds_read_u8_d16 %1, %4 offset:1\n\tds_read_u8_d16 %2, %4 offset:130\n\tds_read_u8_d16 %3, %4 offset:131\n\ts_waitcnt lgkmcnt(0)"
                     : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(addr) : "memory");
        acc += q0 ^ q1 ^ q2 ^ q3;
    }
    const long long t1 = clock64();
    if (lane == 0) out[blockIdx.x * 16 + wave] = t1 - t0;
    if (threadIdx.x == 5) { check[0] = q0; check[1] = acc; }
}

template <int MODE>
void run(const char* name, long long* d_out, unsigned* d_chk)
{
    const int iters = 4000, nblk = 64;
    hipLaunchKernelGGL((k_lds<MODE>), dim3(nblk), dim3(768), TABLE + 64, 0, iters, d_out, d_chk);
    std::vector<long long> h(nblk * 16);
    unsigned chk[2];
    (void)hipMemcpy(h.data(), d_out, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    (void)hipMemcpy(chk, d_chk, sizeof chk, hipMemcpyDeviceToHost);
    double mx = 0;
    for (int b = 0; b < nblk; ++b) for (int w = 0; w < 12; ++w) mx = mx > (double)h[b * 16 + w] ? mx : (double)h[b * 16 + w];
    float f; unsigned u = chk[0]; memcpy(&f, &u, 4);
    printf("%-28s: %7.2f cycles per wave-gather (4 ds_read_u8_d16) per CU with 12 waves resident; sample reg %08x = %.1f\n", name,
           mx / (12.0 * iters), chk[0], f);
}

int main()
{
    long long* d_out; unsigned* d_chk;
    (void)hipMalloc(&d_out, 64 * 16 * sizeof(long long)); (void)hipMalloc(&d_chk, 16);
    run<0>("8x8 tile, 5 texels apart", d_out, d_chk);
    run<1>("random addresses", d_out, d_chk);
    run<2>("one address (broadcast)", d_out, d_chk);
    return 0;
}
