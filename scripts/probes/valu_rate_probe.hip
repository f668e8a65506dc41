// valu_rate_probe.hip -- how many cycles does one wave64 VALU instruction cost a gfx950 SIMD, by opcode, with 1..4 waves per SIMD
// and 1 or 4 independent chains per wave?  (Design input for k_fill: is the unpacked-f32 issue rate 2 or 4 cycles per instruction,
// what do the cube-map / transcendental / conversion instructions cost, what do packed-f32 and LDS atomics cost.)
// Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_probe scripts/probes/valu_rate_probe.hip ; run: /tmp/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

enum { OP_FMA, OP_PKFMA, OP_MUL, OP_CUBEID, OP_CUBEMA, OP_RCP, OP_SQRT, OP_FLOOR, OP_FRACT, OP_CVTU, OP_MED3, OP_MAX, OP_CNDMASK,
       OP_MIX, OP_DSADD, OP_DSMAX, OP_DSREAD, OP_MOVREL, OP_ALIGNBYTE, OP_CVTUB, OP_ADD, OP_SUB, OP_FMAC, OP_CMP, OP_MOV, OP_PKADD, OP_PKMUL, OP_ADDU, OP_MIN, OP_DSREADU8, OP_DSRMW, OP_CVTF16, OP_MULDEN, OP_MULNRM, OP_SUBDEN, OP_PKMULDEN, OP_FMADEN, OP_MAXI, OP_AND, OP_LSHL, OP_MULU24, OP_CVTF16F32, OP_CVTI, OP_CNDS, OP_MADU24, OP_LSHLADD, OP_COUNT};
static const char* NAMES[] = {"v_fma_f32", "v_pk_fma_f32", "v_mul_f32", "v_cubeid_f32", "v_cubema_f32", "v_rcp_f32", "v_sqrt_f32", "v_floor_f32",
                              "v_fract_f32", "v_cvt_u32_f32", "v_med3_f32", "v_max_f32", "v_cndmask_b32", "v_fma_mix_f32", "ds_add_f32", "ds_max_f32",
                              "ds_read_b32", "v_mov(gpr_idx)", "v_alignbyte_b32", "v_cvt_f32_ubyte1", "v_add_f32", "v_sub_f32", "v_fmac_f32", "v_cmp_le_f32", "v_mov_b32",
                              "v_pk_add_f32", "v_pk_mul_f32", "v_add_u32", "v_min_f32", "ds_read_u8", "ds_read+add+ds_write", "v_cvt_f32_f16",
                              "v_mul_f32 denormal*K", "v_mul_f32 normal*K (same form)", "v_sub_f32 den-den", "v_pk_mul_f32 den*K", "v_fma_f32 K*den+den",
                              "v_max_i32", "v_and_b32", "v_lshlrev_b32", "v_mul_u32_u24", "v_cvt_f16_f32", "v_cvt_i32_f32", "v_cndmask_b32 (sgpr mask)", "v_mad_u32_u24", "v_lshl_add_u32"};

template <int OP>
__device__ __forceinline__ void op(float& a, f2& a2, float b, float c, unsigned lds_addr)
{
    if (OP == OP_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
    if (OP == OP_PKFMA) { f2 b2 = {b, b}, c2 = {c, c}; asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a2) : "v"(b2), "v"(c2)); }
    if (OP == OP_MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a) : "v"(b));
    if (OP == OP_CUBEID) asm volatile("v_cubeid_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
    if (OP == OP_CUBEMA) asm volatile("v_cubema_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
    if (OP == OP_RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(a));
    if (OP == OP_SQRT) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a));
    if (OP == OP_FLOOR) asm volatile("v_floor_f32 %0, %0" : "+v"(a));
    if (OP == OP_FRACT) asm volatile("v_fract_f32 %0, %0" : "+v"(a));
    if (OP == OP_CVTU) asm volatile("v_cvt_u32_f32 %0, %0" : "+v"(a));
    if (OP == OP_MED3) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
    if (OP == OP_MAX) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a) : "v"(b));
    if (OP == OP_CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b) : );
    if (OP == OP_ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(b));
    if (OP == OP_SUB) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a) : "v"(b));
    if (OP == OP_FMAC) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
    if (OP == OP_CMP) asm volatile("v_cmp_le_f32 vcc, %0, %1" : : "v"(a), "v"(b) : "vcc");
    if (OP == OP_MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(a) : "v"(b));
    if (OP == OP_PKADD) { f2 b2 = {b, b}; asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a2) : "v"(b2)); }
    if (OP == OP_PKMUL) { f2 b2 = {b, b}; asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a2) : "v"(b2)); }
    if (OP == OP_ADDU) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));
    if (OP == OP_MIN) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a) : "v"(b));
    if (OP == OP_DSREADU8) asm volatile("ds_read_u8 %0, %1" : "=v"(a) : "v"(lds_addr) : "memory");
    if (OP == OP_DSRMW) { float t; asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)\n v_add_f32 %0, %0, %2\n ds_write_b32 %1, %0" : "=&v"(t) : "v"(lds_addr), "v"(b) : "memory"); }
    // denormal operands (round 3: the fill treats ds_read_u8 results as denormal floats): non-chained forms, inputs stay denormal
    if (OP == OP_MULDEN) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a) : "v"(__uint_as_float(lds_addr & 255u | 1u)), "v"(8.507059173023462e37f));
    if (OP == OP_MULNRM) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a) : "v"(b), "v"(8.507059173023462e37f));
    if (OP == OP_SUBDEN) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(a) : "v"(__uint_as_float(lds_addr & 255u | 1u)), "v"(__uint_as_float(77u)));
    if (OP == OP_PKMULDEN) { f2 d2 = {__uint_as_float(lds_addr & 255u | 1u), __uint_as_float(99u)}, k2 = {8.507059173023462e37f, 8.507059173023462e37f}; asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(a2) : "v"(d2), "v"(k2)); }
    if (OP == OP_FMADEN) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(a) : "v"(__uint_as_float(lds_addr & 255u | 1u)), "v"(8.507059173023462e37f), "v"(__uint_as_float(55u)));
    if (OP == OP_CVTF16) asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(a));
    if (OP == OP_MIX) asm volatile("v_fma_mix_f32 %0, %0, %1, %2 op_sel_hi:[0,1,0]" : "+v"(a) : "v"(b), "v"(c));
    if (OP == OP_DSADD) asm volatile("ds_add_f32 %0, %1" : : "v"(lds_addr), "v"(b) : "memory");
    if (OP == OP_DSMAX) asm volatile("ds_max_f32 %0, %1" : : "v"(lds_addr), "v"(b) : "memory");
    if (OP == OP_DSREAD) asm volatile("ds_read_b32 %0, %1" : "=v"(a) : "v"(lds_addr) : "memory");
    if (OP == OP_MOVREL) asm volatile("v_mov_b32 %0, %1" : "=v"(a) : "v"(b));
    if (OP == OP_ALIGNBYTE) asm volatile("v_alignbyte_b32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
    if (OP == OP_CVTUB) asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(a));
    // round 4: integer / conversion flavours (which SQ_INSTS_VALU_* class counter each opcode lands in, and its rate; scripts/valu_class_calibration.py)
    if (OP == OP_MAXI) asm volatile("v_max_i32 %0, %0, %1" : "+v"(a) : "v"(b));
    if (OP == OP_AND) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a) : "v"(b));
    if (OP == OP_LSHL) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a));
    if (OP == OP_MULU24) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a) : "v"(b));
    if (OP == OP_CVTF16F32) asm volatile("v_cvt_f16_f32 %0, %0" : "+v"(a));
    if (OP == OP_CVTI) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a));
    if (OP == OP_CNDS) asm volatile("v_cndmask_b32 %0, %0, %1, s[20:21]" : "+v"(a) : "v"(b) : "s20", "s21");
    if (OP == OP_MADU24) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
    if (OP == OP_LSHLADD) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a) : "v"(b));
}

template <int OP, int ILP>
__global__ void __launch_bounds__(1024) k_probe(int iters, long long* out, float seed, unsigned long long mask)
{
    extern __shared__ float lds[];
    float x[ILP]; f2 x2[ILP];
    for (int i = 0; i < ILP; ++i) { x[i] = seed + threadIdx.x + i; x2[i] = f2{x[i], x[i] + 1.f}; }
    const float b = seed * 0.5f + 1.0f, c = seed + 0.25f;
    const unsigned addr = threadIdx.x * 4;                  // conflict-free: consecutive dwords
    lds[threadIdx.x] = 0.f;
    __syncthreads();
    asm volatile("s_mov_b64 vcc, exec" ::: "vcc");
    unsigned long long saved;
    asm volatile("s_mov_b64 %0, exec\n s_mov_b64 exec, %1" : "=&s"(saved) : "s"(mask));
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < ILP; ++i) op<OP>(x[i], x2[i], b, c, addr);
    }
    if (OP == OP_DSADD || OP == OP_DSMAX || OP == OP_DSREAD || OP == OP_DSREADU8 || OP == OP_DSRMW) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const long long t1 = clock64();
    asm volatile("s_mov_b64 exec, %0" : : "s"(saved));
    float s = 0.f;
    for (int i = 0; i < ILP; ++i) s += x[i] + x2[i][0] + x2[i][1];
    if (s == 12345.678f) out[1000] = 1;                     // keep the chains alive
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int OP, int ILP>
void run(int wps /* waves per SIMD */, long long* d_out, std::vector<long long>& h, unsigned long long mask = ~0ull)
{
    const int iters = 2000, nblk = 64;
    hipLaunchKernelGGL((k_probe<OP, ILP>), dim3(nblk), dim3(256 * wps), 4096, 0, iters, d_out, 1.0f, mask);
    hipMemcpy(h.data(), d_out, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    double mx = 0;
    for (int b = 0; b < nblk; ++b) for (int w = 0; w < 4 * wps; ++w) mx = mx > (double)h[b * 16 + w] ? mx : (double)h[b * 16 + w];
    const double instr_per_simd = (double)wps * iters * 16 * ILP;
    printf("  %-22s ILP %d  waves/SIMD %d  exec %016llx : %6.2f cycles per wave-instruction per SIMD\n", NAMES[OP], ILP, wps, mask, mx / instr_per_simd);
}

template <int OP>
void sweep(long long* d_out, std::vector<long long>& h)
{
    run<OP, 1>(1, d_out, h); run<OP, 4>(2, d_out, h); run<OP, 4>(4, d_out, h);
}
template <int OP>
void masks(long long* d_out, std::vector<long long>& h)
{
    run<OP, 4>(4, d_out, h, 0x00000000ffffffffull);      // low half only: does the SIMD-32 skip an all-inactive pass?
    run<OP, 4>(4, d_out, h, 0xffffffff00000000ull);
    run<OP, 4>(4, d_out, h, 0x5555555555555555ull);      // every other lane: nothing to skip
    run<OP, 4>(4, d_out, h, 0x000000000000ffffull);      // a quarter
    run<OP, 4>(4, d_out, h, 0x0000000000000001ull);
}

int main()
{
    long long* d_out; hipMalloc(&d_out, 4096 * sizeof(long long));
    std::vector<long long> h(2048);
    sweep<OP_FMA>(d_out, h); sweep<OP_PKFMA>(d_out, h); sweep<OP_MUL>(d_out, h); sweep<OP_ADD>(d_out, h); sweep<OP_SUB>(d_out, h); sweep<OP_FMAC>(d_out, h);
    sweep<OP_PKADD>(d_out, h); sweep<OP_PKMUL>(d_out, h); sweep<OP_CMP>(d_out, h); sweep<OP_MOV>(d_out, h); sweep<OP_ADDU>(d_out, h); sweep<OP_MIN>(d_out, h);
    sweep<OP_CUBEID>(d_out, h); sweep<OP_CUBEMA>(d_out, h);
    sweep<OP_RCP>(d_out, h); sweep<OP_SQRT>(d_out, h); sweep<OP_FLOOR>(d_out, h); sweep<OP_FRACT>(d_out, h); sweep<OP_CVTU>(d_out, h);
    sweep<OP_MED3>(d_out, h); sweep<OP_MAX>(d_out, h); sweep<OP_CNDMASK>(d_out, h); sweep<OP_MIX>(d_out, h); sweep<OP_CVTF16>(d_out, h);
    sweep<OP_ALIGNBYTE>(d_out, h); sweep<OP_CVTUB>(d_out, h);
    sweep<OP_MULDEN>(d_out, h); sweep<OP_MULNRM>(d_out, h); sweep<OP_SUBDEN>(d_out, h); sweep<OP_PKMULDEN>(d_out, h); sweep<OP_FMADEN>(d_out, h);
    sweep<OP_DSMAX>(d_out, h); sweep<OP_DSREAD>(d_out, h); sweep<OP_DSREADU8>(d_out, h); sweep<OP_DSRMW>(d_out, h);
    sweep<OP_MAXI>(d_out, h); sweep<OP_AND>(d_out, h); sweep<OP_LSHL>(d_out, h); sweep<OP_MULU24>(d_out, h); sweep<OP_MADU24>(d_out, h); sweep<OP_LSHLADD>(d_out, h);
    sweep<OP_CVTF16F32>(d_out, h); sweep<OP_CVTI>(d_out, h); sweep<OP_CNDS>(d_out, h);
    masks<OP_FMA>(d_out, h); masks<OP_FLOOR>(d_out, h); masks<OP_RCP>(d_out, h); masks<OP_PKFMA>(d_out, h); masks<OP_DSREAD>(d_out, h);
    return 0;
}
