// fill_kernels.h -- the FillVolume pass (Fill.shader:152-274 driven by VPR.cs:495-609) as ONE persistent launch: the kernel templates and
// their launchers.  Two translation units instantiate them: fill.hip (numVoxelsInMetavoxel = 16 / 32 / 64 as compile-time constants -- the
// benchmark configurations) and fill_generic.hip (any other voxel count the reference's inspector accepts, VPR.cs:84, with the count a run-time
// value: GEN below).
//
// Reference shape: one Graphics.Blit per occupied metavoxel (~11k draws at the 32^3 x 32^3 config), each
// fragment owning one voxel column: nv slices x P_mv full 4x4 mat-vec coverage tests, then a serial
// front-to-back propagate + RGBA16F store; draws serialised in z through a UAV (lightPropogationTex).
//
// CDNA4 shape (this file):
//   * unit of work = one wave on an 8x8-column tile of ONE occupied metavoxel (the most compact footprint against a particle's
//     sphere, i.e. the highest lane utilisation in covered slices; a slice store of a wave is eight 64-byte half lines), claimed
//     from a counter in z-major order; the light a voxel column has transmitted so far is handed from the unit of one occupied
//     metavoxel to the unit of the column's next one through a tagged 64-bit word (FillChain below), so the z-order dependency
//     costs one load and one store per column and metavoxel and the light map is written once;
//   * per (wave, particle): the column is a line ps(s) = A + s*B in particle space, so coverage is a quadratic
//     in the slice index; each lane solves it, a DPP OR-reduction merges the per-lane slice masks, and only the
//     wave-uniform slice range is tested (exact test unchanged: |ps|^2 <= 0.25, Fill.shader:172);
//   * particle records are wave-uniform -> scalar loads (s_load_dwordx16), matrix elements live in SGPRs;
//   * per-slice accumulators (density sum, ao max) are register arrays indexed by the uniform slice
//     (s_set_gpr_idx), no LDS, no scratch;
//   * the displacement cubemap is pre-expanded to bilinear footprints (one 16-byte load per covered voxel instead of four texel
//     fetches: gfx950 has no image/sampler hardware), or -- 8-bit maps, the reference asset's format -- lives in LDS as bytes.
// Bound: HBM store of the bricks (8 B/voxel) is the roofline; the kernels are limited well before it by VALU issue (k_fill_lds) and by
// the CU's L1/TA rate of the per-voxel footprint gather (k_fill).
#pragma once
#include <cstdlib>
#include <type_traits>

#include "vpfx_internal.h"

#ifndef VPFX_FILL_PIPE_LDS
#define VPFX_FILL_PIPE_LDS 2  // slices in flight on the LDS-resident cube map (4 ds_read_u8 each; lgkmcnt holds 15)
#endif
// Row pitch of the LDS byte table.  The four byte reads of a wave-gather sit ~5 texels apart across the 8x8 lanes; with the natural pitch
// S + 2 = 130 the 4 rows of a 32-lane group start 2.5 banks apart and collide 3-4-way (SQ_LDS_BANK_CONFLICT = 3/4 of the LDS cycles).
#ifndef VPFX_LDS_PITCH_128
#define VPFX_LDS_PITCH_128 136      // measured 130 / 132 / 136 / 140 / 144: 3.50 / 3.48 / 3.45 / 3.52 / 3.49 ms at C3
#endif
#define VPFX_STR2(x) #x
#define VPFX_STR(x) VPFX_STR2(x)
#ifndef VPFX_PROBE
#define VPFX_PROBE 0              // what-if / phase-timer builds (scripts/fill_phase_profile.py, profiles/r03_fill_whatif_C3_r8.txt): A/B builds only
#endif
#if VPFX_PROBE && !VPFX_AB
#error "VPFX_PROBE builds produce wrong bricks on purpose: make EXTRA='-DVPFX_AB=1 -DVPFX_PROBE=n'"
#endif
#if VPFX_PROBE == 9
// in-kernel phase timer (profiling builds only, scripts/fill_phase_profile.py): wave-cycles by phase, summed over all waves
#ifdef VPFX_FILL_MAIN_TU
__device__ unsigned long long g_fill_prof[12];    // [0..6] wave-cycles by phase, [7] wave lifetime, [8] units with a producer, [9] of which had to poll again, [10] polls
extern "C" __attribute__((visibility("default"))) int vpfx_probe_read(unsigned long long* out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fill_prof), sizeof(g_fill_prof)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[12] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_fill_prof), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#else
static __device__ unsigned long long g_fill_prof[12];    // (the run-time-nv kernels are not what the phase timer is read for)
#endif
#define VPFX_TICK(ph) do { const unsigned long long t_now_ = __builtin_amdgcn_s_memtime(); prof_acc[ph] += t_now_ - prof_last; prof_last = t_now_; } while (0)
#else
#define VPFX_TICK(ph) do { } while (0)
#endif
#define VPFX_LDS_READS 4          // LDS instructions per slice (lgkmcnt bookkeeping).  (Two unaligned ds_read_u16 instead: 8.6 ms)
#ifndef VPFX_FILL_LDS_WAVES
#define VPFX_FILL_LDS_WAVES 16  // waves of the persistent workgroup (one per CU): 16 = 4 per SIMD (<= 128 VGPRs; measured 8 / 12 / 16 waves: 4.00 / 3.61 / 3.56 ms at C3)
#endif
// ... and of the run-time-voxel-count instantiations (GEN): at 16 waves (128-VGPR cap) they spill 3-13 VGPRs (16-56 B of scratch per lane);
// 12 waves (168 VGPRs) hold everything in registers.  Measured at C3nv24 (32^3 x 24^3, round 6): see PERFLOG.
#ifndef VPFX_FILL_LDS_WAVES_GEN
#define VPFX_FILL_LDS_WAVES_GEN 16
#endif
// units per global-counter atomic of the persistent LDS kernel (workgroup-level claim, see k_fill_lds); 1 = one device-scope atomic per unit (A/B)
#ifndef VPFX_FILL_CLAIM
#define VPFX_FILL_CLAIM 16
#endif
#ifndef VPFX_FILL_CLAIM_PREF
#define VPFX_FILL_CLAIM_PREF 15      // the slot whose wave fetches the next block (measured at C3: slot 0 / 12 / 15 -> 2.99 / 2.965 / 2.95 ms)
#endif
#ifndef VPFX_FILL_MIN_CLAIM_LOG2
#define VPFX_FILL_MIN_CLAIM_LOG2 2   // small launches: blocks of at least 4 units (= working waves per workgroup, one per SIMD)
#endif
#define VPFX_FILL_CLAIM_RING 64      // blocks remembered per workgroup (power of two)
#ifndef VPFX_FILL_PIPE
#define VPFX_FILL_PIPE 4      // 2..6; measured at C3: 2 -> 5.33 ms, 3 -> 5.07 (4 waves/SIMD), 4 -> 4.80 (3 waves/SIMD)
#endif

namespace {

typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t pack_half2(float a, float b)
{
    half2_t h = {(_Float16)a, (_Float16)b};      // v_cvt_f16_f32: round-to-nearest-even
    return __builtin_bit_cast(uint32_t, h);
}

// OR-reduction across the 64 lanes of a wave with DPP row shifts/broadcasts; result is wave-uniform.
__device__ __forceinline__ uint32_t wave_or(uint32_t v)
{
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1,3
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2,3
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// s_waitcnt vmcnt(N) that also "touches" the destination registers of the load being waited for, so that the compiler
// orders every later use of Q after the wait (it cannot see the asynchronous write of the inline-asm load).
template <int N>
__device__ __forceinline__ void wait_vm(f32x4& q) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(q) : "n"(N) : "memory"); }
// drain phase of a group of depth D: slot i still has D-1-i younger loads behind it
template <int D>
__device__ __forceinline__ void wait_vm_dyn(int i, f32x4& q)
{
    if (D - 1 - i >= 7) wait_vm<7>(q); else if (D - 1 - i == 6) wait_vm<6>(q); else if (D - 1 - i == 5) wait_vm<5>(q);
    else if (D - 1 - i == 4) wait_vm<4>(q); else if (D - 1 - i == 3) wait_vm<3>(q); else if (D - 1 - i == 2) wait_vm<2>(q);
    else if (D - 1 - i == 1) wait_vm<1>(q); else wait_vm<0>(q);
}

// The per-slice accumulators (density sum, ao max) of a chunk are two register arrays indexed by the wave-uniform slice.  hipcc lowers
// `dens[s] += den` to read-modify-write through v_mov with s_set_gpr_idx (4 v_mov + 8 SALU per covered slice, a sixth of the covered-
// slice loop's instructions); with the arrays pinned to fixed registers the update is the two arithmetic instructions themselves, issued
// inside ONE indexing window with source-0 and destination both relative:  v_add_f32 v[0+s], v[0+s], den ; v_max_i32 v[32+s], v[32+s], net
// (ao and net are >= 0, so the integer max of the bit patterns is the float max; the exec mask restricts both to the covered lanes).
// (The north_star's "scattered into LDS-resident voxel tiles" was built and measured in round 2 -- the wave's (density, ao) tile in LDS with
// read-add-write 4.84 ms, with ds_add_f32 9.14 ms, against 4.66 ms for these register arrays, DESIGN.md 10 -- and removed from the source in
// round 4; the LDS holds the cube map instead.)
typedef float f32x32 __attribute__((ext_vector_type(32)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int CH> struct AccArr;
template <> struct AccArr<32> {
    typedef f32x32 type;
    // "clear it" (Fill.shader:178-181) in place: `dens = 0.f` makes hipcc keep a 32-register zero tuple alive just to copy from
    static __device__ __forceinline__ void clear(f32x32& dens, f32x32& ao)
    {
        asm volatile("v_mov_b32 v0, 0\n\tv_mov_b32 v1, 0\n\tv_mov_b32 v2, 0\n\tv_mov_b32 v3, 0\n\tv_mov_b32 v4, 0\n\tv_mov_b32 v5, 0\n\tv_mov_b32 v6, 0\n\tv_mov_b32 v7, 0\n\tv_mov_b32 v8, 0\n\tv_mov_b32 v9, 0\n\tv_mov_b32 v10, 0\n\tv_mov_b32 v11, 0\n\tv_mov_b32 v12, 0\n\tv_mov_b32 v13, 0\n\tv_mov_b32 v14, 0\n\tv_mov_b32 v15, 0\n\tv_mov_b32 v16, 0\n\tv_mov_b32 v17, 0\n\tv_mov_b32 v18, 0\n\tv_mov_b32 v19, 0\n\tv_mov_b32 v20, 0\n\tv_mov_b32 v21, 0\n\tv_mov_b32 v22, 0\n\tv_mov_b32 v23, 0\n\tv_mov_b32 v24, 0\n\tv_mov_b32 v25, 0\n\tv_mov_b32 v26, 0\n\tv_mov_b32 v27, 0\n\tv_mov_b32 v28, 0\n\tv_mov_b32 v29, 0\n\tv_mov_b32 v30, 0\n\tv_mov_b32 v31, 0\n\tv_mov_b32 v32, 0\n\tv_mov_b32 v33, 0\n\tv_mov_b32 v34, 0\n\tv_mov_b32 v35, 0\n\tv_mov_b32 v36, 0\n\tv_mov_b32 v37, 0\n\tv_mov_b32 v38, 0\n\tv_mov_b32 v39, 0\n\tv_mov_b32 v40, 0\n\tv_mov_b32 v41, 0\n\tv_mov_b32 v42, 0\n\tv_mov_b32 v43, 0\n\tv_mov_b32 v44, 0\n\tv_mov_b32 v45, 0\n\tv_mov_b32 v46, 0\n\tv_mov_b32 v47, 0\n\tv_mov_b32 v48, 0\n\tv_mov_b32 v49, 0\n\tv_mov_b32 v50, 0\n\tv_mov_b32 v51, 0\n\tv_mov_b32 v52, 0\n\tv_mov_b32 v53, 0\n\tv_mov_b32 v54, 0\n\tv_mov_b32 v55, 0\n\tv_mov_b32 v56, 0\n\tv_mov_b32 v57, 0\n\tv_mov_b32 v58, 0\n\tv_mov_b32 v59, 0\n\tv_mov_b32 v60, 0\n\tv_mov_b32 v61, 0\n\tv_mov_b32 v62, 0\n\tv_mov_b32 v63, 0\n\t" : "={v[0:31]}"(dens), "={v[32:63]}"(ao));
    }
    static __device__ __forceinline__ void add_max(f32x32& dens, f32x32& ao, int s, float den, float net)
    {
        // s_nop 1 between s_set_gpr_idx_on and the first indexed VALU is REQUIRED on gfx950: without it the kernel faults at the 32^3
        // benchmark grid (found by bisecting nop positions, scripts/gpu_variants.sh; hipcc's own windows only ever hold one v_mov and
        // never showed it).  With it the bricks are bit-identical to the v_mov lowering (scripts/fill_hash.py, default and EXACT math).
        asm volatile("s_set_gpr_idx_on %4, gpr_idx(SRC0,DST)\n\ts_nop 1\n\tv_add_f32 v0, v0, %2\n\tv_max_i32 v32, v32, %3\n\ts_set_gpr_idx_off"
                     : "+{v[0:31]}"(dens), "+{v[32:63]}"(ao) : "v"(den), "v"(net), "s"(s));     // rewrites M0 like every gpr-idx window hipcc emits itself
    }
};
template <> struct AccArr<16> {
    typedef f32x16 type;
    static __device__ __forceinline__ void clear(f32x16& dens, f32x16& ao)
    {
        asm volatile("v_mov_b32 v0, 0\n\tv_mov_b32 v1, 0\n\tv_mov_b32 v2, 0\n\tv_mov_b32 v3, 0\n\tv_mov_b32 v4, 0\n\tv_mov_b32 v5, 0\n\tv_mov_b32 v6, 0\n\tv_mov_b32 v7, 0\n\tv_mov_b32 v8, 0\n\tv_mov_b32 v9, 0\n\tv_mov_b32 v10, 0\n\tv_mov_b32 v11, 0\n\tv_mov_b32 v12, 0\n\tv_mov_b32 v13, 0\n\tv_mov_b32 v14, 0\n\tv_mov_b32 v15, 0\n\tv_mov_b32 v16, 0\n\tv_mov_b32 v17, 0\n\tv_mov_b32 v18, 0\n\tv_mov_b32 v19, 0\n\tv_mov_b32 v20, 0\n\tv_mov_b32 v21, 0\n\tv_mov_b32 v22, 0\n\tv_mov_b32 v23, 0\n\tv_mov_b32 v24, 0\n\tv_mov_b32 v25, 0\n\tv_mov_b32 v26, 0\n\tv_mov_b32 v27, 0\n\tv_mov_b32 v28, 0\n\tv_mov_b32 v29, 0\n\tv_mov_b32 v30, 0\n\tv_mov_b32 v31, 0\n\t" : "={v[0:15]}"(dens), "={v[16:31]}"(ao));
    }
    static __device__ __forceinline__ void add_max(f32x16& dens, f32x16& ao, int s, float den, float net)
    {
        asm volatile("s_set_gpr_idx_on %4, gpr_idx(SRC0,DST)\n\ts_nop 1\n\tv_add_f32 v0, v0, %2\n\tv_max_i32 v16, v16, %3\n\ts_set_gpr_idx_off"
                     : "+{v[0:15]}"(dens), "+{v[16:31]}"(ao) : "v"(den), "v"(net), "s"(s));     // rewrites M0 like every gpr-idx window hipcc emits itself
    }
};

// LDS-resident cube map (TAB != 0): the bilinear quad is four ds_read_u8 (zero-extended bytes) off one address.  LDS operations
// return in order among themselves, so "lgkmcnt <= 4 * (slices issued behind)" means the oldest slice's four bytes have landed
// (other lgkm traffic -- the compiler's scalar loads -- only adds to the counter, i.e. makes the wait stricter, never laxer).
struct QuadU8 { unsigned a, b, c, d; };          // texels (x0,y0), (x0,y0+1), (x0+1,y0), (x0+1,y0+1)
template <int N>
__device__ __forceinline__ void wait_lgkm(QuadU8& q)
{
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(q.a), "+v"(q.b), "+v"(q.c), "+v"(q.d) : "n"(N) : "memory");
}
template <int D>
__device__ __forceinline__ void wait_lgkm_dyn(int i, QuadU8& q)
{
    constexpr int R = VPFX_LDS_READS;
    if (D - 1 - i >= 3) wait_lgkm<3 * R>(q); else if (D - 1 - i == 2) wait_lgkm<2 * R>(q); else if (D - 1 - i == 1) wait_lgkm<R>(q); else wait_lgkm<0>(q);
}

template <bool EXACT>
__device__ __forceinline__ float fdiv(float a, float b)
{
    return EXACT ? a / b : a * __builtin_amdgcn_rcpf(b);
}

// kernel pointer arguments are passed one by one as __restrict__ so that wave-uniform reads (particle records,
// CSR offsets, MV positions) are provably unclobbered by the brick stores and become scalar (SMEM) loads.
#define FILL_PTR_PARAMS                                                                                      \
    const float* __restrict__ p_mvPos, const int* __restrict__ p_offsets, const int* __restrict__ p_ids,     \
    const float* __restrict__ p_rec, const int* __restrict__ p_brick_index, const int* __restrict__ p_colorder, \
    const float4* __restrict__ p_cubequads, const float* __restrict__ p_depthmap /* nullable */,             \
    const float* __restrict__ p_light_in /* nullable => 1.0 */, float* __restrict__ p_light_out,             \
    uint2* __restrict__ p_bricks, float2* __restrict__ p_dens_ao /* split-fill scratch */,                         \
    const float4* __restrict__ p_ws /* particle world position + diameter */
#define FILL_PTR_ARGS(P) (P).mvPos, (P).offsets, (P).ids, (P).rec, (P).brick_index, (P).colorder, (P).cubequads, \
                         (P).depthmap, (P).light_in, (P).light_out, (P).bricks, (P).dens_ao, (P).ws

struct FillPtrs {
    const float* mvPos; const int* offsets; const int* ids; const float* rec; const int* brick_index;
    const int* colorder; const float4* cubequads; const float* depthmap; const float* light_in;
    float* light_out; uint2* bricks; float2* dens_ao; const float4* ws;
};

// compute_voxel_color (Fill.shader:110-135) for a covered voxel, split in two pipeline stages so that the one
// memory access (the cubemap footprint) can be in flight while the next slice is being addressed:
//   stage 1: texCUBE addressing  -> footprint index + bilinear weights
//   stage 2: bilinear + displacement + smoothstep -> (density contribution, net displacement)
// DONE (displacement scale exactly 1, default math): net displacement == the filtered texel, and the reference's smoothstep(net, 0.7 net, x)
// jumps at net == 0 (x / +0 -> 1, but 0 for any net > 0).  Whether a bilinear weight is EXACTLY 0 then decides between no density and full
// density, so the in-face coordinates must not come out on the other side of an integer than the oracle's.  The reciprocal-based
// coordinates are off by <= ~3e-5; whenever some lane's coordinate lies within 1e-4 of an integer (2.6 % of the wave-slices) the wave
// recomputes them with the oracle's IEEE arithmetic, and the weights are fx - floor(fx) like the oracle's (zero iff fx is an integer).
// Everything else stays on the fast path: <= 1 fp16 ulp like every default-math fill, but no voxel flips sides of the jump.
template <bool EXACT, int TAB, bool DONE>
__device__ __forceinline__ unsigned cube_address(const FillConsts& f, float psx, float psy, float psz, float& tx, float& ty, float lds_bias)
{
    // D3D cube-face selection with the CDNA cube-map VALU instructions (v_cubeid/sc/tc/ma_f32): face id, the two in-face
    // coordinates and 2x the signed major component in four instructions instead of ~25 compares and selects.  Ties
    // between |x|, |y|, |z| resolve z before y before x -- the arithmetic spec adopts exactly that rule.
    const float fid = __builtin_amdgcn_cubeid(psx, psy, psz);          // 0..5 = +X,-X,+Y,-Y,+Z,-Z
    const float sc = __builtin_amdgcn_cubesc(psx, psy, psz);
    const float tc = __builtin_amdgcn_cubetc(psx, psy, psz);
    const float ma2 = fabsf(__builtin_amdgcn_cubema(psx, psy, psz));   // 2 |major|
    const float Sf = f.half_s + f.half_s;
    float fx, fy;
    if (EXACT) {
        // u/2 = sc / (2|major|): scaling by two is exact, so fma(u/2, S, S/2 - 0.5) == fma(u, S/2, S/2 - 0.5) bit for bit
        float uh = 0.f, vh = 0.f;
        if (ma2 > 0.f) { const float inv = 1.0f / ma2; uh = sc * inv; vh = tc * inv; }
        fx = fmaf(uh, Sf, f.half_s_m05); fy = fmaf(vh, Sf, f.half_s_m05);
    } else {
        // fast path: S / (2|major|) once (for the zero vector the clamp turns 1/0 into a finite number, sc = tc = 0 then
        // give the face centre like the EXACT branch), one FMA per axis
        const float invS = Sf * __builtin_amdgcn_rcpf(ma2 + 1.0e-30f);     // + 1e-30: bit-neutral for any real direction, keeps 1/0 finite
        fx = fmaf(sc, invS, f.half_s_m05); fy = fmaf(tc, invS, f.half_s_m05);
#ifndef VPFX_DONE_ALWAYS_EXACT
#define VPFX_DONE_ALWAYS_EXACT 0
#endif
#ifndef VPFX_DONE_SKIP_EXACT
#define VPFX_DONE_SKIP_EXACT 0
#endif
        if (DONE && !VPFX_DONE_SKIP_EXACT) {
            const float wx = fx - floorf(fx), wy = fy - floorf(fy);
            const bool near_int = fminf(wx, wy) < 1.0e-4f || fmaxf(wx, wy) > 1.0f - 1.0e-4f;
            if (VPFX_DONE_ALWAYS_EXACT || __builtin_amdgcn_ballot_w64(near_int)) {   // wave-uniform, rare
                float uh = 0.f, vh = 0.f;
                if (ma2 > 0.f) { const float inv = 1.0f / ma2; uh = sc * inv; vh = tc * inv; }
                fx = fmaf(uh, Sf, f.half_s_m05); fy = fmaf(vh, Sf, f.half_s_m05);
            }
        }
    }
    const float x0 = floorf(fx), y0 = floorf(fy);
    // EXACT keeps the oracle's fx - floor(fx); the fast path uses v_fract_f32 (identical except that a weight that
    // would round up to exactly 1.0 is returned as the largest float below 1).  (Measured without gain, bit-identical bricks: the
    // subtract in the fast path too -- it costs two registers more and with them two spills --, the index as fx - fract(fx) instead of
    // v_floor, the slice index advanced by adds instead of converted.)
#ifndef VPFX_W_SUB
#define VPFX_W_SUB 1     // weights by subtraction (full-rate v_sub instead of v_fract) on the default path too: 3.08 -> 3.04 ms at C3, <= 1 fp16 ulp like before; 0 = A/B
#endif
    tx = (EXACT || DONE || VPFX_W_SUB) ? fx - x0 : __builtin_amdgcn_fractf(fx);
    ty = (EXACT || DONE || VPFX_W_SUB) ? fy - y0 : __builtin_amdgcn_fractf(fy);
    // |sc|, |tc| <= |major|, so fx, fy lie in [-0.5, S - 0.5] (up to the reciprocal's last ulp) and floor() in [-1, S - 1]:
    // the clamp-addressing of the footprint table never needs a min/max here.  The BYTE offset of the column pair,
    // 8 ((face (S+1) + y0 + 1)(S+2) + x0 + 1), is formed in float as three FMAs (exact: small integers) + one conversion.
    const float S2f = Sf + 2.0f;
    if (TAB == 0) return (unsigned)fmaf(fid, 8.0f * ((Sf + 1.0f) * S2f), fmaf(y0, 8.0f * S2f, fmaf(x0, 8.0f, 8.0f * (S2f + 1.0f))));
    // LDS table: bytes [face][S+2][pitch] with the clamp border replicated; byte address (face (S+2) + y0 + 1) pitch + x0 + 1 + base,
    // lds_bias = pitch + 1 + base (exact in float: < 2^24)
    const float pf = (float)f.lds_pitch;
    return (unsigned)fmaf(fid, S2f * pf, fmaf(y0, pf, x0 + lds_bias));
}

// BYTES (VPFX_BYTE_DENORM): q holds the four texel BYTES as loaded by ds_read_u8, reinterpreted as floats -- denormals b * 2^-149 (the
// kernels run with f32 denormals enabled, hipcc's default).  Instead of four quarter-rate v_cvt_f32_u32 the bilinear works on them directly:
// differences of denormals are exact, and the x weight and the two base texels are scaled by K = 2^126 (three full-rate v_mul_f32), so that
// every FMA sees normal-range products: raw comes out as 2^-23 x the filtered byte with exactly the roundings of the converted form (scaling
// by powers of two commutes with rounding; nothing under- or overflows: |K tx (b1 - b0) 2^-149| < 2^-15, smallest non-zero term >= 2^-23 ulp
// of a weight >= 2^-24), and 2^23 is folded into Dk.  Bit-identical bricks, 10 -> 9 instructions, none of them quarter-rate.
#ifndef VPFX_BYTE_DENORM
#define VPFX_BYTE_DENORM 1
#endif
template <bool EXACT, bool DONE, bool BYTES = false>
__device__ __forceinline__ void cube_shade(const FillConsts& f, float one_minus_D /* in a VGPR: an FMA reads one SGPR only */,
                                           const float4 q /* (t00, t01, t10, t11) */, float tx, float ty,
                                           float d2, float opw, float& den, float& net, float Dk /* D, or D/255 for byte texels */,
                                           float smooth_c1 /* 10/3 in a VGPR: a VOP3 FMA reads one SGPR / no literal */)
{
    float a, b;
    if (BYTES) {
        const float K = 8.507059173023462e37f;                                    // 2^126
        const float txK = tx * K;
        a = fmaf(txK, q.z - q.x, q.x * K); b = fmaf(txK, q.w - q.y, q.y * K);
    } else {
        a = fmaf(tx, q.z - q.x, q.x); b = fmaf(tx, q.w - q.y, q.y);
    }
    const float raw = fmaf(ty, b - a, a);
    net = fmaf(Dk, raw, one_minus_D);                                             // netDisplacement   :119
    float t;
    if (EXACT) {
        const float d2q = 4.0f * d2;                                              // dot(2ps, 2ps)     :121
        t = (d2q - net) / (0.7f * net - net);                                     // smoothstep(net, 0.7 net, d2q) :126
        t = fminf(fmaxf(t, 0.f), 1.f);
    } else {
        // same quantity, (4 d2 - net) / (-0.3 net) = 10/3 - (40/3) d2 / net, as reciprocal + multiply + fused clamp.  One input
        // needs the quotient form: displacement scale exactly 1 (DONE) makes net == 0 on a zero texel, where the reference's
        // (x - net) / (0.7 net - net) is x / +0 = +inf -> saturate 1 (density = opacityFactor; 0 / 0 saturates to 0) while
        // 10/3 - (40/3) d2 / 0 would give 0.  (4 d2 - net) * rcp(fma(net, 0.7, -net)) keeps the sign of that zero (+3 % of the kernel:
        // compiled into the D == 1 kernels only).
        // (the saturate rides on the producing instruction's clamp bit; hipcc otherwise spends a v_max_f32 ... clamp after a literal-form FMA)
        if (DONE) {
            const float num = fmaf(d2, 4.0f, -net), rden = __builtin_amdgcn_rcpf(fmaf(net, 0.7f, -net));
            // s_nop 0: rden comes straight out of v_rcp_f32, and on gfx950 a non-transcendental VALU that reads a transcendental's result
            // needs one wait state in between.  hipcc inserts it for its own instructions but does not look inside inline asm: without it
            // the multiply read a stale register whenever the scheduler placed it right behind the v_rcp (NV = 16 / 64 instantiations:
            // thousands of wrong voxels, found by the randomised sweep).
            asm("s_nop 0\n\tv_mul_f32_e64 %0, %1, %2 clamp" : "=v"(t) : "v"(num), "v"(rden));
        } else {
            const float q = d2 * __builtin_amdgcn_rcpf(net);
            asm("v_fma_f32 %0, %1, %2, %3 clamp" : "=v"(t) : "v"(q), "s"(-13.333333f), "v"(smooth_c1));
        }
    }
    // :126-127, :130-131: t*t*(3 - 2t) * opacityFactor * (fade ? opacity : 1).  opw = 1.0 exactly when _FadeOutParticles is off.
    // The fast path folds the wave-uniform factors into the cubic's coefficients (t*t) * (3k - 2k t), k = opacityFactor * opw.
    if (EXACT) {
        const float base = (t * t) * (3.0f - 2.0f * t);
        den = (base * f.opacity_factor) * opw;
    } else {
        const float k = f.opacity_factor * opw;
        den = (t * t) * fmaf(t, -2.0f * k, 3.0f * k);
    }
}

// One 8-byte brick entry.  NT: stored with the non-temporal hint -- the bricks stream out once per fill (2.8 GB at C3, 168 GB at C5, far beyond
// the 32 MB of L2) and are next read by the ray-march long after; as normal stores they push the kernel's small re-used state out of L2: the
// per-wave scratch of the spilled registers, the hand-off words, the particle records.  Measured in round 6 (EA request counters, PERFLOG):
//   nv = 64 (k_fill_lds<64> spills 18 VGPRs = 76 B per lane and unit: 25 GB of scratch written back and re-fetched per C5 fill):
//           traffic 244 -> 199 GB per launch (1.43x -> 1.16x of the algorithmic bytes), fill 166.5 -> 165.3 ms -- taken;
//   nv = 32: the builtin (and the same instruction as inline asm) perturbs register allocation of the 32-slice store loop (scratch 12 -> 376 /
//           156 B per lane) and the fill is 5-7 % SLOWER -- not taken: the nv <= 32 and the run-time-nv kernels keep the plain store.
template <bool NT>
__device__ __forceinline__ void store_brick(uint2* p, uint32_t lo, uint32_t hi)
{
    if constexpr (NT) __builtin_nontemporal_store((unsigned long long)lo | ((unsigned long long)hi << 32), reinterpret_cast<unsigned long long*>(p));
    else *p = make_uint2(lo, hi);
}

// Chained fill.  The unit of work of the persistent kernel is ONE metavoxel of one 8x8-column tile, not the tile's whole walk along the
// light axis: 16 x fewer, 32 x longer units (C3: 16 384 walks of up to 3.4 x the mean for 4 096 waves) left the waves idle for 18 % of the
// kernel (SQ_WAVE_CYCLES), per-metavoxel units (181 k) do not.  What a walk carried in a register -- the light transmitted so far, one
// float per voxel column -- is handed from the unit of one occupied metavoxel to the unit of the next one in that column through memory:
// one 64-bit word per column, tag << 32 | float bits, written and polled with relaxed agent-scope atomics (single-location coherence is
// all it needs: no fence, no L2 write-back).  Units are claimed in z-major order, so the producer of a word was always claimed earlier, by
// a wave that is running and never waits for a later unit: no deadlock; and a consumer only looks at the word after its own
// accumulation, ~70 us after claiming, when the producer -- claimed >= one metavoxel layer earlier -- has long finished.
struct FillChain {
    unsigned long long* words;    // [LH][LW]
    const int* ord;               // [n3] occupied MVs of the column in front of this one
    const int* colcount;          // [nxy]
    const int* occ_list;          // occupied MVs of the slab, z-major
    uint32_t tag_base;            // launch sequence number x (Nz + 1): tags of different launches never collide
    int* error;                   // host-mapped watchdog flag
    uint32_t spin_limit;          // polls before a unit gives up (VPFX_CHAIN_SPIN_LIMIT)
    uint32_t wait_bias;           // 0; the test hook adds an offset to the awaited tag so that it never arrives
};
#define VPFX_CHAIN_SPIN_LIMIT (1u << 22)       // x (s_sleep 8 = 512 cycles + a memory round trip): seconds; a real wait is microseconds
__device__ __forceinline__ float chain_wait(const unsigned long long* w, uint32_t tag, int* error, uint32_t spin_limit, unsigned* polls = nullptr)
{
    unsigned long long v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    for (; (uint32_t)(v >> 32) != tag; ++spins) {
        if (spins > spin_limit) { *error = 1; break; }        // never seen; reported at the caller's next sync instead of hanging the GPU
        __builtin_amdgcn_s_sleep(8);
        v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (polls) *polls = spins;                                // (profiling builds)
    return __uint_as_float((uint32_t)v);
}
__device__ __forceinline__ void chain_publish(unsigned long long* w, float v, uint32_t tag)
{
    __hip_atomic_store(w, ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// MODE 0: fused fill (bricks + light map).  MODE 1: slab-local pass: density/ao to scratch, slab transmittance
// (propagation with T_in = 1) to light_out.
#ifndef VPFX_FILL_WAVES
#define VPFX_FILL_WAVES 3      // min waves per SIMD: caps the kernels at 168 VGPRs (without the cap hipcc takes 165-174 and NV = 64 drops to 2 waves)
#endif
// One wave's share of the fill: an 8x8-column tile (this lane: column px, py) of metavoxel column (xx, yy), walked along the light
// axis zz = z0 .. z1.  TAB = 0: cube-map footprints from the global f32 pair table (p_cubequads);  TAB = 1 / 2: R8 cube map
// resident in LDS (1: S = 128, row pitch 130 as instruction immediates; 2: any S, pitch from FillConsts).
// CHAIN: only metavoxel zz_a of the column, light handed on through `ch` (see FillChain); otherwise zz_a .. zz_b - 1 with the light in a register.
// MATH: 0 = default (v_rcp_f32 in the covered-voxel math), 1 = EXACT (IEEE divisions: bricks bit-identical to the oracle), 2 = default math
// for displacement scale exactly 1 (DONE, see cube_address).
// GEN: the voxel count is the run-time g.nv (any numVoxelsInMetavoxel the reference's inspector accepts, VPR.cs:84; `_NumVoxels` is a FLOAT
// uniform, VPR.cs:527, so nv / 2 in get_voxel_world_pos is a float division for odd counts too -- fnv / 2.0f below) and NV is only the
// capacity class of the per-slice register arrays (32: nv <= 32, 64: nv <= 64).  The 8x8-column tiles that overhang the brick (nv not a
// multiple of 8) run with the overhanging lanes CLAMPED onto the brick's last column / row: they recompute that column bit for bit (same
// inputs, same instructions; the cross-lane steps are ORs and ballots, which duplicates do not change) and every store and the chain's
// publish are masked to the lanes that own a column.  Slices >= nv of the last register chunk are never covered, shaded or stored.
template <int NV, int MATH, int MODE, int TAB, bool CHAIN = false, bool GEN = false>
__device__ __forceinline__ void fill_tile(const GridConsts& g, const FillConsts& f, FILL_PTR_PARAMS, const int xx, const int yy, const int px_in,
                                          const int py_in, const int lane, const unsigned lds_base, const int zz_a, const int zz_b,
                                          const FillChain ch = FillChain{}, unsigned long long* prof_acc = nullptr, unsigned long long* prof_last_p = nullptr)
{
    const int nv = GEN ? g.nv : NV;                  // voxels per metavoxel edge (a literal in the fixed-count kernels)
    const int px = GEN ? min(px_in, nv - 1) : px_in, py = GEN ? min(py_in, nv - 1) : py_in;
    const bool owns = !GEN || (px_in < nv && py_in < nv);      // this lane owns a voxel column of the brick
#if VPFX_PROBE == 9
    unsigned long long& prof_last = *prof_last_p;      // (profiling builds run the LDS kernel only)
#endif
    constexpr bool EXACT = MATH == 1, DONE = MATH == 2;
    constexpr int CH = NV < 32 ? NV : 32;            // slices per register chunk
    constexpr int PIPE = TAB == 0 ? VPFX_FILL_PIPE : VPFX_FILL_PIPE_LDS;   // footprint loads in flight per wave
    static_assert(TAB == 0 || !EXACT, "the LDS (R8) path is default-math only: EXACT keeps the oracle's f32 table arithmetic");
    const float lds_bias = (float)(f.lds_pitch + 1) + (float)lds_base;
    const float Dk = TAB == 0 ? f.D : VPFX_BYTE_DENORM ? f.D_over_255 * 8388608.0f /* 2^23: cube_shade<BYTES> */ : f.D_over_255;
    const int LW = g.Nx * nv;
    const size_t lmi = (size_t)(py + yy * nv) * LW + (px + xx * nv);

    // get_voxel_world_pos: normPos = ((svPos - nv/2)/nv, (slice - nv/2)/nv)             Fill.shader:96-107
    const float fnv = (float)nv;
    const float nx = (((float)px + 0.5f) - fnv / 2.0f) / fnv;
    const float ny = (((float)py + 0.5f) - fnv / 2.0f) / fnv;
    const float nz = (0.0f - fnv / 2.0f) / fnv;
    const float dm = p_depthmap ? p_depthmap[lmi] : 1.0f;                                // tex2D(_LightDepthMap) :217
    const float lsSceneDepth = (dm - f.bq) * f.inv_a;                                    // :218-219

    float prop = (MODE == 0 && p_light_in) ? p_light_in[lmi] : 1.0f;                     // GL.Clear(Color.red) VPR.cs:499
    float one_minus_D = f.one_minus_D;
    asm volatile("" : "+v"(one_minus_D));                                                // keep it in a VGPR (see cube_shade)
    float smooth_c1 = 3.3333333f;
    asm volatile("" : "+v"(smooth_c1));

    for (int zz = zz_a; zz < zz_b; ++zz) {                                               // z-major = draw order VPR.cs:505
        const int mi = (zz * g.Ny + yy) * g.Nx + xx;
        const int bi = p_brick_index[mi];
        if (bi < 0) continue;                                                            // empty MV skipped    VPR.cs:511
        const int ord = CHAIN ? ch.ord[mi] : 0;
        const int off = p_offsets[mi];
        const int n = p_offsets[mi + 1] - off;
        const float mvx = p_mvPos[3 * mi], mvy = p_mvPos[3 * mi + 1], mvz = p_mvPos[3 * mi + 2];
        // _MetavoxelToWorld = TRS(mvPos, lightRot, sb)                                                  VPR.cs:596
        const float v0x = ((g.Rsb[0] * nx + g.Rsb[1] * ny) + g.Rsb[2] * nz) + mvx;
        const float v0y = ((g.Rsb[3] * nx + g.Rsb[4] * ny) + g.Rsb[5] * nz) + mvy;
        const float v0z = ((g.Rsb[6] * nx + g.Rsb[7] * ny) + g.Rsb[8] * nz) + mvz;
        // shadow index                                                                           Fill.shader:211-222
        const float ddx = v0x - f.camp[0], ddy = v0y - f.camp[1], ddz = v0z - f.camp[2];
        const float lz0 = (g.Rl[2] * ddx + g.Rl[5] * ddy) + g.Rl[8] * ddz;
        const float q = (lsSceneDepth - lz0) / g.one;
        const int shadowIndex = !(q < 2.0e9f) ? 2000000000 : (q < -2.0e9f ? -2000000000 : (int)q);

        float T = 0.f;
        if (!CHAIN) { T = (zz == 0) ? f.init_light : prop; prop = T; }                   // :224
        uint2* brick = p_bricks + (size_t)bi * nv * nv * nv;
        float2* scratch = p_dens_ao + (size_t)bi * nv * nv * nv;

        uint32_t prev_texel = 0;                          // grey z-pair bricks: the previous slice's texel of this column
#pragma unroll 1
        for (int c0 = 0; c0 < nv; c0 += CH) {
            typename AccArr<CH>::type dens, ao;
            AccArr<CH>::clear(dens, ao);                                                 // "clear it"  :178-181

            // Vectorised pre-cull: 64 particles of the MV's list at a time, one per lane, sphere vs. this wave's
            // 8x8-column x CH-slice box in voxel units (conservative); survivors are then taken in list order
            // (ascending particle index = the reference's summation order) through the wave-uniform path below.
#pragma unroll 1
            VPFX_TICK(0);                                                                // unit header
            for (int base = 0; base < (VPFX_PROBE == 7 ? 0 : n); base += 64) {          // (probe 7: no particles at all)
            int pid_l = 0;
            bool near = false;
            if (base + lane < n) {
                pid_l = p_ids[off + base + lane];
                const float4 w = p_ws[pid_l];
                const float dx = w.x - mvx, dy = w.y - mvy, dz = w.z - mvz;
                const float sc_v = fnv * g.inv_sb;                                      // voxels per world unit
                // voxel-index coordinates: column centres sit at px + 0.5, slices at s (no half-voxel offset in z, Q4)
                const float cx = ((g.Rl[0] * dx + g.Rl[3] * dy) + g.Rl[6] * dz) * sc_v + 0.5f * fnv;
                const float cy = ((g.Rl[1] * dx + g.Rl[4] * dy) + g.Rl[7] * dz) * sc_v + 0.5f * fnv;
                const float cz = ((g.Rl[2] * dx + g.Rl[5] * dy) + g.Rl[8] * dz) * sc_v + 0.5f * fnv;
                const float rv = 0.5f * w.w * sc_v * 1.002f + 0.05f;
                const float bx0 = (float)(px_in - (lane & 7)) + 0.5f, by0 = (float)(py_in - (lane >> 3)) + 0.5f;   // the tile's first column / row
                const float ex = fmaxf(fmaxf(bx0 - cx, cx - (bx0 + 7.0f)), 0.f);
                const float ey = fmaxf(fmaxf(by0 - cy, cy - (by0 + 7.0f)), 0.f);
                const float ez = fmaxf(fmaxf((float)c0 - cz, cz - (float)(c0 + CH - 1)), 0.f);
                near = (ex * ex + ey * ey) + ez * ez <= rv * rv;
            }
            unsigned long long todo = __builtin_amdgcn_ballot_w64(near);
            VPFX_TICK(1);                                                                // pre-cull
#pragma unroll 1
            while (todo) {
                const int jl = __builtin_ctzll(todo);
                todo &= todo - 1;
                const int pid = __builtin_amdgcn_readlane(pid_l, jl);
                const float* r = p_rec + 16 * (size_t)pid;
                const float r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3], r4 = r[4], r5 = r[5], r6 = r[6], r7 = r[7];
                const float r8 = r[8], r9 = r[9], r10 = r[10], r11 = r[11];
                const float opacity = (f.fade == 1) ? r[12] : 1.0f;
                const float Bx = r[13], By = r[14], Bz = r[15];      // W2P_linear * dstep, per particle per frame (k_bin)
                // ps(s) = A + s*B (arithmetic spec 4.4): A = W2P*(v0,1), B = W2P_linear * dstep
                const float Ax = fmaf(r2, v0z, fmaf(r1, v0y, fmaf(r0, v0x, r3)));
                const float Ay = fmaf(r6, v0z, fmaf(r5, v0y, fmaf(r4, v0x, r7)));
                const float Az = fmaf(r10, v0z, fmaf(r9, v0y, fmaf(r8, v0x, r11)));
                // conservative slice interval of this lane's column (culling only; the exact test follows)
                const float ka = fmaf(Bz, Bz, fmaf(By, By, Bx * Bx));
                const float kh = fmaf(Az, Bz, fmaf(Ay, By, Ax * Bx));
                const float kc = fmaf(Az, Az, fmaf(Ay, Ay, Ax * Ax)) - 0.25f;
                const float disc = fmaf(kh, kh, -ka * kc) + 2.0e-3f * ka;
                const float inv_a = __builtin_amdgcn_rcpf(ka);
                const float sq = __builtin_amdgcn_sqrtf(fmaxf(disc, 0.f)) * inv_a;
                const float sc = -kh * inv_a;
                int lo = (int)ceilf(sc - sq), hi = (int)floorf(sc + sq);
                lo = max(lo, c0); hi = min(hi, GEN ? min(c0 + CH, nv) - 1 : c0 + CH - 1);
                uint32_t m = 0;
                if (disc >= 0.f && lo <= hi) {
                    const int w = hi - lo + 1;
                    m = (w >= 32 ? 0xffffffffu : ((1u << w) - 1u)) << (lo - c0);
                }
                const uint32_t wm = wave_or(m);
                VPFX_TICK(2);                                                            // per-particle set-up
                if (wm == 0 || VPFX_PROBE == 5) continue;          // (probe 5: per-particle set-up only, no slice loop)
                const int s_first = __builtin_ctz(wm), s_last = 31 - __builtin_clz(wm);
                // Two-stage software pipeline over the slices of the range (>= 99 % of them contain a covered voxel):
                //   stage 1: coverage test, cube addressing, footprint load ISSUED (one load per slice, every lane; lanes
                //            without a covered voxel fetch entry 0, an L1 hit)
                //   stage 2: wait for that load only, bilinear + smoothstep + accumulate
                // with PIPE register sets, so that PIPE footprint loads are in flight while the oldest slice is shaded.  hipcc
                // cannot express "wait for the oldest of N loads" here (it emits vmcnt(0) around exec-masked regions), so the
                // load and its wait are inline asm: loads return in order, hence vmcnt(N-1) == "the oldest one has landed".
                using Q = typename std::conditional<TAB == 0, f32x4, QuadU8>::type;
                auto stage1 = [&](int s, float& tx, float& ty, float& d2, bool& hit, Q& q) {
                    const float fs = (float)(c0 + s);
                    const float psx = fmaf(fs, Bx, Ax), psy = fmaf(fs, By, Ay), psz = fmaf(fs, Bz, Az);
                    d2 = fmaf(psz, psz, fmaf(psy, psy, psx * psx));
                    hit = d2 <= 0.25f;                                                   // Fill.shader:172,196
#if VPFX_PROBE == 3
                    tx = psx; ty = psy; const unsigned qi = lds_base + (unsigned)(lane * 4);
#else
                    const unsigned qi = cube_address<EXACT, TAB, DONE>(f, psx, psy, psz, tx, ty, lds_bias);
#endif
                    if constexpr (TAB == 0) {
                        const unsigned off = hit ? qi : 0u;                              // byte offset into the footprint table
                        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(q) : "v"(off), "s"(p_cubequads) : "memory");
                    } else if constexpr (TAB == 1) {
                        const unsigned off = qi;   // every lane reads: any direction addresses inside the table (a select for the lanes without a covered voxel cost more)
#if VPFX_PROBE == 2
                        q.a = q.b = q.c = q.d = off & 255u; asm volatile("" : "+v"(q.a), "+v"(q.b), "+v"(q.c), "+v"(q.d));
#else
                        asm volatile("ds_read_u8 %0, %4\n\tds_read_u8 %1, %4 offset:" VPFX_STR(VPFX_LDS_PITCH_128) "\n\tds_read_u8 %2, %4 offset:1\n\tds_read_u8 %3, %4 offset:" VPFX_STR(VPFX_LDS_PITCH_128) "+1"
                                     : "=&v"(q.a), "=&v"(q.b), "=&v"(q.c), "=&v"(q.d) : "v"(off) : "memory");
#endif
                    } else {
                        const unsigned off = hit ? qi : lds_base;
                        const unsigned off2 = off + (unsigned)f.lds_pitch;                // the row below
                        asm volatile("ds_read_u8 %0, %4\n\tds_read_u8 %1, %5\n\tds_read_u8 %2, %4 offset:1\n\tds_read_u8 %3, %5 offset:1"
                                     : "=&v"(q.a), "=&v"(q.b), "=&v"(q.c), "=&v"(q.d) : "v"(off), "v"(off2) : "memory");
                    }
                };
                auto stage2 = [&](int s, float tx, float ty, float d2, bool hit, const Q& q) {
                    if (hit) {
                        float den, net;
                        float4 qf;
                        if constexpr (TAB == 0) qf = make_float4(q[0], q[1], q[2], q[3]);
#if VPFX_BYTE_DENORM
                        else qf = make_float4(__uint_as_float(q.a), __uint_as_float(q.b), __uint_as_float(q.c), __uint_as_float(q.d));   // the bytes as denormals
#else
                        else qf = make_float4((float)q.a, (float)q.b, (float)q.c, (float)q.d);   // bytes 0..255; 1/255 is folded into Dk
#endif
#if VPFX_PROBE == 1          // what-if timing probes (wrong results): 1 = no shading, 2 = no LDS reads, 3 = no cube addressing
                        den = tx + qf.x; net = ty + qf.y + qf.z + qf.w;
#else
                        cube_shade<EXACT, DONE, TAB != 0 && VPFX_BYTE_DENORM>(f, one_minus_D, qf, tx, ty, d2, opacity, den, net, Dk, smooth_c1);
#endif
                        AccArr<CH>::add_max(dens, ao, s, den, net);                      // :200-201
                    }
                };
                // Slices are processed in groups of 2*D (D = loads in flight) with D register sets; inside a group nothing in
                // flight crosses a branch or a loop back-edge (the compiler may copy registers there, and a copy of a register
                // whose load has not landed would read stale data).  Remainders fall through to smaller groups.
                int s_next = s_first;
                auto group = [&](auto depth) {
                    constexpr int D = decltype(depth)::value;
                    const int s = s_next;
                    float tx[D], ty[D], d2[D]; bool hit[D]; Q q[D];
#pragma unroll
                    for (int i = 0; i < D; ++i) stage1(s + i, tx[i], ty[i], d2[i], hit[i], q[i]);
#pragma unroll
                    for (int i = 0; i < D; ++i) {
                        if constexpr (TAB == 0) wait_vm<D - 1>(q[i]); else wait_lgkm<VPFX_LDS_READS * (D - 1)>(q[i]);
                        stage2(s + i, tx[i], ty[i], d2[i], hit[i], q[i]);
                        stage1(s + D + i, tx[i], ty[i], d2[i], hit[i], q[i]);
                    }
#pragma unroll
                    for (int i = 0; i < D; ++i) {
                        if constexpr (TAB == 0) wait_vm_dyn<D>(i, q[i]); else wait_lgkm_dyn<D>(i, q[i]);
                        stage2(s + D + i, tx[i], ty[i], d2[i], hit[i], q[i]);
                    }
                    s_next = s + 2 * D;
                };
                // full groups of 2*PIPE slices, then (at most) one group of each smaller depth: the remainder is < 2*PIPE
                // slices, so depths PIPE-1 .. 1 and one single slice cover it.  (Sequential ifs on purpose: an else-if chain
                // or a loop over the depth made hipcc keep every depth's register sets alive at once.)
                static_assert(PIPE >= 2 && PIPE <= 6, "remainder schedule written for PIPE = 2..6");
                static_assert(TAB == 0 || PIPE <= 4, "lgkmcnt is a 4-bit counter: at most 4 slices (16 ds_read) in flight");
#pragma unroll 1
                while (s_next + 2 * PIPE - 1 <= s_last) group(std::integral_constant<int, PIPE>{});
                if constexpr (PIPE >= 6) { if (s_next + 9 <= s_last) group(std::integral_constant<int, 5>{}); }
                if constexpr (PIPE >= 5) { if (s_next + 7 <= s_last) group(std::integral_constant<int, 4>{}); }
                if constexpr (PIPE >= 4) { if (s_next + 5 <= s_last) group(std::integral_constant<int, 3>{}); }
                if constexpr (PIPE >= 3) { if (s_next + 3 <= s_last) group(std::integral_constant<int, 2>{}); }
                if (s_next + 1 <= s_last) group(std::integral_constant<int, 1>{});
                if (s_next <= s_last) {
                    const int s = s_next;
                    float tx, ty, d2; bool hit; Q q;
                    stage1(s, tx, ty, d2, hit, q);
                    if constexpr (TAB == 0) wait_vm<0>(q); else wait_lgkm<0>(q);
                    stage2(s, tx, ty, d2, hit, q);
                }
                VPFX_TICK(3);                                                            // covered-slice loop
            }
            }

            if (CHAIN && c0 == 0) {
                // the light that reaches this metavoxel: what the column's previous occupied metavoxel handed on (its unit was claimed earlier)
#if VPFX_PROBE == 9
                if (ord > 0) {
                    unsigned polls = 0;
                    prop = chain_wait(ch.words + lmi, ch.tag_base + (uint32_t)ord + ch.wait_bias, ch.error, ch.spin_limit, &polls);
                    const unsigned long long again = __builtin_amdgcn_ballot_w64(polls > 0);
                    unsigned mx = polls;
                    for (int o = 32; o; o >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, o));
                    prof_acc[8] += 1; prof_acc[9] += again != 0; prof_acc[10] += mx;
                }
#else
                if (ord > 0) prop = chain_wait(ch.words + lmi, ch.tag_base + (uint32_t)ord + ch.wait_bias, ch.error, ch.spin_limit);
#endif
                T = (zz == 0) ? f.init_light : prop;                                     // :224
                prop = T;
                VPFX_TICK(4);                                                            // chain wait
            }
            // propagate + store this chunk                                               Fill.shader:231-269
            // volumeTex[int3(xy, slice)]: RGBA16F, or -- grey ambient: r = g = b bit for bit -- the z-pair entry (luminance | density)(z), (z + 1).
            // The format is wave-uniform: the branch is taken once per chunk, outside the 32 unrolled slices.
            auto propagate_store = [&](auto grey_tag) {
                constexpr bool GREYB = decltype(grey_tag)::value;
                constexpr bool NT_BRICKS = NV == 64 && !GEN;                             // see store_brick
#pragma unroll
                for (int s = 0; s < CH; ++s) {
                    const int sg = c0 + s;
                    if (GEN && sg >= nv) break;                                          // (the last chunk of a run-time voxel count)
                    const bool inShadow = sg >= shadowIndex;
                    if (inShadow) T = 0.0f;
                    else if (sg < f.border_index) prop = T;
                    const size_t vi = ((size_t)sg * nv + py) * nv + px;
                    if (MODE == 0) {
                        const float cr = 0.4f * T + f.amb[0] * ao[s];
                        if (GREYB) {
                            // entry(z) = texel(z), texel(z + 1): stored one slice late, when the slice behind it is known
                            const uint32_t cur = pack_half2(cr, dens[s]);
                            if (owns && sg > 0) store_brick<NT_BRICKS>(brick + (vi - (size_t)nv * nv), prev_texel, cur);
                            if (owns && sg == nv - 1) store_brick<NT_BRICKS>(brick + vi, cur, cur);   // (the last slice is never a footprint's z0)
                            prev_texel = cur;
                        } else {
                            const float cg = 0.4f * T + f.amb[1] * ao[s];
                            const float cb = 0.4f * T + f.amb[2] * ao[s];
                            if (owns) store_brick<NT_BRICKS>(brick + vi, pack_half2(cr, cg), pack_half2(cb, dens[s]));
                        }
                    } else {
                        if (owns) scratch[vi] = make_float2(dens[s], ao[s]);
                    }
                    T *= 1.0f / (1.0f + dens[s]);                                        // rcp(1 + density) :244
                }
            };
            if (VPFX_PROBE == 6) { if (dens[lane & 31] == 123.f) brick[0] = make_uint2(0, 0); }      // (probe 6: no propagate + store)
            else if (MODE == 0 && f.grey) propagate_store(std::true_type{}); else propagate_store(std::false_type{});
            VPFX_TICK(5);                                                                // propagate + store
        }
        if (CHAIN) {
            // the column's last occupied metavoxel writes the light map (every other value of the column is only ever seen by the next unit)
            if (ord + 1 == ch.colcount[yy * g.Nx + xx]) { if (owns) p_light_out[lmi] = prop; }   // lightPropogationTex[..] :250
            else if (owns) chain_publish(ch.words + lmi, prop, ch.tag_base + (uint32_t)ord + 1u);
        }
    }
    if (!CHAIN && owns) p_light_out[lmi] = prop;                                         // lightPropogationTex[..] :250
}

// Launch shape of the global-table path: workgroup = 16x16 voxel columns of one MV column (4 waves, each an 8x8 tile); MV columns
// heaviest first.
// CHAIN: one workgroup per (occupied metavoxel, 16x16-column tile), metavoxels z-major, light handed on through FillChain; the unit is
// taken from a counter when the workgroup STARTS (not from blockIdx: nothing guarantees dispatch order), so a unit's producer always
// started before it.  Otherwise one workgroup walks the tile's whole column (the per-metavoxel entry point, whose light goes through the
// light map like the reference's UAV).
template <int NV, int MATH, int MODE, bool CHAIN, bool GEN = false>
__global__ void __launch_bounds__(256, VPFX_FILL_WAVES)
k_fill(GridConsts g, FillConsts f, FILL_PTR_PARAMS, FillChain ch, int* __restrict__ p_counter)
{
    const int TW = GEN ? (g.nv + 15) >> 4 : NV / 16; // 16x16-column tiles per MV edge
    const int TPM = TW * TW;
    int unit = blockIdx.x;
    if (CHAIN) {
        __shared__ int sh_unit;
        if (threadIdx.x == 0) sh_unit = atomicAdd(p_counter, 1);
        __syncthreads();
        unit = sh_unit;
    }
    const int tile = unit % TPM;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int px = (tile % TW) * 16 + (wave & 1) * 8 + (lane & 7);      // wave = 8x8 columns: most compact footprint,
    const int py = (tile / TW) * 16 + (wave >> 1) * 8 + (lane >> 3);    // highest lane utilisation in covered slices
    if (GEN && (px - (lane & 7) >= g.nv || py - (lane >> 3) >= g.nv)) return;   // a wave whose whole 8x8 tile lies beyond the brick (no barrier follows)
    if (CHAIN) {
        const int mi = ch.occ_list[unit / TPM];
        const int nxy = g.Nx * g.Ny, zz = mi / nxy, col = mi - zz * nxy;
        fill_tile<NV, MATH, MODE, 0, true, GEN>(g, f, p_mvPos, p_offsets, p_ids, p_rec, p_brick_index, p_colorder, p_cubequads, p_depthmap, p_light_in,
                                            p_light_out, p_bricks, p_dens_ao, p_ws, col % g.Nx, col / g.Nx, px, py, lane, 0u, zz, zz + 1, ch);
    } else {
        const int col = p_colorder[unit / TPM];
        fill_tile<NV, MATH, MODE, 0, false, GEN>(g, f, p_mvPos, p_offsets, p_ids, p_rec, p_brick_index, p_colorder, p_cubequads, p_depthmap, p_light_in,
                                      p_light_out, p_bricks, p_dens_ao, p_ws, col % g.Nx, col / g.Nx, px, py, lane, 0u, g.z0, g.z1);
    }
}

// R8 cube maps (the reference's own asset format): the whole map, 6 (S+2)^2 bytes with the clamp border replicated (99 KB at
// S = 128), lives in LDS, so the per-voxel footprint gather never leaves the CU -- on the global table that gather, not the
// arithmetic, is what k_fill waits for (one wave-wide divergent load per covered slice through the CU's single L1/TA path).
// One PERSISTENT workgroup of 16 waves per CU (4 per SIMD) loads the table once; every wave then pulls (metavoxel, 8x8-column tile)
// units from a global work counter on its own.
template <int NV, int MODE, int TAB, bool DONE, bool GEN = false>
__global__ void __launch_bounds__(64 * (GEN ? VPFX_FILL_LDS_WAVES_GEN : VPFX_FILL_LDS_WAVES))
k_fill_lds(GridConsts g, FillConsts f, FILL_PTR_PARAMS, const uint32_t* __restrict__ p_cube_u8, int table_dwords, int* __restrict__ p_counter,
           int nitems, FillChain ch, int claim_log2)
{
    extern __shared__ uint32_t lds_cube[];
#if VPFX_FILL_CLAIM > 1
    __shared__ unsigned s_ticket;
    __shared__ unsigned long long s_block[VPFX_FILL_CLAIM_RING];           // (block + 1) << 32 | first unit of the block; 0 = not fetched yet
    if (threadIdx.x < VPFX_FILL_CLAIM_RING) s_block[threadIdx.x] = 0ull;
    if (threadIdx.x == 64) s_ticket = 0u;
    __syncthreads();
    if (threadIdx.x == 0) s_block[0] = (1ull << 32) | (unsigned)atomicAdd(p_counter, 1 << claim_log2);     // block 0 (in flight during the table copy)
#endif
    for (int i = threadIdx.x; i < table_dwords; i += 64 * (GEN ? VPFX_FILL_LDS_WAVES_GEN : VPFX_FILL_LDS_WAVES)) lds_cube[i] = p_cube_u8[i];
    __syncthreads();
    const unsigned lds_base = (unsigned)(size_t)lds_cube;       // low half of the flat address of an LDS object = its LDS byte offset
    const int T8 = GEN ? (g.nv + 7) >> 3 : NV / 8, TPC = T8 * T8; // 8x8-column tiles per MV column
    const int lane = threadIdx.x & 63;
#if VPFX_FILL_CLAIM > 1
    // Small launches (fewer units than the GPU has wave slots: the reference's own scene, config 1) run with SMALLER blocks and as many
    // working waves per workgroup as a block has units, the other waves leaving once the table is in LDS: 1 248 units are then spread over
    // every CU at one or two waves per SIMD instead of filling 78 CUs at four (launch_fill_lds_variant picks claim_log2).
    if ((int)(threadIdx.x >> 6) >= (1 << claim_log2)) return;
#endif
#if VPFX_PROBE == 9
    unsigned long long prof_acc[12] = {};
    const unsigned long long prof_t0 = __builtin_amdgcn_s_memtime();
    unsigned long long prof_last = prof_t0;
#endif
    for (;;) {
#if VPFX_FILL_CLAIM > 1
        // Workgroup-level claim (round 4).  One device-scope atomicAdd per UNIT on one address hands units out at 11.7 ns apiece (181 k units:
        // a 2.12 ms floor under the kernel, profiles/r03_fill_whatif_C3_r8.txt p5 == p7).  Here the 16 waves of the workgroup draw TICKETS
        // from an LDS counter; ticket t is slot t % CLAIM of the workgroup's block t / CLAIM, a block being CLAIM consecutive units taken
        // from the global counter with ONE atomicAdd.  The wave that draws slot PREF of block b fetches block b + 1 -- after it has seen
        // block b's base, so a workgroup's blocks ascend -- and publishes (b + 2) << 32 | base in a ring; a wave whose block is not there
        // yet polls LDS (a memory round trip at most).  Unlike a per-wave batch the CLAIM units of a block start as waves come free, one
        // ticket apart, and a fetched block waits (CLAIM - PREF) tickets at most, so the z-major hand-out the chain relies on is kept:
        // the lowest unfinished unit is either running or in a block whose workgroup's waves all run LOWER units (tickets and blocks both
        // ascend), which finish without waiting for anything unfinished and then draw it.
        const unsigned CL = 1u << claim_log2, PREF = VPFX_FILL_CLAIM_PREF < CL ? VPFX_FILL_CLAIM_PREF : CL - 1;
        unsigned tk = 0;
        if (lane == 0) tk = __hip_atomic_fetch_add(&s_ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        tk = __builtin_amdgcn_readfirstlane(tk);
        const unsigned blk = tk >> claim_log2, slot = tk & (CL - 1u);
        unsigned long long bw = __hip_atomic_load(&s_block[blk & (VPFX_FILL_CLAIM_RING - 1)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        for (unsigned spins = 0; (unsigned)(bw >> 32) != blk + 1u; ++spins) {
            // a tag beyond the awaited one would mean the ring lapped a wave that sat RING x CLAIM tickets between two instructions: report, never mis-assign
            // (the bound is the ring's own: the chain test hook shortens ch.spin_limit to a few polls)
            // (base = nitems: every slot of the block then fails the `item >= nitems` test below and the wave leaves; 0x7fffffff + slot overflowed
            //  into a negative unit index for slot > 0 -- ADVICE r4)
            if ((unsigned)(bw >> 32) > blk + 1u || spins > VPFX_CHAIN_SPIN_LIMIT) { *ch.error = 1; bw = ((unsigned long long)(blk + 1u) << 32) | (unsigned long long)(unsigned)nitems; break; }
            __builtin_amdgcn_s_sleep(1);
            bw = __hip_atomic_load(&s_block[blk & (VPFX_FILL_CLAIM_RING - 1)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (slot == PREF) {
            int nb = 0;
            if (lane == 0) {
                nb = atomicAdd(p_counter, (int)CL);
                __hip_atomic_store(&s_block[(blk + 1u) & (VPFX_FILL_CLAIM_RING - 1)], ((unsigned long long)(blk + 2u) << 32) | (unsigned)nb,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        const int item = (int)((unsigned)bw & 0x7fffffffu) + (int)slot;
        if (item >= nitems) break;
#else
        int item = 0;
        if (lane == 0) item = atomicAdd(p_counter, 1);
        item = __builtin_amdgcn_readfirstlane(item);
        if (item >= nitems) break;
#endif
        const int sub = item % TPC;
        const int px = (sub % T8) * 8 + (lane & 7), py = (sub / T8) * 8 + (lane >> 3);
        // unit = (occupied metavoxel, tile), metavoxels z-major (a unit's producer is always claimed before it)
        const int mi = ch.occ_list[item / TPC];
        const int nxy = g.Nx * g.Ny, zz = mi / nxy, col = mi - zz * nxy;
        const int xx = col % g.Nx, yy = col / g.Nx;
        fill_tile<NV, DONE ? 2 : 0, MODE, TAB, true, GEN>(g, f, p_mvPos, p_offsets, p_ids, p_rec, p_brick_index, p_colorder, p_cubequads, p_depthmap,
                                              p_light_in, p_light_out, p_bricks, p_dens_ao, p_ws, xx, yy, px, py, lane, lds_base, zz, zz + 1, ch
#if VPFX_PROBE == 9
                                              , prof_acc, &prof_last);
        { const unsigned long long t = __builtin_amdgcn_s_memtime(); prof_acc[6] += t - prof_last; prof_last = t; }   // publish + tail of the unit
#else
                                              );
#endif
    }
#if VPFX_PROBE == 9
    prof_acc[7] = __builtin_amdgcn_s_memtime() - prof_t0;
    if (lane == 0) for (int i = 0; i < 12; ++i) atomicAdd(&g_fill_prof[i], prof_acc[i]);
#endif
}

// Second half of the split (multi-GPU) fill: stream density/ao back, propagate with the true incoming light.
template <int NV, bool GEN = false>
__global__ void __launch_bounds__(256)
k_fill_finish(GridConsts g, FillConsts f, FILL_PTR_PARAMS, const float* __restrict__ p_tau_all /* nullable */, int n_before, size_t plane)
{
    const int nv = GEN ? g.nv : NV;                  // (GEN: run-time voxel count, see fill_tile)
    const int TW = GEN ? (nv + 15) >> 4 : NV / 16;
    const int TPM = TW * TW;
    const int col = blockIdx.x / TPM;
    const int tile = blockIdx.x % TPM;
    const int xx = col % g.Nx, yy = col / g.Nx;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // a streaming pass: wave = 16 columns x 4 rows, so that every row a wave touches is one full 128-byte line of the
    // scratch (float2) and of the brick (4 x f16)
    const int px = (tile % TW) * 16 + (lane & 15);
    const int py = (tile / TW) * 16 + wave * 4 + (lane >> 4);
    if (GEN && (px >= nv || py >= nv)) return;       // columns of an overhanging tile (a streaming pass: no cross-lane step)
    const int LW = g.Nx * nv;
    const size_t lmi = (size_t)(py + yy * nv) * LW + (px + xx * nv);
    const float fnv = (float)nv;
    const float nx = (((float)px + 0.5f) - fnv / 2.0f) / fnv;
    const float ny = (((float)py + 0.5f) - fnv / 2.0f) / fnv;
    const float nz = (0.0f - fnv / 2.0f) / fnv;
    const float dm = p_depthmap ? p_depthmap[lmi] : 1.0f;
    const float lsSceneDepth = (dm - f.bq) * f.inv_a;
    float prop = p_light_in ? p_light_in[lmi] : 1.0f;
    if (p_tau_all) {
        // incoming light = product of the transmittance maps of the slabs nearer the light, in slab order (the gathered
        // [world][LH][LW] buffer of the all-gather, straight from the collective: no intermediate product tensor)
        for (int j = 0; j < n_before; ++j) prop = j == 0 ? p_tau_all[lmi] : prop * p_tau_all[(size_t)j * plane + lmi];
    }
    for (int zz = g.z0; zz < g.z1; ++zz) {
        const int mi = (zz * g.Ny + yy) * g.Nx + xx;
        const int bi = p_brick_index[mi];
        if (bi < 0) continue;
        const float mvx = p_mvPos[3 * mi], mvy = p_mvPos[3 * mi + 1], mvz = p_mvPos[3 * mi + 2];
        const float v0x = ((g.Rsb[0] * nx + g.Rsb[1] * ny) + g.Rsb[2] * nz) + mvx;
        const float v0y = ((g.Rsb[3] * nx + g.Rsb[4] * ny) + g.Rsb[5] * nz) + mvy;
        const float v0z = ((g.Rsb[6] * nx + g.Rsb[7] * ny) + g.Rsb[8] * nz) + mvz;
        const float ddx = v0x - f.camp[0], ddy = v0y - f.camp[1], ddz = v0z - f.camp[2];
        const float lz0 = (g.Rl[2] * ddx + g.Rl[5] * ddy) + g.Rl[8] * ddz;
        const float q = (lsSceneDepth - lz0) / g.one;
        const int shadowIndex = !(q < 2.0e9f) ? 2000000000 : (q < -2.0e9f ? -2000000000 : (int)q);
        float T = (zz == 0) ? f.init_light : prop;
        prop = T;
        uint2* brick = p_bricks + (size_t)bi * nv * nv * nv;
        const float2* scratch = p_dens_ao + (size_t)bi * nv * nv * nv;
        uint32_t prev_texel = 0;
#pragma unroll 8
        for (int sg = 0; sg < nv; ++sg) {
            const size_t vi = ((size_t)sg * nv + py) * nv + px;
            const float2 da = scratch[vi];
            const bool inShadow = sg >= shadowIndex;
            if (inShadow) T = 0.0f;
            else if (sg < f.border_index) prop = T;
            const float cr = 0.4f * T + f.amb[0] * da.y;
            const float cg = 0.4f * T + f.amb[1] * da.y;
            const float cb = 0.4f * T + f.amb[2] * da.y;
            if (f.grey) {
                const uint32_t cur = pack_half2(cr, da.x);
                if (sg > 0) brick[vi - (size_t)nv * nv] = make_uint2(prev_texel, cur);
                if (sg == nv - 1) brick[vi] = make_uint2(cur, cur);
                prev_texel = cur;
            } else brick[vi] = make_uint2(pack_half2(cr, cg), pack_half2(cb, da.x));
            T *= 1.0f / (1.0f + da.x);
        }
    }
    p_light_out[lmi] = prop;
}

// the kernels' pointer block from the context (colorder: the one-column list of the per-metavoxel entry point, else unused)
inline FillPtrs fill_ptrs(const vp_ctx* c, const float* d_light_in, float* d_light_out, const int* colorder)
{
    FillPtrs P{};
    P.mvPos = c->d_mvPos; P.offsets = c->d_offsets; P.ids = c->d_ids; P.rec = c->d_rec;
    P.brick_index = c->d_brick_index; P.colorder = colorder; P.cubequads = c->d_cubequads;
    P.depthmap = c->have_depthmap ? c->d_depthmap : nullptr;
    P.light_in = d_light_in; P.light_out = d_light_out;
    P.bricks = c->d_bricks; P.dens_ao = c->d_dens_ao; P.ws = c->d_ws;
    return P;
}

// dynamic LDS above 64 KB has to be granted per kernel (and per device)
template <typename K>
int allow_big_lds(vp_ctx* c, K kernel, size_t bytes)
{
    VP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return VP_OK;
}

// Host side of a chained launch: light map preset for the columns without an occupied metavoxel (the others are overwritten by their last
// unit), a fresh tag range for the hand-off words.
// (the work counter of the persistent kernels is reset by the same launch that presets the light map: one command in front of the fill
//  instead of a kernel and a memset -- the reference scene's frame is 0.19 ms and every command costs 5-10 us of it)
int chain_begin(vp_ctx* c, const FillPtrs& P, int mode, FillChain& ch)
{
    const size_t lm = (size_t)c->g.Nx * c->g.nv * c->g.Ny * c->g.nv;
    if (mode == 0 && P.light_in) {
        if (P.light_in != P.light_out) VP_HIP(hipMemcpyAsync(P.light_out, P.light_in, lm * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    } else {
        int rc = launch_fill_value(c, P.light_out, lm, 1.0f, c->d_work_counter); if (rc) return rc;
    }
    if (mode == 0 && P.light_in) VP_HIP(hipMemsetAsync(c->d_work_counter, 0, sizeof(int), c->stream));
    const uint32_t span = (uint32_t)c->g.Nz + 1u;
    if (c->chain_seq >= 0xffffffffu / span - 1u) {              // tags would wrap: start over with cleared words
        VP_HIP(hipMemsetAsync(c->d_chain, 0, lm * sizeof(unsigned long long), c->stream));
        c->chain_seq = 0;
    }
    ch.words = c->d_chain; ch.ord = c->d_ord; ch.colcount = c->d_colcount; ch.occ_list = c->d_occ_list; ch.error = c->d_chain_err;
    const bool hook = c->test_chain_timeout;                 // VPFX_TEST_CHAIN_TIMEOUT=1 in the environment at vp_create (watchdog test)
    ch.spin_limit = hook ? 64u : VPFX_CHAIN_SPIN_LIMIT;
    ch.wait_bias = hook ? 0x40000000u : 0u;
    ch.tag_base = c->chain_seq * span;
    ++c->chain_seq;
    return VP_OK;
}

template <int NV, int MODE, int TAB, bool DONE, bool GEN = false>
int launch_fill_lds_variant(vp_ctx* c, const FillPtrs& P)
{
    const size_t bytes = cube_u8_bytes(c->cube_u8_S);
    auto kernel = k_fill_lds<NV, MODE, TAB, DONE, GEN>;
    // dynamic LDS above 64 KB is an opt-in per kernel AND per device: asked for before every launch (a cached "granted" flag would be
    // per process, and a host driving several GPUs launches the same instantiation on each of them)
    { int rc = allow_big_lds(c, kernel, 160 * 1024 - 1024 /* the kernel's static LDS: ticket counter + block ring */); if (rc) return rc; }
    const int TPC = GEN ? ((c->g.nv + 7) / 8) * ((c->g.nv + 7) / 8) : (NV / 8) * (NV / 8);
    FillChain ch{};
    const int nitems = c->h_meta.occupied * TPC;
    { int rc = chain_begin(c, P, MODE, ch); if (rc) return rc; }
    if (nitems == 0) return VP_OK;
    constexpr int WV = GEN ? VPFX_FILL_LDS_WAVES_GEN : VPFX_FILL_LDS_WAVES;
    static_assert(VPFX_FILL_CLAIM == 1 || VPFX_FILL_CLAIM >= WV, "a full block holds a unit for every wave of the workgroup (tickets are drawn as waves come free)");
    // units per block = working waves per workgroup: the full 16 once there is a block for every CU, else halved until there is (at least
    // 4 = one wave per SIMD).  Measured (profiles/r04_ab/fill_small_launches_spread_over_all_cus.txt): DEMO / C1 have 1 248 units; blocks of
    // 16 / 8 / 4 / 2 / 1 -> fill 0.128 / 0.092 / 0.090 / 0.116 / 0.147 ms (DEMO), 0.085 / 0.070 / 0.070 / 0.094 / 0.126 (C1); bit-identical bricks.
    int claim_log2 = 4;
    if (VPFX_FILL_CLAIM > 1) while (claim_log2 > VPFX_FILL_MIN_CLAIM_LOG2 && (nitems >> claim_log2) < c->num_cus) --claim_log2;
#if VPFX_AB
    { const char* sw = getenv("VPFX_FILL_CLAIM_LOG2"); if (sw && sw[0] >= '0' && sw[0] <= '4') claim_log2 = sw[0] - '0'; }
#endif
    const int per = 1 << claim_log2, nblocks = (nitems + per - 1) / per;
    const int grid = nblocks < c->num_cus ? nblocks : c->num_cus;                      // one persistent workgroup per CU
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(64 * WV), bytes, c->stream, c->g, c->fc, FILL_PTR_ARGS(P), (const uint32_t*)c->d_cube_u8,
                       (int)(bytes / 4), c->d_work_counter, nitems, ch, claim_log2);
    return VP_OK;
}

template <int NV, bool DONE, bool GEN = false>
int launch_fill_lds_nv(vp_ctx* c, int mode, const FillPtrs& P)
{
    if constexpr (GEN) {                    // the run-time-nv kernels take the table's row pitch as a value for every S (TAB = 2)
        return mode == 0 ? launch_fill_lds_variant<NV, 0, 2, DONE, true>(c, P) : launch_fill_lds_variant<NV, 1, 2, DONE, true>(c, P);
    } else {
        const bool s128 = c->cube_u8_S == 128;
        if (mode == 0) return s128 ? launch_fill_lds_variant<NV, 0, 1, DONE>(c, P) : launch_fill_lds_variant<NV, 0, 2, DONE>(c, P);
        return s128 ? launch_fill_lds_variant<NV, 1, 1, DONE>(c, P) : launch_fill_lds_variant<NV, 1, 2, DONE>(c, P);
    }
}

template <int NV, int MATH, bool GEN = false>
int launch_fill_chain(vp_ctx* c, int mode, const FillPtrs& P, const FillChain& ch, dim3 grid, int tl)
{
    const dim3 block(256);
    if (mode == 0) hipLaunchKernelGGL((k_fill<NV, MATH, 0, true, GEN>), grid, block, tl, c->stream, c->g, c->fc, FILL_PTR_ARGS(P), ch, c->d_work_counter);
    else           hipLaunchKernelGGL((k_fill<NV, MATH, 1, true, GEN>), grid, block, tl, c->stream, c->g, c->fc, FILL_PTR_ARGS(P), ch, c->d_work_counter);
    return VP_OK;
}

template <int NV, bool GEN = false>
int launch_fill_nv(vp_ctx* c, int mode, const FillPtrs& P, int math)
{
    const int nv = GEN ? c->g.nv : NV;
    const int TPM = ((nv + 15) / 16) * ((nv + 15) / 16);
    const dim3 block(256);
    if (mode == 2) {
        hipLaunchKernelGGL((k_fill_finish<NV, GEN>), dim3(c->g.Nx * c->g.Ny * TPM), block, 0, c->stream, c->g, c->fc, FILL_PTR_ARGS(P), c->finish_tau_all,
                           c->finish_n_before, (size_t)c->g.Nx * nv * c->g.Ny * nv);
        return VP_OK;
    }
    FillChain ch{};
    { int rc = chain_begin(c, P, mode, ch); if (rc) return rc; }
    if (c->h_meta.occupied == 0) return VP_OK;
    const dim3 grid(c->h_meta.occupied * TPM);
    const int tl = 0;
    return math == 1 ? launch_fill_chain<NV, 1, GEN>(c, mode, P, ch, grid, tl) : math == 2 ? launch_fill_chain<NV, 2, GEN>(c, mode, P, ch, grid, tl)
                                                                                          : launch_fill_chain<NV, 0, GEN>(c, mode, P, ch, grid, tl);
}

}  // namespace
