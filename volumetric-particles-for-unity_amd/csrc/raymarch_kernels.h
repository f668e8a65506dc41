// raymarch_kernels.h -- the RayMarch pass + inter-metavoxel blending as ONE launch, one thread per eye ray: the kernel templates and their
// launchers.  Two translation units instantiate them: raymarch.hip (numVoxelsInMetavoxel = 16 / 32 / 64 as the template constant NV) and
// raymarch_generic.hip (NV = 0: any other voxel count the reference accepts, VPR.cs:84, read from RmConsts at run time).
//
// Reference shape (VPR.cs:637-794, RM.shader:14-18,95-302): one DrawMeshNow(cube) per occupied metavoxel
// (~11k draws); every covered fragment redoes the ray set-up, marches that MV back-to-front on a ray lattice
// that is global per ray, and the ROP blends the MV's premultiplied result into particlesRT with OVER
// (zz <= zBoundary, (x,y) far->near) or UNDER (zz > zBoundary, near->far).
//
// CDNA4 shape: a wave owns an 8x8 pixel tile (neighbouring rays hit the same bricks -> L1/L2 reuse; the final
// float4 store is eight 128-byte lines).  Each thread sets its ray up once in "grid space" (MV (x,y,z) spans
// [x,x+1)^3, light-aligned), then walks the light-axis slabs -- zz is the reference's major draw order in both
// phases -- and inside a slab visits the (x,y) cells its ray crosses (integer DDA) in the reference's sorted order
// (rank table built on the host from the same keys as SortMetavoxelSlicesFarToNearFromEye).  Per MV it runs the
// reference's back-to-front sample loop with software trilinear filtering of the RGBA16F brick (no image
// hardware on gfx950) and blends in registers, front to back (the reverse of the reference's OVER phase, then its UNDER
// phase -- the same image up to rounding; the FLAGS kernel keeps the literal sequence).  Rays stop once saturated
// (1 - dst.a == 0: every later blend is a no-op).
// Bound: compulsory HBM traffic is one read of every contributing brick + one image store; the sampling itself
// is L1/L2-resident (64 B requested per sample).
#pragma once
#include <type_traits>

#include "vpfx_internal.h"

namespace {

typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

struct F4 { float x, y, z, w; };

__device__ __forceinline__ F4 unpack_texel(uint2 u)
{
    const half2_t a = __builtin_bit_cast(half2_t, u.x), b = __builtin_bit_cast(half2_t, u.y);
    return F4{(float)a[0], (float)a[1], (float)b[0], (float)b[1]};
}
// A texel pair (x0, x0+1) of one brick row: 16 bytes = {r0|g0, b0|a0, r1|g1, b1|a1} as packed halves.
struct __attribute__((aligned(8))) TexelPair { uint32_t rg0, ba0, rg1, ba1; };

// d = w * t + acc with t fp16 (low or high half of a dword), w and acc f32, f32 arithmetic: one v_fma_mix_f32 -- the mixed-precision FMA
// reads the packed halves directly, so no v_cvt_f32_f16 / unpacking is spent on the texel values.  (v_fma_mix_f32 issues at 4.3 cycles per
// wave, plain f32 mul / add / fma at 2.6 -- profiles/r02_valu_rate_probe.txt.)
#define VPFX_MIX_FMA(NAME, HI)                                                                                            \
    __device__ __forceinline__ float NAME(float w, uint32_t t, float acc)                                                 \
    {                                                                                                                     \
        float d;                                                                                                          \
        asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0," #HI ",0] op_sel_hi:[0,1,0]" : "=v"(d) : "v"(w), "v"(t), "v"(acc));  \
        return d;                                                                                                         \
    }
VPFX_MIX_FMA(mix_fma_lo, 0)
VPFX_MIX_FMA(mix_fma_hi, 1)


#ifndef VPFX_RM_PROBE
#define VPFX_RM_PROBE 0
#endif
#if VPFX_RM_PROBE == 9
// in-kernel phase timer of k_raymarch (profiling builds only, scripts/raymarch_phase_profile.py): wave-cycles by phase, all waves, 64 replicated rows
#ifndef VPFX_RM_MAIN_TU
static __device__ unsigned long long g_rm_prof[64][8];       // (the run-time-nv kernels are not what the phase timer is read for)
#else
__device__ unsigned long long g_rm_prof[64][8];
extern "C" __attribute__((visibility("default"))) int vpfx_rm_probe_read(unsigned long long* out, int reset)
{
    static unsigned long long h[64][8];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_rm_prof), sizeof(h)) != hipSuccess) return -1;
    for (int i = 0; i < 8; ++i) { out[i] = 0; for (int r = 0; r < 64; ++r) out[i] += h[r][i]; }
    if (reset) { for (auto& row : h) for (auto& x : row) x = 0; if (hipMemcpyToSymbol(HIP_SYMBOL(g_rm_prof), h, sizeof(h)) != hipSuccess) return -1; }
    return 0;
}
#endif
#define VPFX_RM_TICK(ph) do { const unsigned long long t_now_ = __builtin_amdgcn_s_memtime(); rm_prof[ph] += t_now_ - rm_last; rm_last = t_now_; } while (0)
#define VPFX_RM_PROF_ARGS , unsigned long long* rm_prof, unsigned long long& rm_last
#define VPFX_RM_PROF_PASS , rm_prof, rm_last
#define VPFX_RM_PROF_DUMMY unsigned long long rm_prof[8] = {}; unsigned long long rm_last = 0;
#else
#define VPFX_RM_TICK(ph) do { } while (0)
#define VPFX_RM_PROF_ARGS
#define VPFX_RM_PROF_PASS
#define VPFX_RM_PROF_DUMMY
#endif
#ifndef VPFX_RM_SPI
#define VPFX_RM_SPI 4          // lattice samples per loop iteration on grey bricks (see march_mv); 2 = rounds 2-3 (A/B, with VPFX_RM_WAVES_GREY=5)
#endif
#ifndef VPFX_RM_OCC_LDS
#define VPFX_RM_OCC_LDS 1       // occupancy bitmask of the grid in LDS for the cell walk (k_raymarch); 0 = A/B
#endif
#define VPFX_RM_OCC_WORDS 1024  // Nz * Ny <= 1024 and Nx <= 32 (C1 .. C4); larger grids walk on the per-cell records alone
#ifndef VPFX_RM_CELLINFO
#define VPFX_RM_CELLINFO 1      // per-cell (translation, brick slot) records for the streaming cell walk (see k_raymarch; written by k_rm_prepare): 0.967 -> 0.951 ms at C3; 0 = A/B
#endif

// Explicitly issued 16-byte loads for the two-samples-per-iteration loop: the compiler otherwise sinks the second
// sample's loads below the first sample's filter (four loads in flight instead of eight).  The asynchronous register
// write is invisible to the compiler, so wait_quad() takes the destinations as in/out operands: every use is ordered
// after the s_waitcnt.  scripts/check_fill_asm.py verifies the generated ISA (no access to a destination before its wait).
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int OFF>
__device__ __forceinline__ void issue_load16(u32x4& q, const void* p)
{
    asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(q) : "v"(p), "n"(OFF) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_quad(u32x4& a, u32x4& b, u32x4& c, u32x4& d)
{
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_pair(u32x4& a, u32x4& b)
{
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory");
}
// the footprint's second row (y + 1): the row pitch is an instruction immediate where the voxel count is the template constant NV, a pointer
// step where it is a run-time value (NV = 0)
template <int NV>
__device__ __forceinline__ void issue_load16_row1(u32x4& q, const uint2* p, int nv)
{
    if constexpr (NV != 0) issue_load16<NV * 8>(q, p); else issue_load16<0>(q, p + nv);
}
__device__ __forceinline__ TexelPair as_pair(const u32x4 q) { return TexelPair{q[0], q[1], q[2], q[3]}; }
__device__ __forceinline__ float lerpf(float a, float b, float t) { return fmaf(t, b - a, a); }

// D3D11 float -> UNORM8 -> float round trip: clamp, scale, round to nearest
__device__ __forceinline__ float unorm8(float x) { return floorf(fminf(fmaxf(x, 0.f), 1.f) * 255.0f + 0.5f) / 255.0f; }

struct RayCtx {
    float ogx, ogy, ogz;      // ray origin (csAABBStart) in grid space               (traversal only)
    float dgx, dgy, dgz;      // normalised direction in grid space                     (traversal only)
    float ivx, ivy, ivz;      // 1/dg
    float lx, ly, lz;         // linear part of mvRay.o = C2M_linear * csAABBStart      RM.shader:216
    float dx, dy, dz;         // mvRay.d = normalize(C2M_linear * csRayDir)             RM.shader:217
    float idx, idy, idz;      // 1 / mvRay.d                                            RM.shader:99
    float startz, dirz;       // camera-space z of origin / direction (for the clip + depth tests)
    float sceneDepth;
    int tCameraG;             // camera lattice index estimated in grid space (traversal clamp only)
};

// Ray set-up of one pixel (RM.shader:188-224), once per ray: the camera-space ray, its image in grid space (traversal) and in
// metavoxel space minus the per-MV translation (exactly the reference's per-draw arithmetic).
__device__ __forceinline__ RayCtx ray_setup(const RmConsts& k, int col, int row, const float* __restrict__ scene_depth)
{
    float dx = (2.0f * ((float)col + 0.5f) / (float)k.W) - 1.0f;
    const float dy = (2.0f * ((float)row + 0.5f) / (float)k.H) - 1.0f;
    dx *= k.aspect;
    const float dz = k.neg_inv_tan;
    const float inv = 1.0f / sqrtf((dx * dx + dy * dy) + dz * dz);
    const float dirx = dx * inv, diry = dy * inv, dirz = dz * inv;
    const float st = k.zMin / dirz;
    const float sx = dirx * st, sy = diry * st, sz = dirz * st;                            // csAABBStart :212
    RayCtx R;
    R.ogx = ((k.c2g[0] * sx + k.c2g[1] * sy) + k.c2g[2] * sz) + k.c2g[3];
    R.ogy = ((k.c2g[4] * sx + k.c2g[5] * sy) + k.c2g[6] * sz) + k.c2g[7];
    R.ogz = ((k.c2g[8] * sx + k.c2g[9] * sy) + k.c2g[10] * sz) + k.c2g[11];
    float gx = (k.c2g[0] * dirx + k.c2g[1] * diry) + k.c2g[2] * dirz;
    float gy = (k.c2g[4] * dirx + k.c2g[5] * diry) + k.c2g[6] * dirz;
    float gz = (k.c2g[8] * dirx + k.c2g[9] * diry) + k.c2g[10] * dirz;
    const float ginv = 1.0f / sqrtf((gx * gx + gy * gy) + gz * gz);
    R.dgx = gx * ginv; R.dgy = gy * ginv; R.dgz = gz * ginv;                               // mvRay.d :217
    R.ivx = 1.0f / R.dgx; R.ivy = 1.0f / R.dgy; R.ivz = 1.0f / R.dgz;
    R.startz = sz; R.dirz = dirz;
    R.sceneDepth = scene_depth ? scene_depth[(size_t)row * k.W + col] : 3.0e38f;
    {
        const float cx = k.camg[0] - R.ogx, cy = k.camg[1] - R.ogy, cz = k.camg[2] - R.ogz;
        R.tCameraG = (int)(sqrtf((cx * cx + cy * cy) + cz * cz) / k.mvStep);
    }
    // the ray in metavoxel space, minus the per-MV translation (exactly the reference's per-draw arithmetic)
    R.lx = (k.c2m_lin[0] * sx + k.c2m_lin[1] * sy) + k.c2m_lin[2] * sz;
    R.ly = (k.c2m_lin[3] * sx + k.c2m_lin[4] * sy) + k.c2m_lin[5] * sz;
    R.lz = (k.c2m_lin[6] * sx + k.c2m_lin[7] * sy) + k.c2m_lin[8] * sz;
    {
        const float mx = (k.c2m_lin[0] * dirx + k.c2m_lin[1] * diry) + k.c2m_lin[2] * dirz;
        const float my = (k.c2m_lin[3] * dirx + k.c2m_lin[4] * diry) + k.c2m_lin[5] * dirz;
        const float mz = (k.c2m_lin[6] * dirx + k.c2m_lin[7] * diry) + k.c2m_lin[8] * dirz;
        const float minv = 1.0f / sqrtf((mx * mx + my * my) + mz * mz);
        R.dx = mx * minv; R.dy = my * minv; R.dz = mz * minv;
        R.idx = 1.0f / R.dx; R.idy = 1.0f / R.dy; R.idz = 1.0f / R.dz;
    }

    return R;
}

// One metavoxel for one ray: RM.shader frag (166-302), arithmetic as the reference lays it out: the ray is
// expressed in THIS metavoxel's unit-cube space through _CameraToMetavoxel (mv translation column `tr`), so
// tEntry / tExit / tCamera are bit-identical to the per-draw values.  Returns false when the rasteriser would
// not have produced a fragment (or the shader's own box test misses); src is premultiplied (rgb, 1 - T).
// GREY: the brick holds (luminance, density) fp16 pairs -- the fill stores that when the ambient colour is grey (the reference's default,
// scene:9021), where r = g = b bit for bit (diffuse is the scalar 0.4 T, Fill.shader:239-241) -- as Z-PAIR entries: entry (x, y, z) =
// texel(x, y, z), texel(x, y, z + 1), 8 bytes.  One 16-byte load at (x0, y, z0) then fetches the x-pair of BOTH z planes of a trilinear
// footprint: two loads per sample instead of four (the kernel is bound by the L1's rate of wave-loads), and half the
// channels to filter.  Same image bit for bit as RGBA16F bricks.
template <int NV, bool WRAP, bool FLAGS, bool GREY>
__device__ __forceinline__ bool march_mv(const RmConsts& k, const RayCtx& R, const uint2* __restrict__ brick, const float4 tr,
                                         F4& src, int& nsamp VPFX_RM_PROF_ARGS)
{
    static_assert(!(WRAP && GREY), "grey bricks are only used with border >= 1 (the footprint never wraps)");
    const int nv = NV ? NV : k.nv;                   // NV = 0: run-time voxel count (raymarch_generic.hip)
    const float fnv = (float)nv;
    const float ox = R.lx + tr.x, oy = R.ly + tr.y, oz = R.lz + tr.z;                     // mvRay.o :216
    // IntersectBox(mvRay, -0.5, 0.5)                                                      RM.shader:95-118
    const float tbx = R.idx * (-0.5f - ox), tby = R.idy * (-0.5f - oy), tbz = R.idz * (-0.5f - oz);
    const float ttx = R.idx * (0.5f - ox), tty = R.idy * (0.5f - oy), ttz = R.idz * (0.5f - oz);
    const float tminx = fminf(ttx, tbx), tminy = fminf(tty, tby), tminz = fminf(ttz, tbz);
    const float tmaxx = fmaxf(ttx, tbx), tmaxy = fmaxf(tty, tby), tmaxz = fmaxf(ttz, tbz);
    const float t1 = fmaxf(fmaxf(tminx, tminy), fmaxf(tminx, tminz));
    const float t2 = fminf(fminf(tmaxx, tmaxy), fminf(tmaxx, tmaxz));
    if (t1 > t2) return false;
    // back-face fragment must survive near/far clipping and ZTest Less                    RM.shader:14
    const float exitDepth = -(R.startz + R.dirz * (t2 * k.s));
    if (!(exitDepth > k.nearc) || !(exitDepth <= k.farc) || !(exitDepth < R.sceneDepth)) return false;
    int tEntry = (int)ceilf(t1 / k.mvStep);                                               // :236
    const int tExit = (int)floorf(t2 / k.mvStep);                                         // :237
    const float cx = tr.x - ox, cy = tr.y - oy, cz = tr.z - oz;                           // mvCameraPos - mvRay.o :238
    const int tCamera = (int)(sqrtf((cx * cx + cy * cy) + cz * cz) / k.mvStep);           // :239
    tEntry = max(tEntry, tCamera);                                                        // :240
    float rr = 0.f, rg = 0.f, rb = 0.f, trans = 1.0f;
    // samplePos = (mvRayPos + 0.5)(1 - 2 bo) + bo, texel = samplePos*nv - 0.5 (:255-258) is affine in the lattice index:
    // texel(si) = f0 + si * fs, one FMA per axis per sample
    const float f0x = fmaf(ox + 0.5f, k.texScale, k.texBias), f0y = fmaf(oy + 0.5f, k.texScale, k.texBias),
                f0z = fmaf(oz + 0.5f, k.texScale, k.texBias);
    const float fsx = (k.mvStep * R.dx) * k.texScale, fsy = (k.mvStep * R.dy) * k.texScale, fsz = (k.mvStep * R.dz) * k.texScale;
    // tex3D(_VolumeTexture, samplePos) at lattice index si (:255-262), split into address / fetch / filter so that the
    // loads of two samples can be issued back to back before either is filtered
    struct Addr { const uint2* p; float wx, wy, wz; int ix, iy, iz; };
    struct Quad { TexelPair t00, t10, t01, t11; };    // [z][y]: texels (x0, x0+1)
    struct QuadG { uint32_t a0, a1, b0, b1, c0, c1, d0, d1; };   // grey: [z0y0], [z0y1], [z1y0], [z1y1] x (x0, x0+1), each lum | dens
    auto address = [&](float fi /* lattice index, an exact integer */) -> Addr {
        const float fx = fmaf(fi, fsx, f0x), fy = fmaf(fi, fsy, f0y), fz = fmaf(fi, fsz, f0z);
        const float x0 = floorf(fx), y0 = floorf(fy), z0 = floorf(fz);
        Addr a;
        a.wx = fx - x0; a.wy = fy - y0; a.wz = fz - z0;
        if (!WRAP) {
            // border >= 1: the 2x2x2 footprint never leaves the brick (texel coords lie in [b-0.5, nv-b-0.5]), so the
            // x-neighbours are one 16-byte load and the y / z neighbours fixed offsets from one base address.
            // (z0*NV + y0)*NV + x0 in float (exact: small integers), one conversion
            const int idx = (int)fmaf(fmaf(z0, fnv, y0), fnv, x0);
            a.p = brick + idx;                       // grey z-pair entries are 8 bytes like RGBA16F texels
            a.ix = a.iy = a.iz = 0;
        } else {
            a.p = brick;
            a.ix = (int)x0; a.iy = (int)y0; a.iz = (int)z0;
        }
        return a;
    };
    auto fetch = [&](const Addr& a) -> Quad {
        Quad q;
        if (!WRAP) {
            const uint2* p = a.p;
            q.t00 = *reinterpret_cast<const TexelPair*>(p);
            q.t10 = *reinterpret_cast<const TexelPair*>(p + nv);
            q.t01 = *reinterpret_cast<const TexelPair*>(p + nv * nv);
            q.t11 = *reinterpret_cast<const TexelPair*>(p + nv * nv + nv);
        } else {
            int ix0, iy0, iz0, ix1, iy1, iz1;                                                          // wrap = Repeat  VPR.cs:770
            if constexpr (NV != 0) {
                ix0 = a.ix & (NV - 1); iy0 = a.iy & (NV - 1); iz0 = a.iz & (NV - 1);
                ix1 = (ix0 + 1) & (NV - 1); iy1 = (iy0 + 1) & (NV - 1); iz1 = (iz0 + 1) & (NV - 1);
            } else {
                // any voxel count: texel coordinates lie in [-0.5, nv - 0.5] (the samples sit inside the unit cube), so floor() is in
                // [-1, nv - 1] up to rounding -- one conditional step either way is the full modulo
                auto wrap = [&](int i) { i += i < 0 ? nv : 0; return i >= nv ? i - nv : i; };
                ix0 = wrap(a.ix); iy0 = wrap(a.iy); iz0 = wrap(a.iz);
                ix1 = wrap(ix0 + 1); iy1 = wrap(iy0 + 1); iz1 = wrap(iz0 + 1);
            }
            const int r00 = (iz0 * nv + iy0) * nv, r10 = (iz0 * nv + iy1) * nv, r01 = (iz1 * nv + iy0) * nv, r11 = (iz1 * nv + iy1) * nv;
            const uint2 a0 = brick[r00 + ix0], a1 = brick[r00 + ix1], b0 = brick[r10 + ix0], b1 = brick[r10 + ix1];
            const uint2 c0 = brick[r01 + ix0], c1 = brick[r01 + ix1], d0 = brick[r11 + ix0], d1 = brick[r11 + ix1];
            q.t00 = TexelPair{a0.x, a0.y, a1.x, a1.y}; q.t10 = TexelPair{b0.x, b0.y, b1.x, b1.y};
            q.t01 = TexelPair{c0.x, c0.y, c1.x, c1.y}; q.t11 = TexelPair{d0.x, d0.y, d1.x, d1.y};
        }
        return q;
    };
    auto fetch_grey = [&](const Addr& a) -> QuadG {
        // entry (x, y, z) = texel(z), texel(z + 1): 16 bytes at (x0, y, z0) = [x0 z0, x0 z1, x1 z0, x1 z1]
        const TexelPair r0 = *reinterpret_cast<const TexelPair*>(a.p), r1 = *reinterpret_cast<const TexelPair*>(a.p + nv);
        return QuadG{r0.rg0, r0.rg1, r1.rg0, r1.rg1, r0.ba0, r0.ba1, r1.ba0, r1.ba1};
    };
    // Trilinear filter = bilinear weighted sum over (y, z) of the x0 texels and of the x1 texels (v_fma_mix_f32: f16 texel, f32 weight and
    // accumulator), then one lerp in x: 6 weight instructions + 8 FMAs and 2 lerp instructions per channel (the full eight-weight sum costs
    // 15 + 8 per channel).  Same value as the reference's lerp cascade up to f32 rounding (~1e-7).
    // grey: two channels (luminance in the low halves, density in the high halves)
    auto filter_grey = [&](const QuadG& q, const Addr& a) -> F4 {
        const float ay = 1.0f - a.wy, az = 1.0f - a.wz;
        const float w00 = ay * az, w10 = a.wy * az, w01 = ay * a.wz, w11 = a.wy * a.wz;         // [z][y]
        const float l0 = mix_fma_lo(w11, q.d0, mix_fma_lo(w01, q.c0, mix_fma_lo(w10, q.b0, mix_fma_lo(w00, q.a0, 0.f))));
        const float l1 = mix_fma_lo(w11, q.d1, mix_fma_lo(w01, q.c1, mix_fma_lo(w10, q.b1, mix_fma_lo(w00, q.a1, 0.f))));
        const float d0 = mix_fma_hi(w11, q.d0, mix_fma_hi(w01, q.c0, mix_fma_hi(w10, q.b0, mix_fma_hi(w00, q.a0, 0.f))));
        const float d1 = mix_fma_hi(w11, q.d1, mix_fma_hi(w01, q.c1, mix_fma_hi(w10, q.b1, mix_fma_hi(w00, q.a1, 0.f))));
        const float lum = lerpf(l0, l1, a.wx), den = lerpf(d0, d1, a.wx);
        return F4{lum, lum, lum, den};
    };
    auto filter = [&](const Quad& q, const Addr& a) -> F4 {
        const float ay = 1.0f - a.wy, az = 1.0f - a.wz;
        const float w00 = ay * az, w10 = a.wy * az, w01 = ay * a.wz, w11 = a.wy * a.wz;         // [z][y]
        const float r0 = mix_fma_lo(w11, q.t11.rg0, mix_fma_lo(w01, q.t01.rg0, mix_fma_lo(w10, q.t10.rg0, mix_fma_lo(w00, q.t00.rg0, 0.f))));
        const float r1 = mix_fma_lo(w11, q.t11.rg1, mix_fma_lo(w01, q.t01.rg1, mix_fma_lo(w10, q.t10.rg1, mix_fma_lo(w00, q.t00.rg1, 0.f))));
        const float g0 = mix_fma_hi(w11, q.t11.rg0, mix_fma_hi(w01, q.t01.rg0, mix_fma_hi(w10, q.t10.rg0, mix_fma_hi(w00, q.t00.rg0, 0.f))));
        const float g1 = mix_fma_hi(w11, q.t11.rg1, mix_fma_hi(w01, q.t01.rg1, mix_fma_hi(w10, q.t10.rg1, mix_fma_hi(w00, q.t00.rg1, 0.f))));
        const float b0 = mix_fma_lo(w11, q.t11.ba0, mix_fma_lo(w01, q.t01.ba0, mix_fma_lo(w10, q.t10.ba0, mix_fma_lo(w00, q.t00.ba0, 0.f))));
        const float b1 = mix_fma_lo(w11, q.t11.ba1, mix_fma_lo(w01, q.t01.ba1, mix_fma_lo(w10, q.t10.ba1, mix_fma_lo(w00, q.t00.ba1, 0.f))));
        const float a0 = mix_fma_hi(w11, q.t11.ba0, mix_fma_hi(w01, q.t01.ba0, mix_fma_hi(w10, q.t10.ba0, mix_fma_hi(w00, q.t00.ba0, 0.f))));
        const float a1 = mix_fma_hi(w11, q.t11.ba1, mix_fma_hi(w01, q.t01.ba1, mix_fma_hi(w10, q.t10.ba1, mix_fma_hi(w00, q.t00.ba1, 0.f))));
        return F4{lerpf(r0, r1, a.wx), lerpf(g0, g1, a.wx), lerpf(b0, b1, a.wx), lerpf(a0, a1, a.wx)};
    };
    auto blend = [&](const F4& c, float density) {
        const float bf = __builtin_amdgcn_rcpf(1.0f + density);                           // :272
        rr = fmaf(bf, rr - c.x, c.x);                                                     // lerp(color, result, bf) :274
        if (!GREY) { rg = fmaf(bf, rg - c.y, c.y); rb = fmaf(bf, rb - c.z, c.z); }        // (grey: g and b are r, copied at the end)
        trans *= bf;                                                                      // :275
    };
    // soft particles (:267-270) only touch lattice indices below tCamera + _SoftDistance; everything farther from the camera
    // (the bulk, and the part marched first: back to front, :254) runs without the fade.  Two lattice samples per iteration
    // so that eight texel-pair loads are in flight.
    const int tSoft = max(tEntry, min(tExit + 1, tCamera + k.soft));
    int si = tExit;
    float fsi = (float)si;                              // the index as a float, stepped with adds (one conversion per metavoxel, not per sample)
    VPFX_RM_TICK(2);                                    // per-metavoxel set-up (box test, lattice range, texel lattice)
    // (Measured and dropped: issuing the next two samples' eight loads before filtering the current two -- two register sets, 8-16 loads
    // in flight per wave -- 1.49 vs 1.48 ms: loads in flight per wave are not what limits the kernel.)
    // (Grey bricks, measured: four samples per iteration -- 8 loads in flight -- 1.02 ms at 4 waves/SIMD against 1.09 for two, but the
    // two-sample loop fits 5 waves/SIMD: 1.00 ms.)
#if VPFX_RM_SPI == 4
    // Four lattice samples per iteration on grey bricks (eight loads in flight), the two-sample loop below takes the remainder.  Launches with
    // few waves per SIMD (small screens: the reference's own 1024 x 768 demo, config 1) run the loop at the pace of one memory round trip per
    // iteration: 4 samples at 4 waves/SIMD against 2 at 5 -- DEMO 0.118 -> 0.103 ms, C1 0.139 -> 0.122, C2 0.460 -> 0.450, C3 0.940 = 0.940
    // (profiles/r04_ab/raymarch_four_samples_per_iteration.txt).  Same arithmetic per sample in the same order: bit-identical images.
    if constexpr (GREY && !FLAGS) {          // (the debug-view / UNORM8 kernels keep the two-sample loop: their extra state would spill)
        for (; si - 3 >= tSoft; si -= 4) {
            const Addr a0 = address(fsi), a1 = address(fsi - 1.0f), a2 = address(fsi - 2.0f), a3 = address(fsi - 3.0f);
            fsi -= 4.0f;
            u32x4 u0, u1, v0, v1, w0, w1, x0, x1;
            issue_load16<0>(u0, a0.p); issue_load16_row1<NV>(u1, a0.p, nv);
            issue_load16<0>(v0, a1.p); issue_load16_row1<NV>(v1, a1.p, nv);
            issue_load16<0>(w0, a2.p); issue_load16_row1<NV>(w1, a2.p, nv);
            issue_load16<0>(x0, a3.p); issue_load16_row1<NV>(x1, a3.p, nv);
            wait_pair<6>(u0, u1);
            const F4 c0 = filter_grey(QuadG{u0[0], u0[2], u1[0], u1[2], u0[1], u0[3], u1[1], u1[3]}, a0);
            blend(c0, c0.w);
            wait_pair<4>(v0, v1);
            const F4 c1 = filter_grey(QuadG{v0[0], v0[2], v1[0], v1[2], v0[1], v0[3], v1[1], v1[3]}, a1);
            blend(c1, c1.w);
            wait_pair<2>(w0, w1);
            const F4 c2 = filter_grey(QuadG{w0[0], w0[2], w1[0], w1[2], w0[1], w0[3], w1[1], w1[3]}, a2);
            blend(c2, c2.w);
            wait_pair<0>(x0, x1);
            const F4 c3 = filter_grey(QuadG{x0[0], x0[2], x1[0], x1[2], x0[1], x0[3], x1[1], x1[3]}, a3);
            blend(c3, c3.w);
        }
    }
#endif
    for (; si - 1 >= tSoft; si -= 2) {
        const Addr a0 = address(fsi), a1 = address(fsi - 1.0f);
        fsi -= 2.0f;
        Quad q0, q1;
        if (GREY) {
            u32x4 u0, u1, v0, v1;
            issue_load16<0>(u0, a0.p); issue_load16_row1<NV>(u1, a0.p, nv);
            issue_load16<0>(v0, a1.p); issue_load16_row1<NV>(v1, a1.p, nv);
            wait_pair<2>(u0, u1);
            const F4 c0 = filter_grey(QuadG{u0[0], u0[2], u1[0], u1[2], u0[1], u0[3], u1[1], u1[3]}, a0);
            blend(c0, c0.w);
            wait_pair<0>(v0, v1);
            const F4 c1 = filter_grey(QuadG{v0[0], v0[2], v1[0], v1[2], v0[1], v0[3], v1[1], v1[3]}, a1);
            blend(c1, c1.w);
        } else if (!WRAP) {
            u32x4 u0, u1, u2, u3, v0, v1, v2, v3;
            const uint2* z0p = a0.p + nv * nv; const uint2* z1p = a1.p + nv * nv;       // the z+1 plane is beyond the 12-bit offset
            issue_load16<0>(u0, a0.p); issue_load16_row1<NV>(u1, a0.p, nv); issue_load16<0>(u2, z0p); issue_load16_row1<NV>(u3, z0p, nv);
            issue_load16<0>(v0, a1.p); issue_load16_row1<NV>(v1, a1.p, nv); issue_load16<0>(v2, z1p); issue_load16_row1<NV>(v3, z1p, nv);
            wait_quad<4>(u0, u1, u2, u3);
            q0 = Quad{as_pair(u0), as_pair(u1), as_pair(u2), as_pair(u3)};
            const F4 c0 = filter(q0, a0);
            blend(c0, c0.w);
            wait_quad<0>(v0, v1, v2, v3);
            q1 = Quad{as_pair(v0), as_pair(v1), as_pair(v2), as_pair(v3)};
            const F4 c1 = filter(q1, a1);
            blend(c1, c1.w);
        } else {
            q0 = fetch(a0); q1 = fetch(a1);
            const F4 c0 = filter(q0, a0);
            blend(c0, c0.w);
            const F4 c1 = filter(q1, a1);
            blend(c1, c1.w);
        }
    }
    for (; si >= tEntry; --si) {
        const Addr a = address((float)si);
        F4 c;
        if constexpr (GREY) c = filter_grey(fetch_grey(a), a); else c = filter(fetch(a), a);
        const int dc = si - tCamera;
        blend(c, dc < k.soft ? c.w * ((float)dc * k.inv_soft) : c.w);
    }
    VPFX_RM_TICK(3);                                    // the sample loops
    const int ns = max(0, tExit - tEntry + 1);
    nsamp += ns;
    if (GREY) { rg = rr; rb = rr; }
    src = F4{rr, rg, rb, 1.0f - trans};                                                   // :301
    if (FLAGS && (k.flags & VP_RM_SHOW_NUM_SAMPLES)) {                                    // debug view :283-299
        src = ns < 5 ? F4{0.f, 0.2f, 0.f, 0.5f} : ns < 10 ? F4{0.f, 0.5f, 0.f, 0.5f} : ns < 20 ? F4{0.5f, 0.5f, 0.f, 0.5f}
            : ns < 30 ? F4{0.6f, 0.4f, 0.f, 0.5f} : ns < 40 ? F4{0.6f, 0.f, 0.f, 0.5f} : ns < 50 ? F4{0.8f, 0.f, 0.f, 0.5f}
            : F4{1.0f, 0.f, 0.f, 0.5f};
    }
    return true;
}

// DrawOrderColoring (RM.shader:123-138): bright green -> dull green -> bright blue -> dull blue -> bright red -> dull red along
// the submission order; integer arithmetic as in the shader.
__device__ __forceinline__ F4 draw_order_color(int order_index, int num_covered)
{
    const int per = max((int)ceilf((float)num_covered / 3.0f), 1);      // numColorsPerChannel (>= 1: an MV is being drawn)
    const int sel = order_index / per, idx = order_index % per;
    const float v = (float)(per - idx) / (float)per;
    return sel == 0 ? F4{0.f, v, 0.f, 1.f} : sel == 1 ? F4{0.f, 0.f, v, 1.f} : F4{v, 0.f, 0.f, 1.f};
}

#define RM_ORDER_MAX 8192       // super-tiles the in-LDS rank sort of the dispatch order holds (tile_cost / k_tile_rank)
// rank of 64 super-tiles per workgroup: wave w counts, for each of them, the costlier tiles among the w-th sixteenth of all
__global__ void __launch_bounds__(1024)
k_tile_rank(const float* __restrict__ cost_in, int nsuper, int* __restrict__ order)
{
    __shared__ __attribute__((aligned(16))) float cost[RM_ORDER_MAX];
    __shared__ int rank[64];
    const int npad = (nsuper + 3) & ~3;
    for (int i = threadIdx.x; i < npad; i += 1024) cost[i] = i < nsuper ? cost_in[i] : -1.0f;   // padding never outranks a tile
    if (threadIdx.x < 64) rank[threadIdx.x] = 0;
    __syncthreads();
    const int i = blockIdx.x * 64 + (threadIdx.x & 63), chunk = threadIdx.x >> 6;
    const int per = (((npad + 15) / 16) + 3) & ~3;
    const int j0 = chunk * per, j1 = min(npad, j0 + per);
    if (i < nsuper) {
        const float c = cost[i];
        int r = 0;
        for (int j = j0; j < j1; j += 4) {
            const float4 cj = *reinterpret_cast<const float4*>(&cost[j]);
            r += (cj.x > c || (cj.x == c && j < i)) ? 1 : 0;
            r += (cj.y > c || (cj.y == c && j + 1 < i)) ? 1 : 0;
            r += (cj.z > c || (cj.z == c && j + 2 < i)) ? 1 : 0;
            r += (cj.w > c || (cj.w == c && j + 3 < i)) ? 1 : 0;
        }
        atomicAdd(&rank[threadIdx.x & 63], r);
    }
    __syncthreads();
    if (threadIdx.x < 64 && i < nsuper) order[rank[threadIdx.x]] = i;
}

#if VPFX_AB   // measured and dropped (round 4, profiles/r04_ab/raymarch_wave_shape_and_xcd_affine.txt): one compact screen region per XCD cut the L2->fabric
              // read volume by only 3-8 % and cost 25-40 % in time (the regions' work is unequal whatever the estimate says); run with VPFX_RM_XCD_AFFINE=1
// XCD-affine dispatch order (round 4).  Workgroup b runs on XCD b % 8 and every XCD has its own 4 MiB L2; with super-tiles dealt out round
// robin in cost order, the super-tiles that share a brick sit on 3-6 different XCDs and every one of them pulls the brick's lines into its L2
// again (measured: 2.4 x the sampled bricks' bytes leave the memory side at C3 and at C5, profiles/r04_*).  Here the screen is cut into 8
// COMPACT regions -- contiguous runs of a Hilbert curve over the super-tile grid (curve[] from the host, fixed per resolution) -- of equal
// weight, weight = (cost share + count share) / 2: the cost share balances the XCDs' work, the count share bounds a region at
// nsuper / 4 super-tiles so that the launch can be sized without reading anything back.  Region r's super-tiles go to dispatch positions
// r, 8 + r, 16 + r, ... most expensive first; the rest of its positions hold -1 (the workgroup exits).  One workgroup; scheduling only.
__global__ void __launch_bounds__(1024)
k_tile_regions(const float* __restrict__ cost_in, const int* __restrict__ curve, int nsuper, int cap /* positions per XCD */, int* __restrict__ order)
{
    __shared__ float cost[RM_ORDER_MAX];           // along the curve
    __shared__ float pre[RM_ORDER_MAX];
    __shared__ float part[1024];
    __shared__ int start[9];
    const int t = threadIdx.x;
    for (int i = t; i < 8 * cap; i += 1024) order[i] = -1;
    float acc = 0.f;
    for (int j = t; j < nsuper; j += 1024) { const float c = cost_in[curve[j]]; cost[j] = c; acc += c; }
    part[t] = acc;
    __syncthreads();
    for (int o = 512; o; o >>= 1) { if (t < o) part[t] += part[t + o]; __syncthreads(); }
    const float total = part[0];
    __syncthreads();
    // inclusive prefix of the weights along the curve: each thread owns a contiguous chunk
    const int per = (nsuper + 1023) / 1024, j0 = min(t * per, nsuper), j1 = min(j0 + per, nsuper);
    const float wc = total > 0.f ? 0.5f / total : 0.f, wn = (total > 0.f ? 0.5f : 1.0f) / (float)nsuper;
    float run = 0.f;
    for (int j = j0; j < j1; ++j) { run += cost[j] * wc + wn; pre[j] = run; }
    part[t] = run;
    __syncthreads();
    if (t == 0) { float a = 0.f; for (int i = 0; i < 1024; ++i) { const float v = part[i]; part[i] = a; a += v; } }
    __syncthreads();
    for (int j = j0; j < j1; ++j) pre[j] += part[t];
    __syncthreads();
    // region of position j = floor(8 x exclusive prefix): monotone along the curve, so region r is the run [start[r], start[r + 1])
    auto region = [&](int j) { return min(7, (int)(8.0f * (j ? pre[j - 1] : 0.f))); };
    if (t < 9) {
        int lo = 0, hi = nsuper;                                   // first j with region(j) >= t
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (region(mid) >= t) hi = mid; else lo = mid + 1; }
        start[t] = t == 8 ? nsuper : lo;
    }
    __syncthreads();
    for (int j = j0; j < j1; ++j) {
        const int r = region(j), a = start[r], b = start[r + 1];
        const float c = cost[j];
        int rk = 0;
        for (int i = a; i < b; ++i) rk += (cost[i] > c || (cost[i] == c && i < j)) ? 1 : 0;
        if (rk < cap) order[8 * rk + r] = curve[j];
    }
}

#endif  // VPFX_AB (k_tile_regions)

// PARTIAL = false: the reference's single render target.  PARTIAL = true: OVER-phase and UNDER-phase MVs of the
// owned slab composite into two separate images (multi-GPU partial images).
// FLAGS = false compiles the vp_raymarch_params.flags paths (UNORM8 emulation, debug views) out of the hot loop.
// One wave per workgroup (no LDS, no barriers: a finished wave frees its slot at once), five waves per SIMD.
#ifndef VPFX_RM_WAVES
#define VPFX_RM_WAVES 4      // 115 VGPRs, no scratch (5 waves = 96 VGPRs spills 13 registers: measured 1.63 vs 1.58 ms at C3)
#endif
#ifndef VPFX_RM_WAVES_PARTIAL
#define VPFX_RM_WAVES_PARTIAL 3   // partial images + flag paths (debug views of a slab): the one combination that needs > 128 VGPRs
#endif
#ifndef VPFX_RM_WAVES_GREY
#define VPFX_RM_WAVES_GREY 4      // round 4: four samples per iteration (VPFX_RM_SPI) at 4 waves/SIMD; rounds 2-3: two samples, 95 VGPRs, 5 waves/SIMD (1.00 vs 1.09 ms at C3 then)
#endif
// (RmHandoff: vpfx_internal.h)
#define VPFX_RM_HANDOFF_CUTOFF 2.98023224e-8f          // 2^-25

template <int NV, bool PARTIAL, bool WRAP, bool FLAGS, bool GREY>
__global__ void __launch_bounds__(64, (PARTIAL && FLAGS) ? VPFX_RM_WAVES_PARTIAL : (GREY && !FLAGS) ? VPFX_RM_WAVES_GREY : VPFX_RM_WAVES)
k_raymarch(RmConsts k, const int* __restrict__ brick_index, const uint2* __restrict__ bricks, const float4* __restrict__ mvtrans,
           const int* __restrict__ rank, const float* __restrict__ scene_depth, float4* __restrict__ img_over, float4* __restrict__ img_under,
           unsigned long long* __restrict__ samples, int* __restrict__ brick_hit, const int* __restrict__ tile_order, int early_out,
           RmHandoff ho, const float4* __restrict__ cellinfo, const uint32_t* __restrict__ occmask, int order_len)
{
    const int lane = threadIdx.x;
    // XCD-aware tile order: workgroup b lands on XCD b % 8 (observed dispatch order; used for speed only), and each XCD
    // has its own L2.  Screen tiles are grouped into super-tiles (64x32 px, about one brick's footprint); a super-tile is
    // rendered entirely by one XCD, so a brick is pulled into ~2-4 L2s instead of all eight, while consecutive super-tiles
    // of the dispatch order (cost-sorted, see tile_cost) alternate XCDs (load balance).
    constexpr int LX = VPFX_RM_LX, LY = VPFX_RM_LY, WPS = 4 << (LX + LY);            // waves per super-tile
    constexpr int SW = 16 << LX, SH = 16 << LY;                                     // super-tile size in pixels
    const int sgx = (k.W + SW - 1) / SW, sgy = (k.H + SH - 1) / SH;
    const int q = (int)(blockIdx.x >> 3);
    const int within = q % WPS;                                                    // wave of the super-tile
    const int slot = (q / WPS) * 8 + (int)(blockIdx.x & 7u);                       // position in the dispatch order
    if (slot >= (tile_order ? order_len : sgx * sgy)) return;
    const int sti = tile_order ? tile_order[slot] : slot;                          // super-tile index
    if (sti < 0) return;                                                           // (padding of an XCD's share of the dispatch order)
    // The wave's pixel block: 2^wave_lx pixels along the lane-fastest screen axis x 64 / 2^wave_lx along the other (8 x 8, 16 x 4 or 32 x 2;
    // RmConsts.wave_lx, hl_build_rm_consts).  A brick row is a run of 128-byte lines along grid x, and at one lattice index the rows of a
    // pixel block sit on different (y, z) brick rows, so the lines a wave-sample touches ~ (rows of the block) x (lines per row): an
    // elongated block touches fewer, at the price of a less compact ray bundle.
    const int bwl = k.lane_transpose ? 6 - k.wave_lx : k.wave_lx, bhl = 6 - bwl;    // log2 of the block's width and height in pixels
    const int wx = within & ((SW >> bwl) - 1), wy = within >> (LX + 4 - bwl);
#if VPFX_RM_OCC_LDS
    // The cell walk crosses about two empty cells for every occupied one, and learning that a cell is empty used to cost a dependent global
    // load (31 % of the wave time sat in the walk, scripts/raymarch_phase_profile.py).  The grid's occupancy is one bit per cell: 4 KB at
    // 32^3, copied into LDS by the wave (one round trip for the whole copy), so that only occupied cells go to memory for their record.
    __shared__ uint32_t s_occ[VPFX_RM_OCC_WORDS];
    if (!FLAGS && k.occ_lds) {
        const int nw = k.Nz * k.Ny;
        for (int i = lane; i < nw; i += 64) s_occ[i] = occmask[i];
        __syncthreads();
    }
#endif
    // wave = 8 x 8 px.  (16 x 4 px touches fewer brick rows per load -- the L1 tag rate is what binds this kernel -- but loses traversal
    // coherence: 1.50 ms against 1.47; 4 x 16 px: 1.57.)
    // lanes row-major in the tile: the L1 serves a wave-load one lane quad per cycle when the quad's addresses share a 128-B line
    // (scripts/probes/l1_gather_probe.hip), and four pixels in a screen row share a brick row more often than a 2 x 2 px quad does
    // (Z-order lanes: 1.62 ms against 1.46).
    const int lx = k.lane_transpose ? lane >> bhl : lane & ((1 << bwl) - 1), ly = k.lane_transpose ? lane & ((1 << bhl) - 1) : lane >> bwl;   // (hl_build_rm_consts)
    const int col = (sti % sgx) * SW + (wx << bwl) + lx;
    const int row = (sti / sgx) * SH + (wy << bhl) + ly;
    if (col >= k.W || row >= k.H) return;

#if VPFX_RM_PROBE == 9
    unsigned long long rm_prof[8] = {};
    const unsigned long long rm_t0 = __builtin_amdgcn_s_memtime();
    unsigned long long rm_last = rm_t0;
#endif
    const RayCtx R = ray_setup(k, col, row, scene_depth);
    VPFX_RM_TICK(0);                                           // ray set-up

    // ONE accumulator: a slab kernel (PARTIAL) composites its phase-A slabs, stores that image when the first phase-B slab comes up and
    // starts over (two live images cost 4 registers that the 5-waves/SIMD budget does not have)
    F4 dst{0.f, 0.f, 0.f, 0.f};                                                            // OnPreRender clear  VPR.cs:171
    bool storedA = false;
    const size_t pi = (size_t)row * k.W + col;
    int nsamp = 0;

    // ray vs. the owned part of the grid
    float tg0, tg1;
    {
        const float bx0 = R.ivx * (0.f - R.ogx), bx1 = R.ivx * ((float)k.Nx - R.ogx);
        const float by0 = R.ivy * (0.f - R.ogy), by1 = R.ivy * ((float)k.Ny - R.ogy);
        const float bz0 = R.ivz * ((float)k.z0 - R.ogz), bz1 = R.ivz * ((float)k.z1 - R.ogz);
        tg0 = fmaxf(fmaxf(fminf(bx0, bx1), fminf(by0, by1)), fminf(bz0, bz1));
        tg1 = fminf(fminf(fmaxf(bx0, bx1), fmaxf(by0, by1)), fmaxf(bz0, bz1));
        if (FLAGS && (k.flags & (VP_RM_SHOW_NUM_SAMPLES | VP_RM_SHOW_BLEND_FUNC | VP_RM_SHOW_DRAW_ORDER))) {
            // The debug views colour every RASTERISED fragment (RM.shader:170-181 return before the box test; :283-299 also colours
            // a fragment with zero samples), i.e. every metavoxel whose exit point lies in front of the camera -- including those
            // the lattice clamp below never samples (all of them when the camera is outside the z-slab, quirk Q13: tCamera is an
            // unsigned distance).  So the walk starts at the camera itself (signed parameter along the ray), not at tCamera.
            const float tcs = ((k.camg[0] - R.ogx) * R.dgx + (k.camg[1] - R.ogy) * R.dgy) + (k.camg[2] - R.ogz) * R.dgz;
            tg0 = fmaxf(tg0, tcs);
        } else {
            tg0 = fmaxf(tg0, ((float)R.tCameraG - 2.0f) * k.mvStep);                       // nothing is sampled behind the camera
        }
    }
    const int nxy = k.Nx * k.Ny;
    bool done = !(tg0 <= tg1);
    float tin = 1.0f, aA = 0.f;
    const float cutoff = (PARTIAL && ho.t_in) ? VPFX_RM_HANDOFF_CUTOFF : k.alpha_cutoff;
    if (PARTIAL && ho.t_in) {
        int code = 0;                                          // product of the maps = sum of their codes (RmHandoff: t = 2^(-code / 8))
        for (int j = 0; j < ho.n_in; ++j) code += ho.t_in[(size_t)j * ho.plane + pi];
        tin = __builtin_amdgcn_exp2f(-0.125f * (float)code);
        if (early_out && tin <= cutoff) done = true;           // hidden by the slabs in front before this one starts
    }

    // Slab order.  The reference draws phase A (zz <= zBoundary) zz ascending, cells far -> near, blended OVER, then phase B
    // (zz > zBoundary) zz ascending, cells near -> far, blended UNDER (VPR.cs:652-711): front to back that is the REVERSE of
    // phase A followed by phase B.  Premultiplied over/under are the same associative operator seen from the two ends, so the
    // default kernel composites everything front to back with UNDER -- equal up to rounding (~1e-7) -- which lets a ray stop
    // as soon as it is saturated in EITHER phase (a top-down camera is all phase A).  The FLAGS kernel keeps the reference's
    // literal sequence: the UNORM8 render-target emulation re-quantises after every blend, so there the order is the result.
    const int nslab = k.z1 - k.z0;
    const int nA = min(max(k.zB - k.z0 + 1, 0), nslab);                                    // owned phase-A slabs
    for (int it = 0; it < nslab && !done; ++it) {
        const int zz = (!FLAGS && it < nA) ? k.z0 + nA - 1 - it : k.z0 + it;
        // parameter range of the ray inside slab zz
        float ta, tb;
        if (R.dgz != 0.f) {
            const float a = R.ivz * ((float)zz - R.ogz), b = R.ivz * ((float)(zz + 1) - R.ogz);
            ta = fmaxf(fminf(a, b), tg0); tb = fminf(fmaxf(a, b), tg1);
        } else {
            if ((int)floorf(R.ogz) != zz) continue;
            ta = tg0; tb = tg1;
        }
        if (!(ta <= tb)) continue;
        const bool phaseA = zz <= k.zB;                                                    // VPR.cs:667 vs :697
        const bool over = FLAGS && phaseA;                                                 // literal OVER, cells far -> near
        if (PARTIAL && !phaseA && !storedA) {
            img_over[pi] = make_float4(dst.x, dst.y, dst.z, dst.w);
            aA = dst.w;
            tin *= 1.0f - dst.w;                               // the slab's own phase-A image hides its phase-B image too
            dst = F4{0.f, 0.f, 0.f, 0.f};
            storedA = true;
        }
        F4 d = dst;
        const int* occ = brick_index + zz * nxy;
        int last = over ? 0x7fffffff : -1;
        // Walk of the (x,y) cells the ray crosses inside this slab, t in [ta, tb]: an integer DDA -- the cell index is stepped
        // across whichever cell face the ray reaches first and never re-derived from a position, so no cell the line crosses is
        // skipped, however short the crossing (a walk that re-locates itself at "t + epsilon" skips crossings shorter than its
        // epsilon, and with them the occasional lattice sample: measured 1 sample in 2e5).  Which metavoxel a sample belongs to
        // is decided by the reference's own box test inside march_mv; the walk only has to offer every candidate.
        // The only state of a walk is the integer cell (cx, cy): everything else is recomputed from it, so that little stays
        // live across the march of a metavoxel.
        auto walk_start = [&](int& cx, int& cy) { cx = (int)floorf(fmaf(ta, R.dgx, R.ogx)); cy = (int)floorf(fmaf(ta, R.dgy, R.ogy)); };
        // returns the current cell (-1: outside the grid) and steps to the next one (branch-free); fin is set with the last cell
        int ocx = 0, ocy = 0;                                   // the cell walk_step returned (before its step)
        auto walk_step = [&](int& cx, int& cy, bool& fin) -> int {
            // parameter at which the ray leaves cell column cx / cell row cy (from the integer index, never accumulated)
            const float tx = R.dgx != 0.f ? R.ivx * ((float)cx + ((R.dgx > 0.f ? 1.0f : 0.0f) - R.ogx)) : 3.0e38f;
            const float ty = R.dgy != 0.f ? R.ivy * ((float)cy + ((R.dgy > 0.f ? 1.0f : 0.0f) - R.ogy)) : 3.0e38f;
            const int cur = (cx >= 0 && cx < k.Nx && cy >= 0 && cy < k.Ny) ? cy * k.Nx + cx : -1;
            ocx = cx; ocy = cy;
            fin = !(fminf(tx, ty) < tb);
            const bool stepx = tx < ty;
            cx += stepx ? (R.dgx > 0.f ? 1 : -1) : 0;
            cy += stepx ? 0 : (R.dgy > 0.f ? 1 : -1);
            return cur;
        };
        const int max_cells = 2 * (k.Nx + k.Ny) + 8;
        // The reference's order inside a slab is the GLOBAL (x,y) sort (rank), not the order along this ray.  Almost always
        // the two agree (ranks ascend along the ray), so the cells are marched optimistically as ONE walk meets them -- O(cells)
        // instead of one selection walk per cell (O(cells^2): a ray running along a slab crosses up to Nx + Ny cells).  If an
        // occupied cell turns up whose rank is below the last one marched, the order was not the ray's: the slab's blends are
        // rolled back (dst and the sample count as they were at the slab's start) and the slab is redone with selection by rank,
        // which the literal OVER order always uses.  (A separate look-ahead walk checking the ranks first cost 3.6 % of the kernel.)
        bool stream = !over;
        const F4 d_start = d;
        const int ns_start = nsamp;
        int wcx, wcy, walked = 0;                               // streaming walk
        walk_start(wcx, wcy);
        bool wfin = false;
        for (;;) {
            int best_r = over ? -1 : 0x7fffffff, best_cell = -1;
#if VPFX_RM_CELLINFO
            float4 ci = make_float4(0.f, 0.f, 0.f, __int_as_float(-1));
#endif
            if (stream) {
                // next occupied cell along the ray
                while (!wfin && walked < max_cells) {
                    const int cell = walk_step(wcx, wcy, wfin);
                    ++walked;
#if VPFX_RM_CELLINFO
                    // one 16-byte record per cell = (translation of _CameraToMetavoxel, brick slot or -1): occupancy, slot and translation
                    // arrive with ONE load instead of three dependent ones (occupancy -> translation; the rank load runs beside it)
                    if (!FLAGS && cell >= 0) {
#if VPFX_RM_OCC_LDS
                        if (k.occ_lds && !((s_occ[zz * k.Ny + ocy] >> ocx) & 1u)) continue;      // empty: no trip to memory
#endif
                        const float4 q4 = cellinfo[zz * nxy + cell];
                        const int r = rank[cell];
                        if (__float_as_int(q4.w) >= 0) {
                            if (r > last) { best_r = r; best_cell = cell; ci = q4; }
                            else { d = d_start; nsamp = ns_start; stream = false; last = -1; }
                            break;
                        }
                        continue;
                    }
#endif
                    if (cell >= 0 && occ[cell] >= 0) {
                        const int r = rank[cell];
                        if (r > last) { best_r = r; best_cell = cell; }
                        else { d = d_start; nsamp = ns_start; stream = false; last = -1; }     // ranks do not ascend along this ray: redo
                        break;
                    }
                }
                if (!stream) continue;
            } else {
                // select the next occupied cell of this slab: rank ascending = near -> far (descending for the literal OVER order)
                int cx, cy;
                walk_start(cx, cy);
                bool fin = false;
                for (int guard = 0; guard < max_cells && !fin; ++guard) {
                    const int cell = walk_step(cx, cy, fin);
                    if (cell >= 0 && occ[cell] >= 0) {
                        const int r = rank[cell];
                        const bool better = over ? (r < last && r > best_r) : (r > last && r < best_r);
                        if (better) { best_r = r; best_cell = cell; }
                    }
                }
            }
            if (best_cell < 0) break;
            last = best_r;
#if VPFX_RM_CELLINFO
            const bool have_ci = !FLAGS && stream;
            const int bi = have_ci ? __float_as_int(ci.w) : occ[best_cell];
            const float4 trv = have_ci ? ci : mvtrans[bi];
#else
            const int bi = occ[best_cell];
            const float4 trv = mvtrans[bi];
#endif
            F4 src;
            const int ns0 = nsamp;
            const uint2* brick = bricks + (size_t)bi * (NV ? NV * NV * NV : k.nv * k.nv * k.nv);
            VPFX_RM_TICK(1);                                   // cell walk: the next occupied metavoxel along the ray
            if (!march_mv<NV, WRAP, FLAGS, GREY>(k, R, brick, trv, src, nsamp VPFX_RM_PROF_PASS)) { VPFX_RM_TICK(2); continue; }
            if (nsamp != ns0) brick_hit[bi] = 1;
            if (FLAGS && (k.flags & VP_RM_SHOW_BLEND_FUNC)) // debug view: yellow = OVER, cyan = UNDER   RM.shader:174-181
                src = phaseA ? F4{0.5f, 0.5f, 0.f, 1.f} : F4{0.f, 0.5f, 0.5f, 1.f};
            if (FLAGS && (k.flags & VP_RM_SHOW_DRAW_ORDER)) // debug view: position in the global submission order      RM.shader:170-173
                src = draw_order_color(__float_as_int(mvtrans[bi].w), k.num_covered);
            if (over) {                         // Blend One OneMinusSrcAlpha                                  VPR.cs:659-662
                const float ia = 1.0f - src.w;
                d.x = src.x + d.x * ia; d.y = src.y + d.y * ia; d.z = src.z + d.z * ia; d.w = src.w + d.w * ia;
            } else {                            // Blend OneMinusDstAlpha One                                  VPR.cs:688-691
                const float ia = 1.0f - d.w;
                d.x = src.x * ia + d.x; d.y = src.y * ia + d.y; d.z = src.z * ia + d.z; d.w = src.w * ia + d.w;
            }
            if (FLAGS && (k.flags & VP_RM_QUANTIZE_UNORM8)) {   // particlesRT is ARGB32: the ROP stores UNORM8 (Q19)    VPR.cs:228
                d.x = unorm8(d.x); d.y = unorm8(d.y); d.z = unorm8(d.z); d.w = unorm8(d.w);
            }
            VPFX_RM_TICK(4);                                   // inter-metavoxel blend
        }
        VPFX_RM_TICK(1);                                       // (the walk that found no further cell in this slab)
        // saturated: everything farther along the ray is multiplied by (1 - dst.a) == 0.  (A saturated phase-A image of a slab
        // also hides the slab's own phase-B image, which is composited behind it.)
        dst = d;
        if (PARTIAL) {
            if (!over && early_out && (1.0f - d.w) * tin <= cutoff) done = true;
            if (ho.zsamples && nsamp != ns_start)               // uniform address: hipcc reduces over the wave, one atomic per wave and slice
                atomicAdd(ho.zsamples + (blockIdx.x & (VPFX_ZPROF_COPIES - 1)) * k.Nz + zz, (unsigned)(nsamp - ns_start));
        } else if (!over && early_out && 1.0f - d.w <= k.alpha_cutoff) done = true;
    }

    if (FLAGS && (k.flags & VP_RM_SHOW_RAY_SAMPLES)) dst = F4{(float)nsamp, (float)nsamp, (float)nsamp, 1.0f};   // perf view: samples per ray
    if (PARTIAL && storedA) {
        img_under[pi] = make_float4(dst.x, dst.y, dst.z, dst.w);
    } else {
        img_over[pi] = make_float4(dst.x, dst.y, dst.z, dst.w);
        if (PARTIAL) img_under[pi] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (PARTIAL && ho.t_out0) {
        // code = floor(-8 log2 t), capped at 255: the decoded 2^(-code / 8) is never below t (a conservative bound: it can only make a
        // ray behind march a little longer, never stop it early), within a factor 2^(1/8) of it down to 2^-31.9
        auto encode = [](float t) { return (uint8_t)(int)fminf(-8.0f * __builtin_amdgcn_logf(t), 255.0f); };
        const float t0 = storedA ? 1.0f - aA : 1.0f - dst.w;
        ho.t_out0[pi] = encode(t0);
        if (ho.t_out1) ho.t_out1[pi] = encode(storedA ? t0 * (1.0f - dst.w) : t0);
    }
    if (nsamp) atomicAdd(samples, (unsigned long long)nsamp);
#if VPFX_RM_PROBE == 9
    VPFX_RM_TICK(5);                                           // image / hand-off stores
    rm_prof[7] = __builtin_amdgcn_s_memtime() - rm_t0;
    if (lane == (int)__builtin_amdgcn_readfirstlane(__builtin_ctzll(__builtin_amdgcn_ballot_w64(true))))
        for (int i = 0; i < 8; ++i) atomicAdd(&g_rm_prof[blockIdx.x & 63][i], rm_prof[i]);
#endif
}

#if VPFX_AB   // measured-and-dropped A/B variant (2.2-3.9 x slower): only in `make EXTRA=-DVPFX_AB=1` builds, run with VPFX_RM_FLAT=1
// ---------------------------------------------------------------------------------------------------------------------------------
// k_raymarch_flat: the same ray-march with a WAVE-COHERENT traversal (whole-grid or slab contexts; border >= 1, no flag paths).
// k_raymarch nests "for every metavoxel the ray meets { for every sample in it }": the 64 rays of a wave enter, cross and leave a metavoxel
// at different lattice indices, and a lane that has finished ITS samples of the metavoxel idles until the slowest ray of the wave has
// finished (59 % of the lanes of a wave-sample hold a sample at the benchmark view; scripts/lane_bound.py: 93 % if a lane only ever
// idled once its whole ray is done).  Here every lane is a small state machine -- "looking for my next metavoxel" / "sampling it" -- and the
// wave alternates between two phases: lanes without a metavoxel advance their cell walk to the next occupied cell and set that metavoxel up
// (the reference's per-draw arithmetic, unchanged), then every lane that has one takes its next two lattice samples.  A lane whose
// metavoxel is exhausted blends it and rejoins at the next advance phase instead of waiting for its neighbours.
// Same arithmetic in the same order per ray as k_raymarch (the image is bit-identical, the executed samples are the oracle's).  The
// reference's global draw order is taken as the order along the ray (ranks ascend along it); a ray that meets a lower rank after a higher one
// (rare: k_raymarch rolls such a slab back) is marched afresh by the selection-by-rank loop at the end.
template <int NV, bool GREY>
struct FlatMv {                 // the metavoxel a lane is sampling
    const uint2* brick;
    float f0x, f0y, f0z;        // texel coordinate of lattice index 0 (affine: texel(si) = f0 + si * fs)
    float si, tEntry, tCamera;  // next lattice index (counting down, back to front), first index of the interval, camera index: exact integers
    float rr, rg, rb, trans;    // the reference's back-to-front accumulators of this metavoxel (RM.shader:244-275)
    int bi;
};

template <int NV, bool PARTIAL, bool GREY>
__global__ void __launch_bounds__(64, VPFX_RM_WAVES)
k_raymarch_flat(RmConsts k, const int* __restrict__ brick_index, const uint2* __restrict__ bricks, const float4* __restrict__ mvtrans,
                const int* __restrict__ rank, const float* __restrict__ scene_depth, float4* __restrict__ img_over, float4* __restrict__ img_under,
                unsigned long long* __restrict__ samples, int* __restrict__ brick_hit, const int* __restrict__ tile_order, int early_out,
                RmHandoff ho)
{
    const int lane = threadIdx.x;
    constexpr int LX = VPFX_RM_LX, LY = VPFX_RM_LY, WPS = 4 << (LX + LY);
    const int tgx = (k.W + 15) >> 4, tgy = (k.H + 15) >> 4;
    const int sgx = (tgx + (1 << LX) - 1) >> LX, sgy = (tgy + (1 << LY) - 1) >> LY;
    const int q = (int)(blockIdx.x >> 3);
    const int within = q % WPS, wave = within & 3, j = within >> 2;
    const int slot = (q / WPS) * 8 + (int)(blockIdx.x & 7u);
    if (slot >= sgx * sgy) return;
    const int sti = tile_order ? tile_order[slot] : slot;
    const int ttx = ((sti % sgx) << LX) + (j & ((1 << LX) - 1)), tty = ((sti / sgx) << LY) + (j >> LX);
    if (ttx >= tgx || tty >= tgy) return;
    const int lx = k.lane_transpose ? lane >> 3 : lane & 7, ly = k.lane_transpose ? lane & 7 : lane >> 3;   // (hl_build_rm_consts)
    const int col = ttx * 16 + (wave & 1) * 8 + lx;
    const int row = tty * 16 + (wave >> 1) * 8 + ly;
    if (col >= k.W || row >= k.H) return;

    const RayCtx R = ray_setup(k, col, row, scene_depth);
    const size_t pi = (size_t)row * k.W + col;
    F4 dst{0.f, 0.f, 0.f, 0.f};
    bool storedA = false;
    int nsamp = 0;
    float tg0, tg1;
    {
        const float bx0 = R.ivx * (0.f - R.ogx), bx1 = R.ivx * ((float)k.Nx - R.ogx);
        const float by0 = R.ivy * (0.f - R.ogy), by1 = R.ivy * ((float)k.Ny - R.ogy);
        const float bz0 = R.ivz * ((float)k.z0 - R.ogz), bz1 = R.ivz * ((float)k.z1 - R.ogz);
        tg0 = fmaxf(fmaxf(fminf(bx0, bx1), fminf(by0, by1)), fminf(bz0, bz1));
        tg1 = fminf(fminf(fmaxf(bx0, bx1), fmaxf(by0, by1)), fmaxf(bz0, bz1));
        tg0 = fmaxf(tg0, ((float)R.tCameraG - 2.0f) * k.mvStep);
    }
    const int nxy = k.Nx * k.Ny;
    float tin = 1.0f, aA = 0.f;
    const float cutoff = (PARTIAL && ho.t_in) ? VPFX_RM_HANDOFF_CUTOFF : k.alpha_cutoff;
    bool alive = tg0 <= tg1;                       // still has metavoxels to look for or to sample
    if (PARTIAL && ho.t_in) {
        int code = 0;
        for (int jj = 0; jj < ho.n_in; ++jj) code += ho.t_in[(size_t)jj * ho.plane + pi];
        tin = __builtin_amdgcn_exp2f(-0.125f * (float)code);
        if (early_out && tin <= cutoff) alive = false;
    }
    const int nslab = k.z1 - k.z0;
    const int nA = min(max(k.zB - k.z0 + 1, 0), nslab);
    const int max_cells = 2 * (k.Nx + k.Ny) + 8;
    // per-ray constants of the sampling (march_mv): texel step per lattice index
    const float fsx = (k.mvStep * R.dx) * k.texScale, fsy = (k.mvStep * R.dy) * k.texScale, fsz = (k.mvStep * R.dz) * k.texScale;

    // lane state
    int it = -1, zz = 0, wcx = 0, wcy = 0, walked = 0, last = -1, ns_layer = 0;
    float ta = 0.f, tb = 0.f;
    bool in_layer = false, wfin = false, in_mv = false, slow = false;
    FlatMv<NV, GREY> m{};

    auto walk_step = [&](int& cx, int& cy, bool& fin) -> int {
        const float tx = R.dgx != 0.f ? R.ivx * ((float)cx + ((R.dgx > 0.f ? 1.0f : 0.0f) - R.ogx)) : 3.0e38f;
        const float ty = R.dgy != 0.f ? R.ivy * ((float)cy + ((R.dgy > 0.f ? 1.0f : 0.0f) - R.ogy)) : 3.0e38f;
        const int cur = (cx >= 0 && cx < k.Nx && cy >= 0 && cy < k.Ny) ? cy * k.Nx + cx : -1;
        fin = !(fminf(tx, ty) < tb);
        const bool stepx = tx < ty;
        cx += stepx ? (R.dgx > 0.f ? 1 : -1) : 0;
        cy += stepx ? 0 : (R.dgy > 0.f ? 1 : -1);
        return cur;
    };
    // the slab's layer is finished: profile, early-out (the single-GPU rule, or the hand-off's bound)
    auto end_layer = [&]() {
        in_layer = false;
        if (PARTIAL && ho.zsamples && nsamp != ns_layer)
            atomicAdd(ho.zsamples + (blockIdx.x & (VPFX_ZPROF_COPIES - 1)) * k.Nz + zz, (unsigned)(nsamp - ns_layer));
        if (early_out && (PARTIAL ? (1.0f - dst.w) * tin : 1.0f - dst.w) <= cutoff) alive = false;
    };

    for (;;) {
        // ---- advance: lanes without a metavoxel look for their next one ---------------------------------------------------------
        // Finding and setting up a metavoxel is a chain of dependent loads (cell occupancy -> rank, translation): ~2 000 cycles in which the
        // whole wave stalls.  Run per lane the moment it becomes idle, that chain ran every other iteration (3.85 ms at C3 against
        // 0.97 for the nested kernel).  So idle lanes WAIT until VPFX_FLAT_BATCH of them can advance together -- or nobody can sample.
#ifndef VPFX_FLAT_BATCH
#define VPFX_FLAT_BATCH 16
#endif
        const unsigned long long want = __builtin_amdgcn_ballot_w64(alive && !in_mv);
        const bool go = __builtin_popcountll(want) >= VPFX_FLAT_BATCH || !__builtin_amdgcn_ballot_w64(in_mv);
        if (go && alive && !in_mv) {
            for (;;) {
                if (!in_layer) {
                    if (++it >= nslab) { alive = false; break; }
                    zz = it < nA ? k.z0 + nA - 1 - it : k.z0 + it;
                    if (R.dgz != 0.f) {
                        const float a = R.ivz * ((float)zz - R.ogz), b = R.ivz * ((float)(zz + 1) - R.ogz);
                        ta = fmaxf(fminf(a, b), tg0); tb = fminf(fmaxf(a, b), tg1);
                    } else {
                        if ((int)floorf(R.ogz) != zz) continue;
                        ta = tg0; tb = tg1;
                    }
                    if (!(ta <= tb)) continue;
                    if (PARTIAL && !(zz <= k.zB) && !storedA) {          // first phase-B slice of a slab: the phase-A image is complete
                        img_over[pi] = make_float4(dst.x, dst.y, dst.z, dst.w);
                        aA = dst.w;
                        tin *= 1.0f - dst.w;
                        dst = F4{0.f, 0.f, 0.f, 0.f};
                        storedA = true;
                    }
                    wcx = (int)floorf(fmaf(ta, R.dgx, R.ogx)); wcy = (int)floorf(fmaf(ta, R.dgy, R.ogy));
                    wfin = false; walked = 0; last = -1; in_layer = true; ns_layer = nsamp;
                }
                // next occupied cell along the ray inside this slice
                int cell = -1;
                const int* occ = brick_index + zz * nxy;
                while (!wfin && walked < max_cells) {
                    const int c = walk_step(wcx, wcy, wfin);
                    ++walked;
                    if (c >= 0 && occ[c] >= 0) { cell = c; break; }
                }
                if (cell < 0) { end_layer(); if (!alive) break; continue; }
                const int r = rank[cell];
                if (r <= last) { slow = true; alive = false; break; }    // the draw order is not the order along this ray: redone below
                last = r;
                // ---- set the metavoxel up: RM.shader frag (166-240), arithmetic as in march_mv ---------------------------------------
                const int bi = occ[cell];
                const float4 tr = mvtrans[bi];
                const float ox = R.lx + tr.x, oy = R.ly + tr.y, oz = R.lz + tr.z;
                const float tbx = R.idx * (-0.5f - ox), tby = R.idy * (-0.5f - oy), tbz = R.idz * (-0.5f - oz);
                const float ttx2 = R.idx * (0.5f - ox), tty2 = R.idy * (0.5f - oy), ttz2 = R.idz * (0.5f - oz);
                const float tminx = fminf(ttx2, tbx), tminy = fminf(tty2, tby), tminz = fminf(ttz2, tbz);
                const float tmaxx = fmaxf(ttx2, tbx), tmaxy = fmaxf(tty2, tby), tmaxz = fmaxf(ttz2, tbz);
                const float t1 = fmaxf(fmaxf(tminx, tminy), fmaxf(tminx, tminz));
                const float t2 = fminf(fminf(tmaxx, tmaxy), fminf(tmaxx, tmaxz));
                if (t1 > t2) continue;
                const float exitDepth = -(R.startz + R.dirz * (t2 * k.s));
                if (!(exitDepth > k.nearc) || !(exitDepth <= k.farc) || !(exitDepth < R.sceneDepth)) continue;
                int tEntry = (int)ceilf(t1 / k.mvStep);
                const int tExit = (int)floorf(t2 / k.mvStep);
                const float cx = tr.x - ox, cy = tr.y - oy, cz = tr.z - oz;
                const int tCamera = (int)(sqrtf((cx * cx + cy * cy) + cz * cz) / k.mvStep);
                tEntry = max(tEntry, tCamera);
                // a fragment with an empty interval still blends (0, 0, 0, 0): a no-op under UNDER, skipped
                if (tExit < tEntry) continue;
                nsamp += tExit - tEntry + 1;
                brick_hit[bi] = 1;
                m.brick = bricks + (size_t)bi * NV * NV * NV;
                m.f0x = fmaf(ox + 0.5f, k.texScale, k.texBias); m.f0y = fmaf(oy + 0.5f, k.texScale, k.texBias); m.f0z = fmaf(oz + 0.5f, k.texScale, k.texBias);
                m.si = (float)tExit; m.tEntry = (float)tEntry; m.tCamera = (float)tCamera;
                m.rr = m.rg = m.rb = 0.f; m.trans = 1.0f; m.bi = bi;
                in_mv = true;
                break;
            }
        }
        if (!__builtin_amdgcn_ballot_w64(alive)) break;                  // every lane of the wave is finished (or waits for the slow path)
        // ---- sample: every lane that has a metavoxel takes its next two lattice samples (back to front) ------------------------------
        if (in_mv) {
            struct Addr { const uint2* p; float wx, wy, wz; };
            auto address = [&](float fi) -> Addr {
                const float fx = fmaf(fi, fsx, m.f0x), fy = fmaf(fi, fsy, m.f0y), fz = fmaf(fi, fsz, m.f0z);
                const float x0 = floorf(fx), y0 = floorf(fy), z0 = floorf(fz);
                Addr a;
                a.wx = fx - x0; a.wy = fy - y0; a.wz = fz - z0;
                a.p = m.brick + (int)fmaf(fmaf(z0, (float)NV, y0), (float)NV, x0);
                return a;
            };
            auto blend = [&](float cr, float cg, float cb, float density) {
                const float bf = __builtin_amdgcn_rcpf(1.0f + density);
                m.rr = fmaf(bf, m.rr - cr, cr);
                if (!GREY) { m.rg = fmaf(bf, m.rg - cg, cg); m.rb = fmaf(bf, m.rb - cb, cb); }
                m.trans *= bf;
            };
            // soft particles (RM.shader:267-270) fade the `soft` lattice points nearest the camera; whether a lane is there is per lane,
            // whether the wave has to look is wave-uniform (never, for a camera outside the volume)
            const bool two = m.si - 1.0f >= m.tEntry;
            const bool in_soft = (m.si - (two ? 1.0f : 0.0f)) - m.tCamera < (float)k.soft;
            const bool any_soft = __builtin_amdgcn_ballot_w64(in_soft) != 0;
            const Addr a0 = address(m.si), a1 = address(two ? m.si - 1.0f : m.si);
            auto fade = [&](float den, float fi) { const float dc = fi - m.tCamera; return dc < (float)k.soft ? den * (dc * k.inv_soft) : den; };
            if (GREY) {
                u32x4 u0, u1, v0, v1;
                issue_load16<0>(u0, a0.p); issue_load16<NV * 8>(u1, a0.p);
                issue_load16<0>(v0, a1.p); issue_load16<NV * 8>(v1, a1.p);
                wait_pair<2>(u0, u1);
                {
                    const float ay = 1.0f - a0.wy, az = 1.0f - a0.wz;
                    const float w00 = ay * az, w10 = a0.wy * az, w01 = ay * a0.wz, w11 = a0.wy * a0.wz;
                    const float l0 = mix_fma_lo(w11, u1[1], mix_fma_lo(w01, u0[1], mix_fma_lo(w10, u1[0], mix_fma_lo(w00, u0[0], 0.f))));
                    const float l1 = mix_fma_lo(w11, u1[3], mix_fma_lo(w01, u0[3], mix_fma_lo(w10, u1[2], mix_fma_lo(w00, u0[2], 0.f))));
                    const float d0 = mix_fma_hi(w11, u1[1], mix_fma_hi(w01, u0[1], mix_fma_hi(w10, u1[0], mix_fma_hi(w00, u0[0], 0.f))));
                    const float d1 = mix_fma_hi(w11, u1[3], mix_fma_hi(w01, u0[3], mix_fma_hi(w10, u1[2], mix_fma_hi(w00, u0[2], 0.f))));
                    const float lum = lerpf(l0, l1, a0.wx), den = lerpf(d0, d1, a0.wx);
                    blend(lum, lum, lum, any_soft ? fade(den, m.si) : den);
                }
                wait_pair<0>(v0, v1);
                if (two) {
                    const float ay = 1.0f - a1.wy, az = 1.0f - a1.wz;
                    const float w00 = ay * az, w10 = a1.wy * az, w01 = ay * a1.wz, w11 = a1.wy * a1.wz;
                    const float l0 = mix_fma_lo(w11, v1[1], mix_fma_lo(w01, v0[1], mix_fma_lo(w10, v1[0], mix_fma_lo(w00, v0[0], 0.f))));
                    const float l1 = mix_fma_lo(w11, v1[3], mix_fma_lo(w01, v0[3], mix_fma_lo(w10, v1[2], mix_fma_lo(w00, v0[2], 0.f))));
                    const float d0 = mix_fma_hi(w11, v1[1], mix_fma_hi(w01, v0[1], mix_fma_hi(w10, v1[0], mix_fma_hi(w00, v0[0], 0.f))));
                    const float d1 = mix_fma_hi(w11, v1[3], mix_fma_hi(w01, v0[3], mix_fma_hi(w10, v1[2], mix_fma_hi(w00, v0[2], 0.f))));
                    const float lum = lerpf(l0, l1, a1.wx), den = lerpf(d0, d1, a1.wx);
                    blend(lum, lum, lum, any_soft ? fade(den, m.si - 1.0f) : den);
                }
            } else {
                u32x4 u0, u1, u2, u3, v0, v1, v2, v3;
                const uint2* z0p = a0.p + NV * NV; const uint2* z1p = a1.p + NV * NV;
                issue_load16<0>(u0, a0.p); issue_load16<NV * 8>(u1, a0.p); issue_load16<0>(u2, z0p); issue_load16<NV * 8>(u3, z0p);
                issue_load16<0>(v0, a1.p); issue_load16<NV * 8>(v1, a1.p); issue_load16<0>(v2, z1p); issue_load16<NV * 8>(v3, z1p);
                auto filt = [&](const u32x4& t00, const u32x4& t10, const u32x4& t01, const u32x4& t11, const Addr& a, float fi) {
                    const float ay = 1.0f - a.wy, az = 1.0f - a.wz;
                    const float w00 = ay * az, w10 = a.wy * az, w01 = ay * a.wz, w11 = a.wy * a.wz;
                    const float r0 = mix_fma_lo(w11, t11[0], mix_fma_lo(w01, t01[0], mix_fma_lo(w10, t10[0], mix_fma_lo(w00, t00[0], 0.f))));
                    const float r1 = mix_fma_lo(w11, t11[2], mix_fma_lo(w01, t01[2], mix_fma_lo(w10, t10[2], mix_fma_lo(w00, t00[2], 0.f))));
                    const float g0 = mix_fma_hi(w11, t11[0], mix_fma_hi(w01, t01[0], mix_fma_hi(w10, t10[0], mix_fma_hi(w00, t00[0], 0.f))));
                    const float g1 = mix_fma_hi(w11, t11[2], mix_fma_hi(w01, t01[2], mix_fma_hi(w10, t10[2], mix_fma_hi(w00, t00[2], 0.f))));
                    const float b0 = mix_fma_lo(w11, t11[1], mix_fma_lo(w01, t01[1], mix_fma_lo(w10, t10[1], mix_fma_lo(w00, t00[1], 0.f))));
                    const float b1 = mix_fma_lo(w11, t11[3], mix_fma_lo(w01, t01[3], mix_fma_lo(w10, t10[3], mix_fma_lo(w00, t00[3], 0.f))));
                    const float q0 = mix_fma_hi(w11, t11[1], mix_fma_hi(w01, t01[1], mix_fma_hi(w10, t10[1], mix_fma_hi(w00, t00[1], 0.f))));
                    const float q1 = mix_fma_hi(w11, t11[3], mix_fma_hi(w01, t01[3], mix_fma_hi(w10, t10[3], mix_fma_hi(w00, t00[3], 0.f))));
                    const float den = lerpf(q0, q1, a.wx);
                    blend(lerpf(r0, r1, a.wx), lerpf(g0, g1, a.wx), lerpf(b0, b1, a.wx), any_soft ? fade(den, fi) : den);
                };
                wait_quad<4>(u0, u1, u2, u3);
                filt(u0, u1, u2, u3, a0, m.si);
                wait_quad<0>(v0, v1, v2, v3);
                if (two) filt(v0, v1, v2, v3, a1, m.si - 1.0f);
            }
            m.si -= 2.0f;
            if (m.si < m.tEntry) {
                // the metavoxel is done: src = (rgb, 1 - T) premultiplied (RM.shader:301), blended UNDER (VPR.cs:688-691)
                if (GREY) { m.rg = m.rr; m.rb = m.rr; }
                const float sa = 1.0f - m.trans, ia = 1.0f - dst.w;
                dst.x = m.rr * ia + dst.x; dst.y = m.rg * ia + dst.y; dst.z = m.rb * ia + dst.z; dst.w = sa * ia + dst.w;
                in_mv = false;
            }
        }
    }

    // ---- rays whose draw order is not their depth order: marched afresh, metavoxels selected by rank within every slice (the reference's
    //      literal order; O(cells^2) walks, rare) ---------------------------------------------------------------------------------------
    if (slow) {
        dst = F4{0.f, 0.f, 0.f, 0.f}; storedA = false; nsamp = 0; aA = 0.f;
        float tin2 = 1.0f;
        if (PARTIAL && ho.t_in) { int code = 0; for (int jj = 0; jj < ho.n_in; ++jj) code += ho.t_in[(size_t)jj * ho.plane + pi]; tin2 = __builtin_amdgcn_exp2f(-0.125f * (float)code); }
        bool done = false;
        for (int it2 = 0; it2 < nslab && !done; ++it2) {
            const int z2 = it2 < nA ? k.z0 + nA - 1 - it2 : k.z0 + it2;
            float sa, sb;
            if (R.dgz != 0.f) {
                const float a = R.ivz * ((float)z2 - R.ogz), b = R.ivz * ((float)(z2 + 1) - R.ogz);
                sa = fmaxf(fminf(a, b), tg0); sb = fminf(fmaxf(a, b), tg1);
            } else {
                if ((int)floorf(R.ogz) != z2) continue;
                sa = tg0; sb = tg1;
            }
            if (!(sa <= sb)) continue;
            if (PARTIAL && !(z2 <= k.zB) && !storedA) {
                img_over[pi] = make_float4(dst.x, dst.y, dst.z, dst.w);
                aA = dst.w; tin2 *= 1.0f - dst.w;
                dst = F4{0.f, 0.f, 0.f, 0.f}; storedA = true;
            }
            const int* occ = brick_index + z2 * nxy;
            const int ns0 = nsamp;
            int lastr = -1;
            tb = sb;                                           // walk_step reads the slice's end from tb
            for (;;) {
                int best_r = 0x7fffffff, best_cell = -1, cx = (int)floorf(fmaf(sa, R.dgx, R.ogx)), cy = (int)floorf(fmaf(sa, R.dgy, R.ogy));
                bool fin = false;
                for (int guard = 0; guard < max_cells && !fin; ++guard) {
                    const int c = walk_step(cx, cy, fin);
                    if (c >= 0 && occ[c] >= 0) { const int r = rank[c]; if (r > lastr && r < best_r) { best_r = r; best_cell = c; } }
                }
                if (best_cell < 0) break;
                lastr = best_r;
                const int bi = occ[best_cell];
                F4 src;
                const int nsb = nsamp;
                VPFX_RM_PROF_DUMMY
                if (!march_mv<NV, false, false, GREY>(k, R, bricks + (size_t)bi * NV * NV * NV, mvtrans[bi], src, nsamp VPFX_RM_PROF_PASS)) continue;
                if (nsamp != nsb) brick_hit[bi] = 1;
                const float ia = 1.0f - dst.w;
                dst.x = src.x * ia + dst.x; dst.y = src.y * ia + dst.y; dst.z = src.z * ia + dst.z; dst.w = src.w * ia + dst.w;
            }
            if (PARTIAL && ho.zsamples && nsamp != ns0)
                atomicAdd(ho.zsamples + (blockIdx.x & (VPFX_ZPROF_COPIES - 1)) * k.Nz + z2, (unsigned)(nsamp - ns0));
            if (early_out && (PARTIAL ? (1.0f - dst.w) * tin2 : 1.0f - dst.w) <= cutoff) done = true;
        }
    }

    if (PARTIAL && storedA) {
        img_under[pi] = make_float4(dst.x, dst.y, dst.z, dst.w);
    } else {
        img_over[pi] = make_float4(dst.x, dst.y, dst.z, dst.w);
        if (PARTIAL) img_under[pi] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (PARTIAL && ho.t_out0) {
        auto encode = [](float t) { return (uint8_t)(int)fminf(-8.0f * __builtin_amdgcn_logf(t), 255.0f); };
        const float t0 = storedA ? 1.0f - aA : 1.0f - dst.w;
        ho.t_out0[pi] = encode(t0);
        if (ho.t_out1) ho.t_out1[pi] = encode(storedA ? t0 * (1.0f - dst.w) : t0);
    }
    if (nsamp) atomicAdd(samples, (unsigned long long)nsamp);
}

#endif  // VPFX_AB (k_raymarch_flat)

// RenderMetavoxel(xx, yy, zz, orderIndex) (VPR.cs:766-794) as the reference submits it: ONE metavoxel, every pixel of the target,
// blended into particlesRT with the blend state RenderMetavoxels set (VPR.cs:659-662 OVER / 688-691 UNDER).  The per-metavoxel
// entry point of the C ABI (debugging, literal replays of the reference's draw loop); the frame path is k_raymarch.
template <int NV, bool WRAP, bool GREY>
__global__ void __launch_bounds__(256)
k_raymarch_one(RmConsts k, const uint2* __restrict__ brick, float4 tr, const float* __restrict__ scene_depth, float4* __restrict__ img,
               int blend_over, int order_index, unsigned long long* __restrict__ samples)
{
    const int col = blockIdx.x * 16 + (threadIdx.x & 15), row = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (col >= k.W || row >= k.H) return;
    const RayCtx R = ray_setup(k, col, row, scene_depth);
    F4 src;
    int nsamp = 0;
    VPFX_RM_PROF_DUMMY
    if (!march_mv<NV, WRAP, true, GREY>(k, R, brick, tr, src, nsamp VPFX_RM_PROF_PASS)) return;  // no fragment: the ROP is not touched
    if (k.flags & VP_RM_SHOW_BLEND_FUNC) src = blend_over ? F4{0.5f, 0.5f, 0.f, 1.f} : F4{0.f, 0.5f, 0.5f, 1.f};
    if (k.flags & VP_RM_SHOW_DRAW_ORDER) src = draw_order_color(order_index, k.num_covered);
    const size_t pi = (size_t)row * k.W + col;
    const float4 q = img[pi];
    F4 d{q.x, q.y, q.z, q.w};
    if (blend_over) {
        const float ia = 1.0f - src.w;
        d.x = src.x + d.x * ia; d.y = src.y + d.y * ia; d.z = src.z + d.z * ia; d.w = src.w + d.w * ia;
    } else {
        const float ia = 1.0f - d.w;
        d.x = src.x * ia + d.x; d.y = src.y * ia + d.y; d.z = src.z * ia + d.z; d.w = src.w * ia + d.w;
    }
    if (k.flags & VP_RM_QUANTIZE_UNORM8) { d.x = unorm8(d.x); d.y = unorm8(d.y); d.z = unorm8(d.z); d.w = unorm8(d.w); }
    img[pi] = make_float4(d.x, d.y, d.z, d.w);
    if (nsamp) atomicAdd(samples, (unsigned long long)nsamp);
}

// Cost-sorted dispatch order or raster order?  The order only matters when the launch does not fit the GPU at once: an image whose waves are
// all resident together (two per SIMD counted, the least any instantiation gets) ends with its slowest wave whatever the order, and the
// estimate + sort are 8 us of such a frame (C1: 1 024 waves).
bool rm_ordered(const vp_ctx* c, const RmConsts& k)
{
#ifdef VPFX_RM_NO_ORDER
    return false;
#else
    const int nsuper = rm_num_super_tiles(k.W, k.H);
    const long long waves = (long long)nsuper * (4 << (VPFX_RM_LX + VPFX_RM_LY));
    // A launch that runs at the memory side's pace (k.wave_lx == 4: the lattice is sparser than the texels, nothing a wave fetches is reused --
    // config 5) keeps RASTER order: neighbouring super-tiles then run at the same time and the lines their waves share (the halo of every
    // wave's footprint) are still in a cache when the second one asks, which is worth more than a short tail -- C5 7.63 -> 6.75 ms, every
    // memory-paced view of the camera sweep -8 ... -13 %; the cost order stays where the kernel is issue-bound (C3 0.944 against 1.014 in
    // raster order, C2 0.451 against 0.516).  Blocks of 2^n x 2^n super-tiles ranked by cost with raster order inside were worse than both
    // at C5 (7.50 / 7.65 / 8.29 / 8.05 ms for n = 1..4): profiles/r04_ab/raymarch_dispatch_order_memory_paced.{txt,patch}.
    return nsuper <= RM_ORDER_MAX && waves > (long long)c->num_cus * 8 && k.wave_lx != 4;
#endif
}

template <int NV, bool PARTIAL, bool WRAP, bool FLAGS, bool GREY = false>
void launch_rm_variant(vp_ctx* c, const RmConsts& k, float* d_over, float* d_under, int early_out, const RmHandoff& ho)
{
    const int nsuper = rm_num_super_tiles(k.W, k.H);
    const int* order = nullptr;
    int order_len = nsuper;
#ifndef VPFX_RM_NO_ORDER
    if (rm_ordered(c, k)) {
        float* cost = reinterpret_cast<float*>(c->d_tile_order + rm_order_ints(nsuper));     // written by k_rm_prepare's trailing workgroups
#if VPFX_AB
        if (c->rm_xcd_affine && c->d_tile_curve) {
            const int cap = rm_order_cap(nsuper);
            hipLaunchKernelGGL(k_tile_regions, dim3(1), dim3(1024), 0, c->stream, cost, c->d_tile_curve, nsuper, cap, c->d_tile_order);
            order_len = 8 * cap;
        } else
#endif
            hipLaunchKernelGGL(k_tile_rank, dim3((nsuper + 63) / 64), dim3(1024), 0, c->stream, cost, nsuper, c->d_tile_order);
        order = c->d_tile_order;
    }
#endif
    const dim3 grid(((order_len + 7) / 8) * 8 * (4 << (VPFX_RM_LX + VPFX_RM_LY))), block(64);
    hipLaunchKernelGGL((k_raymarch<NV, PARTIAL, WRAP, FLAGS, GREY>), grid, block, 0, c->stream, k, c->d_brick_index, c->d_bricks, c->d_mvtrans,
                       c->d_rank, c->d_scene_depth, (float4*)d_over, (float4*)d_under, c->d_samples, c->d_brick_hit, order, early_out, ho,
                       (const float4*)c->d_cellinfo, (const uint32_t*)c->d_occmask, order_len);
}

#if VPFX_AB
template <int NV, bool PARTIAL, bool GREY>
void launch_rm_flat(vp_ctx* c, const RmConsts& k, float* d_over, float* d_under, int early_out, const RmHandoff& ho)
{
    const int nsuper = rm_num_super_tiles(k.W, k.H);
    const int* order = nullptr;
    if (rm_ordered(c, k)) {
        float* cost = reinterpret_cast<float*>(c->d_tile_order + rm_order_ints(nsuper));
        hipLaunchKernelGGL(k_tile_rank, dim3((nsuper + 63) / 64), dim3(1024), 0, c->stream, cost, nsuper, c->d_tile_order);
        order = c->d_tile_order;
    }
    const dim3 grid(((nsuper + 7) / 8) * 8 * (4 << (VPFX_RM_LX + VPFX_RM_LY))), block(64);
    hipLaunchKernelGGL((k_raymarch_flat<NV, PARTIAL, GREY>), grid, block, 0, c->stream, k, c->d_brick_index, c->d_bricks, c->d_mvtrans,
                       c->d_rank, c->d_scene_depth, (float4*)d_over, (float4*)d_under, c->d_samples, c->d_brick_hit, order, early_out, ho);
}

#endif

template <int NV>
void launch_rm_nv(vp_ctx* c, const RmConsts& k, float* d_over, float* d_under, int early_out, const RmHandoff& ho)
{
    const bool wrap = c->g.b < 1;          // only a border-less brick can filter across its faces (wrap = Repeat)
#if VPFX_AB
    if (c->rm_flat && !wrap && !k.flags) { // the wave-coherent traversal (border >= 1, no debug views / UNORM8 emulation)
        if (c->bricks_grey) { if (d_under) launch_rm_flat<NV, true, true>(c, k, d_over, d_under, early_out, ho); else launch_rm_flat<NV, false, true>(c, k, d_over, d_under, early_out, ho); }
        else { if (d_under) launch_rm_flat<NV, true, false>(c, k, d_over, d_under, early_out, ho); else launch_rm_flat<NV, false, false>(c, k, d_over, d_under, early_out, ho); }
        return;
    }
#endif
    const int sel = (d_under ? 4 : 0) | (wrap ? 2 : 0) | (k.flags ? 1 : 0);
    if (c->bricks_grey) {                  // (luminance, density) bricks: only ever filled with border >= 1
        switch (sel & 5) {
        case 0: launch_rm_variant<NV, false, false, false, true>(c, k, d_over, d_under, early_out, ho); break;
        case 1: launch_rm_variant<NV, false, false, true, true>(c, k, d_over, d_under, early_out, ho); break;
        case 4: launch_rm_variant<NV, true, false, false, true>(c, k, d_over, d_under, early_out, ho); break;
        default: launch_rm_variant<NV, true, false, true, true>(c, k, d_over, d_under, early_out, ho); break;
        }
        return;
    }
    switch (sel) {
    case 0: launch_rm_variant<NV, false, false, false>(c, k, d_over, d_under, early_out, ho); break;
    case 1: launch_rm_variant<NV, false, false, true>(c, k, d_over, d_under, early_out, ho); break;
    case 2: launch_rm_variant<NV, false, true, false>(c, k, d_over, d_under, early_out, ho); break;
    case 3: launch_rm_variant<NV, false, true, true>(c, k, d_over, d_under, early_out, ho); break;
    case 4: launch_rm_variant<NV, true, false, false>(c, k, d_over, d_under, early_out, ho); break;
    case 5: launch_rm_variant<NV, true, false, true>(c, k, d_over, d_under, early_out, ho); break;
    case 6: launch_rm_variant<NV, true, true, false>(c, k, d_over, d_under, early_out, ho); break;
    default: launch_rm_variant<NV, true, true, true>(c, k, d_over, d_under, early_out, ho); break;
    }
}

}  // namespace
