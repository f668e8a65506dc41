// fill.hip -- the FillVolume pass: launch entry points of libvpfx's fill (launch_fill, launch_fill_one, the cube-map tables) and the
// instantiations of the kernel templates of fill_kernels.h for numVoxelsInMetavoxel = 16 / 32 / 64 (the voxel count a compile-time
// constant: the benchmark configurations).  Every other voxel count goes to fill_generic.hip.
#define VPFX_FILL_MAIN_TU 1
#include "fill_kernels.h"

namespace {

// Expand the cubemap into the footprint table.  A bilinear footprint is the quad of texels (ix,iy),(ix,iy+1),(ix+1,iy),
// (ix+1,iy+1) with clamp addressing, ix, iy in [-1, S-1] (the two x-neighbours of each row land in registers q.x/q.z and
// q.y/q.w, so the x-lerp is one packed sub + one packed fma on consecutive register pairs).  The table stores COLUMN PAIRS
// P(face, iy, ix) = texels (ix,iy),(ix,iy+1) for ix in [-1, S], iy in [-1, S-1]: the 16 bytes at P(iy, ix) are
// P(iy, ix), P(iy, ix+1) = that quad, fetched with one 8-byte-aligned 16-byte load.  Against one float4 per quad this
// halves the table (0.8 MB at S = 128) and a 128-byte line spans 16 texels in x instead of 8, so neighbouring lanes and
// slices share more lines: the per-voxel gather is bound by L2 requests (every lane pulls its own line), and this cut
// k_fill from 5.57 to 5.09 ms at C3.
// The source is the caller's cube map as uploaded: f32 texels, or R8 UNORM bytes (texel = byte / 255, the D3D11 UNORM -> float
// conversion; the reference's asset is 8-bit).  A texel outside [0, 1] (or NaN) raises *bad: netDisplacement >= 0 is what
// Fill.shader:119-126 and the ao max rely on, so vp_fill refuses such a map instead of silently diverging.
__device__ __forceinline__ float cube_texel(const float* c, size_t i) { return c[i]; }
__device__ __forceinline__ float cube_texel(const uint8_t* c, size_t i) { return (float)c[i] / 255.0f; }

template <typename T>
__global__ void __launch_bounds__(256)
k_build_cubequads(const T* __restrict__ cube, int S, float4* __restrict__ quads_, int* __restrict__ bad)
{
    float2* pairs = reinterpret_cast<float2*>(quads_);
    const int n = 6 * (S + 1) * (S + 2);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int ix = i % (S + 2) - 1, iy = (i / (S + 2)) % (S + 1) - 1, face = i / ((S + 2) * (S + 1));
    const int x0 = min(max(ix, 0), S - 1);
    const int y0 = min(max(iy, 0), S - 1), y1 = min(max(iy + 1, 0), S - 1);
    const size_t fo = (size_t)face * S * S;
    const float a = cube_texel(cube, fo + y0 * S + x0), b = cube_texel(cube, fo + y1 * S + x0);
    if (!(a >= 0.f && a <= 1.f) || !(b >= 0.f && b <= 1.f)) *bad = 1;
    pairs[i] = make_float2(a, b);
}

__global__ void __launch_bounds__(256)
k_fill_value(float* __restrict__ d, size_t n, float v, int* __restrict__ zero_word /* nullable */)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i == 0 && zero_word) *zero_word = 0;
    if (i < n) d[i] = v;
}

// LDS image of an R8 cube map: bytes [face][S+2][S+2], texel (ix, iy) for ix, iy in [-1, S] with clamp addressing baked in, so the
// kernel's quad (ix, iy) .. (ix+1, iy+1), ix, iy in [-1, S-1], is four unguarded byte reads.
__global__ void __launch_bounds__(256)
k_build_cube_u8(const uint8_t* __restrict__ cube, int S, int pitch, uint8_t* __restrict__ out, int nbytes)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nbytes) return;
    const int S2 = S + 2, n = 6 * S2 * pitch;
    uint8_t v = 0;
    if (i < n) {
        const int ix = i % pitch - 1, iy = (i / pitch) % S2 - 1, face = i / (S2 * pitch);      // columns beyond S + 1 are padding
        v = cube[(size_t)face * S * S + min(max(iy, 0), S - 1) * S + min(max(ix, 0), S - 1)];
    }
    out[i] = v;
}

}  // namespace

int launch_fill_value(vp_ctx* c, float* d, size_t n, float v, int* zero_word)
{
    hipLaunchKernelGGL(k_fill_value, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, d, n, v, zero_word);
    VP_HIP(hipGetLastError());
    return VP_OK;
}

int cube_u8_pitch(int S) { return S == 128 ? VPFX_LDS_PITCH_128 : S + 2; }
size_t cube_u8_bytes(int S) { return (((size_t)6 * (S + 2) * cube_u8_pitch(S)) + 15) & ~(size_t)15; }

int launch_build_cube_u8(vp_ctx* c, const void* d_cube_r8, int S)
{
    const int nbytes = (int)cube_u8_bytes(S);
    hipLaunchKernelGGL(k_build_cube_u8, dim3((nbytes + 255) / 256), dim3(256), 0, c->stream, (const uint8_t*)d_cube_r8, S, cube_u8_pitch(S), (uint8_t*)c->d_cube_u8, nbytes);
    VP_HIP(hipGetLastError());
    return VP_OK;
}

int launch_build_cubequads(vp_ctx* c, const void* d_cube, int format, int S, int* d_bad)
{
    const int n = 6 * (S + 1) * (S + 2);
    if (format == VP_CUBEMAP_R8)
        hipLaunchKernelGGL(k_build_cubequads<uint8_t>, dim3((n + 255) / 256), dim3(256), 0, c->stream, (const uint8_t*)d_cube, S, c->d_cubequads, d_bad);
    else
        hipLaunchKernelGGL(k_build_cubequads<float>, dim3((n + 255) / 256), dim3(256), 0, c->stream, (const float*)d_cube, S, c->d_cubequads, d_bad);
    VP_HIP(hipGetLastError());
    return VP_OK;
}

// FillMetavoxel(xx, yy, zz) (VPR.cs:559-609): the same kernel restricted to one metavoxel -- one MV column, zz in [zz, zz+1),
// incoming light read from and transmitted light written back to the light-propagation map, exactly the UAV traffic of one
// reference draw (Fill.shader:224, 250).
int launch_fill_one(vp_ctx* c, int xx, int yy, int zz)
{
    const int col = yy * c->g.Nx + xx;
    VP_HIP(hipMemcpyAsync(c->d_onecol, &col, sizeof(int), hipMemcpyHostToDevice, c->stream));   // pageable source: consumed on return
    const FillPtrs P = fill_ptrs(c, c->d_lightmap, c->d_lightmap, c->d_onecol);
    GridConsts g = c->g;
    g.z0 = zz; g.z1 = zz + 1;
    const int math = c->cfg.exact_math == 1 ? 1 : c->fc.d_is_one ? 2 : 0;  // see launch_fill
    const dim3 block(256);
#define VPFX_FILL_ONE(NV)                                                                                            \
    do {                                                                                                              \
        const dim3 grid((NV / 16) * (NV / 16));                                                                       \
        if (math == 1)      hipLaunchKernelGGL((k_fill<NV, 1, 0, false>), grid, block, 0, c->stream, g, c->fc, FILL_PTR_ARGS(P), FillChain{}, (int*)nullptr);  \
        else if (math == 2) hipLaunchKernelGGL((k_fill<NV, 2, 0, false>), grid, block, 0, c->stream, g, c->fc, FILL_PTR_ARGS(P), FillChain{}, (int*)nullptr);  \
        else                hipLaunchKernelGGL((k_fill<NV, 0, 0, false>), grid, block, 0, c->stream, g, c->fc, FILL_PTR_ARGS(P), FillChain{}, (int*)nullptr);  \
    } while (0)
    switch (c->g.nv) {
    case 16: VPFX_FILL_ONE(16); break;
    case 32: VPFX_FILL_ONE(32); break;
    case 64: VPFX_FILL_ONE(64); break;
    default: { int rc = launch_fill_one_generic(c, g, math); if (rc) return rc; } break;      // any other voxel count: fill_generic.hip
    }
#undef VPFX_FILL_ONE
    VP_HIP(hipGetLastError());
    return VP_OK;      // (the pool's format is fixed by vp_fill_begin: c->fc.grey == c->bricks_grey here)
}

int launch_fill(vp_ctx* c, int mode, const float* d_light_in, float* d_light_out)
{
    const FillPtrs P = fill_ptrs(c, d_light_in, d_light_out, nullptr);
    // math: 1 = IEEE divisions everywhere (parity builds, float table only); 2 = displacement scale exactly 1, where net displacement
    // == texel and the reference's smoothstep(net, 0.7 net, x) jumps at net == 0: default math with the flip-deciding quantities exact
    // (cube_address; round 2 sent these fills to the EXACT kernels at about twice the time); 0 = default.
    const int math = c->cfg.exact_math == 1 ? 1 : c->fc.d_is_one ? 2 : 0;
    // R8 cube map resident as a byte table that fits LDS: the persistent LDS kernel (default math only; EXACT keeps the f32 table)
    const bool lds = mode != 2 && math != 1 && c->cube_u8_S > 0 && c->cube_u8_S == c->cubeS && c->cfg.reserved[0] != VPFX_CFG_NO_LDS_CUBEMAP;
    const int evi = mode == 2 ? 3 : 1;
    VP_HIP(hipEventRecord(c->ev[evi][0], c->stream));
    int rc = VP_OK;
    switch (c->g.nv) {
#define VPFX_FILL_NV(NV) rc = !lds ? launch_fill_nv<NV>(c, mode, P, math) : math == 2 ? launch_fill_lds_nv<NV, true>(c, mode, P) : launch_fill_lds_nv<NV, false>(c, mode, P)
    case 16: VPFX_FILL_NV(16); break;
    case 32: VPFX_FILL_NV(32); break;
    case 64: VPFX_FILL_NV(64); break;
#undef VPFX_FILL_NV
    default: rc = launch_fill_generic(c, mode, d_light_in, d_light_out, math, lds); break;   // any other voxel count: fill_generic.hip
    }
    if (rc) return rc;
    VP_HIP(hipGetLastError());
    if (mode != 1) c->bricks_grey = c->fc.grey != 0;          // the format the bricks now hold
    VP_HIP(hipEventRecord(c->ev[evi][1], c->stream));
    c->ev_valid[evi] = true;
    return VP_OK;
}
