// fill_generic.hip -- the FillVolume pass for every numVoxelsInMetavoxel that is not 16 / 32 / 64.
//
// The reference takes the voxel count from a public inspector field (VPR.cs:84), sizes the 3D textures with it as is (VPR.cs:312-314) and
// hands it to the shader as the float uniform _NumVoxels (VPR.cs:527); FillVolume.shader runs any value up to its NUM_VOXELS = 32 column
// array (Fill.shader:16,156,164).  libvpfx's benchmark kernels carry the count as a template constant; the instantiations here carry it
// as a run-time value (GEN = true in fill_kernels.h: overhanging 8x8-column tiles clamped and masked, the register-array capacity class
// NV = 32 for counts up to 32 and NV = 64 above) -- same arithmetic in the same order per voxel, so the EXACT build stays bit-identical to the
// oracle and the default build within 1 fp16 ulp, at whatever speed the generic index arithmetic allows.  An R8 cube map that fits LDS
// still runs the persistent LDS kernel (row pitch as a value: TAB = 2, for every S).
#include "fill_kernels.h"

namespace {

template <int NV>
int fill_generic_nv(vp_ctx* c, int mode, const FillPtrs& P, int math, bool lds)
{
    if (!lds) return launch_fill_nv<NV, true>(c, mode, P, math);
    return math == 2 ? launch_fill_lds_nv<NV, true, true>(c, mode, P) : launch_fill_lds_nv<NV, false, true>(c, mode, P);
}

template <int NV>
void fill_one_generic_nv(vp_ctx* c, const GridConsts& g, const FillPtrs& P, int math)
{
    const int tw = (g.nv + 15) / 16;
    const dim3 grid(tw * tw), block(256);
    if (math == 1)      hipLaunchKernelGGL((k_fill<NV, 1, 0, false, true>), grid, block, 0, c->stream, g, c->fc, FILL_PTR_ARGS(P), FillChain{}, (int*)nullptr);
    else if (math == 2) hipLaunchKernelGGL((k_fill<NV, 2, 0, false, true>), grid, block, 0, c->stream, g, c->fc, FILL_PTR_ARGS(P), FillChain{}, (int*)nullptr);
    else                hipLaunchKernelGGL((k_fill<NV, 0, 0, false, true>), grid, block, 0, c->stream, g, c->fc, FILL_PTR_ARGS(P), FillChain{}, (int*)nullptr);
}

}  // namespace

// launch_fill's body for a run-time voxel count (the caller records the stage events and the brick format)
int launch_fill_generic(vp_ctx* c, int mode, const float* d_light_in, float* d_light_out, int math, bool lds)
{
    if (c->g.nv < 2 || c->g.nv > 64) return vp_fail(c, VP_ERR_UNSUPPORTED, "num_voxels %d outside [2, 64]", c->g.nv);
    const FillPtrs P = fill_ptrs(c, d_light_in, d_light_out, nullptr);
    return c->g.nv <= 32 ? fill_generic_nv<32>(c, mode, P, math, lds) : fill_generic_nv<64>(c, mode, P, math, lds);
}

// launch_fill_one's kernel launch for a run-time voxel count (g: the context's grid constants restricted to the one metavoxel slice)
int launch_fill_one_generic(vp_ctx* c, const GridConsts& g, int math)
{
    if (g.nv < 2 || g.nv > 64) return vp_fail(c, VP_ERR_UNSUPPORTED, "num_voxels %d outside [2, 64]", g.nv);
    const FillPtrs P = fill_ptrs(c, c->d_lightmap, c->d_lightmap, c->d_onecol);
    if (g.nv <= 32) fill_one_generic_nv<32>(c, g, P, math); else fill_one_generic_nv<64>(c, g, P, math);
    return VP_OK;
}
