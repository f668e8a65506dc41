// raymarch_generic.hip -- the RayMarch pass for every numVoxelsInMetavoxel that is not 16 / 32 / 64 (the reference passes the inspector
// value to the shader as the float uniform _NumVoxels, VPR.cs:84, 722, and sizes the volume textures with it, VPR.cs:312-314): the kernel
// templates of raymarch_kernels.h with NV = 0 -- brick strides and the wrap of a border-less brick computed from RmConsts::nv at run time,
// the footprint's second row a pointer step instead of an instruction immediate.  Same arithmetic per sample in the same order.
#include "raymarch_kernels.h"

int launch_raymarch_generic(vp_ctx* c, const RmConsts& k, float* d_over, float* d_under, int early_out, const RmHandoff& ho)
{
    if (k.nv < 2 || k.nv > 64) return vp_fail(c, VP_ERR_UNSUPPORTED, "num_voxels %d outside [2, 64]", k.nv);
    launch_rm_nv<0>(c, k, d_over, d_under, early_out, ho);
    return VP_OK;
}

int launch_raymarch_one_generic(vp_ctx* c, const RmConsts& k, const void* brick_v, const float* tr, int blend_over, int order_index, float* d_img)
{
    if (k.nv < 2 || k.nv > 64) return vp_fail(c, VP_ERR_UNSUPPORTED, "num_voxels %d outside [2, 64]", k.nv);
    const uint2* brick = (const uint2*)brick_v;
    const float4 trv = make_float4(tr[0], tr[1], tr[2], 0.f);
    const dim3 grid((k.W + 15) / 16, (k.H + 15) / 16), block(256);
    if (c->bricks_grey)
        hipLaunchKernelGGL((k_raymarch_one<0, false, true>), grid, block, 0, c->stream, k, brick, trv, c->d_scene_depth, (float4*)d_img, blend_over,
                           order_index, c->d_samples);
    else if (c->g.b < 1)
        hipLaunchKernelGGL((k_raymarch_one<0, true, false>), grid, block, 0, c->stream, k, brick, trv, c->d_scene_depth, (float4*)d_img, blend_over,
                           order_index, c->d_samples);
    else
        hipLaunchKernelGGL((k_raymarch_one<0, false, false>), grid, block, 0, c->stream, k, brick, trv, c->d_scene_depth, (float4*)d_img, blend_over,
                           order_index, c->d_samples);
    return VP_OK;
}
