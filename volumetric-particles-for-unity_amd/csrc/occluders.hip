// occluders.hip -- the two scene-occlusion inputs of the path, produced on the GPU from occluder solids (boxes, capped cylinders, ellipsoids)
// (SURVEY section 8(f) row 1).  In the reference they come from Unity's rasteriser:
//   * lightDepthMap: lightCamera.RenderWithShader(GenerateLightDepthMap) (VPR.cs:184): ortho camera at
//     gridCenter - fwd*200 (VPR.cs:365), extents = the grid's x/y size (VPR.cs:338-342), Cull Front + ZTest Less
//     (LDM.shader:6) => per texel the NEAREST BACK FACE, stored as D3D ortho depth (z - near)/(far - near);
//   * the main camera's depth buffer used by the ray-march's ZTest Less (VPR.cs:204, RM.shader:14): nearest front face.
// A rasteriser is not needed for convex analytic solids: one thread per texel / pixel intersects its (pixel-centre) ray with every solid.
// The reference's scene (Default layer, VPR.cs:346 cullingMask): two scaled cubes, two unit cubes, four unit cylinders (scene:1755,5462,8382,8623).
#include "vpfx_internal.h"

namespace {

struct CamRows { float r[12]; };       // rows of cameraToWorld's upper 3 x 4

// ray (o, d) vs oriented box: entry / exit parameters; false if missed.  d need not be normalised.
__device__ __forceinline__ bool ray_obb(const vp_occluder& b, float ox, float oy, float oz, float dx, float dy, float dz, float& t0, float& t1)
{
    const float px = ox - b.center[0], py = oy - b.center[1], pz = oz - b.center[2];
    t0 = -3.0e38f; t1 = 3.0e38f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float ax = b.axes[3 * k], ay = b.axes[3 * k + 1], az = b.axes[3 * k + 2];
        const float lo = (ax * px + ay * py) + az * pz;
        const float ld = (ax * dx + ay * dy) + az * dz;
        const float h = b.half_extent[k];
        if (ld != 0.f) {
            const float inv = 1.0f / ld;
            const float ta = (-h - lo) * inv, tb = (h - lo) * inv;
            t0 = fmaxf(t0, fminf(ta, tb));
            t1 = fminf(t1, fmaxf(ta, tb));
        } else if (lo < -h || lo > h) {
            return false;
        }
    }
    return t0 <= t1;
}

// ray vs capped cylinder (axis = local y) / ellipsoid in the solid's normalised local space l = diag(1/h) A (p - c): the map is affine, so t is
// the world ray's parameter.  Solved about the foot point (closest approach to the axis / centre): the light camera sits 200 units out and
// b^2 - ac about the origin would lose ~4 digits.  Operation order: fixed by DESIGN.md section 4 (no contraction; the CPU checker spells out the same sequence).
__device__ __forceinline__ bool ray_quadric(const vp_occluder& b, float ox, float oy, float oz, float dx, float dy, float dz, float& t0, float& t1)
{
    const bool cyl = b.type == VP_OCC_CYLINDER;
    const float wx = ox - b.center[0], wy = oy - b.center[1], wz = oz - b.center[2];
    float p[3], q[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float ax = b.axes[3 * k], ay = b.axes[3 * k + 1], az = b.axes[3 * k + 2];
        p[k] = ((ax * wx + ay * wy) + az * wz) / b.half_extent[k];
        q[k] = ((ax * dx + ay * dy) + az * dz) / b.half_extent[k];
    }
    const float qq = cyl ? q[0] * q[0] + q[2] * q[2] : (q[0] * q[0] + q[1] * q[1]) + q[2] * q[2];
    t0 = -3.0e38f; t1 = 3.0e38f;
    if (qq > 0.f) {
        const float pq = cyl ? p[0] * q[0] + p[2] * q[2] : (p[0] * q[0] + p[1] * q[1]) + p[2] * q[2];
        const float tc = -(pq / qq);
        const float f0 = p[0] + tc * q[0], f1 = p[1] + tc * q[1], f2 = p[2] + tc * q[2];
        const float dist2 = cyl ? f0 * f0 + f2 * f2 : (f0 * f0 + f1 * f1) + f2 * f2;
        if (dist2 > 1.0f) return false;
        const float half = sqrtf((1.0f - dist2) / qq);                // IEEE divide and square root (hipcc's default for fp32)
        t0 = tc - half; t1 = tc + half;
    } else {
        const float dist2 = cyl ? p[0] * p[0] + p[2] * p[2] : (p[0] * p[0] + p[1] * p[1]) + p[2] * p[2];
        if (dist2 > 1.0f) return false;
    }
    if (cyl) {
        if (q[1] != 0.f) {
            const float inv = 1.0f / q[1];
            const float ta = (-1.0f - p[1]) * inv, tb = (1.0f - p[1]) * inv;
            t0 = fmaxf(t0, fminf(ta, tb));
            t1 = fminf(t1, fmaxf(ta, tb));
        } else if (p[1] < -1.0f || p[1] > 1.0f) {
            return false;
        }
    }
    return t0 <= t1;
}

__device__ __forceinline__ bool ray_solid(const vp_occluder& b, float ox, float oy, float oz, float dx, float dy, float dz, float& t0, float& t1)
{
    return b.type == VP_OCC_BOX ? ray_obb(b, ox, oy, oz, dx, dy, dz, t0, t1) : ray_quadric(b, ox, oy, oz, dx, dy, dz, t0, t1);
}

__global__ void __launch_bounds__(256)
k_light_depth(GridConsts g, const vp_occluder* __restrict__ boxes, int n, float nearz, float farz, float cam_dist, float* __restrict__ out)
{
    const int LW = g.Nx * g.nv, LH = g.Ny * g.nv;
    const int X = blockIdx.x * 16 + (threadIdx.x & 15), Y = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (X >= LW || Y >= LH) return;
    // ortho frustum Ortho(-r, r, -t, t, near, far), r = Nx*s/2, t = Ny*s/2, centred on gridCenter     VPR.cs:338-342
    const float r = (float)g.Nx * g.s * 0.5f, t = (float)g.Ny * g.s * 0.5f;
    const float lx = -r + ((float)X + 0.5f) / (float)LW * (2.0f * r);
    const float ly = -t + ((float)Y + 0.5f) / (float)LH * (2.0f * t);
    const float cx = g.gc[0] - g.fwd[0] * cam_dist, cy = g.gc[1] - g.fwd[1] * cam_dist, cz = g.gc[2] - g.fwd[2] * cam_dist;
    const float ox = cx + g.Rl[0] * lx + g.Rl[1] * ly, oy = cy + g.Rl[3] * lx + g.Rl[4] * ly, oz = cz + g.Rl[6] * lx + g.Rl[7] * ly;
    float zmin = 3.0e38f;
    for (int i = 0; i < n; ++i) {
        float t0, t1;
        if (!ray_solid(boxes[i], ox, oy, oz, g.fwd[0], g.fwd[1], g.fwd[2], t0, t1)) continue;
        // Cull Front: the back face (exit point) is rasterised; it must be inside the clip volume
        if (t1 >= nearz && t1 <= farz) zmin = fminf(zmin, t1);
    }
    out[(size_t)Y * LW + X] = zmin < 3.0e38f ? (zmin - nearz) / (farz - nearz) : 1.0f;      // cleared depth = 1
}

__global__ void __launch_bounds__(256)
k_scene_depth(int W, int H, float aspect, float neg_inv_tan, float nearc, float farc, const CamRows cam /* cameraToWorld, rows, by value */,
              const vp_occluder* __restrict__ boxes, int n, float* __restrict__ out)
{
    const float* c2w = cam.r;
    const int col = blockIdx.x * 16 + (threadIdx.x & 15), row = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (col >= W || row >= H) return;
    float dx = (2.0f * ((float)col + 0.5f) / (float)W - 1.0f) * aspect;
    float dy = 2.0f * ((float)row + 0.5f) / (float)H - 1.0f;
    float dz = neg_inv_tan;                               // camera space looks down -z; depth = -z = t * (-dz) with this d
    const float wx = (c2w[0] * dx + c2w[1] * dy) + c2w[2] * dz, wy = (c2w[4] * dx + c2w[5] * dy) + c2w[6] * dz,
                wz = (c2w[8] * dx + c2w[9] * dy) + c2w[10] * dz;
    float best = 3.0e38f;
    for (int i = 0; i < n; ++i) {
        float t0, t1;
        if (!ray_solid(boxes[i], c2w[3], c2w[7], c2w[11], wx, wy, wz, t0, t1)) continue;
        const float te = t0 > 0.f ? t0 : t1;              // camera inside the box: its far wall is what is drawn
        const float depth = te * (-dz);
        if (te > 0.f && depth >= nearc && depth <= farc) best = fminf(best, depth);
    }
    out[(size_t)row * W + col] = best;
}

}  // namespace

int launch_light_depth(vp_ctx* c, float nearz, float farz, float cam_dist, float* d_out)
{
    const int LW = c->g.Nx * c->g.nv, LH = c->g.Ny * c->g.nv;
    hipLaunchKernelGGL(k_light_depth, dim3((LW + 15) / 16, (LH + 15) / 16), dim3(256), 0, c->stream, c->g, c->d_occluders, c->n_occluders,
                       nearz, farz, cam_dist, d_out);
    VP_HIP(hipGetLastError());
    return VP_OK;
}

int launch_scene_depth(vp_ctx* c, const vp_camera* cam, float* d_out)
{
    const int W = c->cfg.width, H = c->cfg.height;
    CamRows rows;                                         // a kernel argument: no copy command and no host wait in front of the frame's launches
    for (int r = 0; r < 3; ++r)
        for (int k = 0; k < 4; ++k) rows.r[r * 4 + k] = cam->camera_to_world[k * 4 + r];
    const float aspect = (float)W / (float)H;
    const float nit = -(1.0f / (float)tan((double)cam->fov_y * 0.5));
    hipLaunchKernelGGL(k_scene_depth, dim3((W + 15) / 16, (H + 15) / 16), dim3(256), 0, c->stream, W, H, aspect, nit, cam->near_clip,
                       cam->far_clip > 0.f ? cam->far_clip : 3.0e38f, rows, c->d_occluders, c->n_occluders, d_out);
    VP_HIP(hipGetLastError());
    return VP_OK;
}
