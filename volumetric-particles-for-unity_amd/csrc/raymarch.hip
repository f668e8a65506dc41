// raymarch.hip -- the RayMarch pass: launch entry points of libvpfx's ray-march (launch_raymarch, launch_raymarch_one, blend, composite), the
// per-frame preparation kernel, and the instantiations of the kernel templates of raymarch_kernels.h for numVoxelsInMetavoxel = 16 / 32 / 64
// (the voxel count a compile-time constant: the benchmark configurations).  Every other voxel count goes to raymarch_generic.hip.
#define VPFX_RM_MAIN_TU 1
#include "raymarch_kernels.h"

namespace {

// Dispatch order of the screen super-tiles (64x32 px): most expensive first, so that the long rays are not what the
// tail of the launch waits for (per-wave work spans 0 .. ~600 samples; dispatched in raster order ~25 % of the wave
// slots idle).  Cost estimate per super-tile = fraction of the ray inside occupied metavoxels (x its length in the owned
// part of the grid), from four of its rays; each ray is probed at 64 points by the 64 lanes of one wave, so the estimate
// costs ONE dependent load.  Run by the trailing workgroups of k_rm_prepare (one per super-tile); k_tile_rank then rank-sorts in LDS.  Scheduling only: the image does not depend on the order.
__device__ __forceinline__ void tile_cost(const RmConsts& k, const int* __restrict__ brick_index, int sgx, float* __restrict__ cost_out, const int sti)
{
    __shared__ float part[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    constexpr int SW = 16 << VPFX_RM_LX, SH = 16 << VPFX_RM_LY;                    // super-tile size in pixels
    const float colf = fminf((float)((sti % sgx) * SW + SW / 4 + (SW / 2) * (wave & 1)), (float)k.W - 1.f);
    const float rowf = fminf((float)((sti / sgx) * SH + SH / 4 + (SH / 2) * (wave >> 1)), (float)k.H - 1.f);
    float dx = (2.0f * (colf + 0.5f) / (float)k.W) - 1.0f;
    const float dy = (2.0f * (rowf + 0.5f) / (float)k.H) - 1.0f;
    dx *= k.aspect;
    const float dz = k.neg_inv_tan;
    const float inv = 1.0f / sqrtf((dx * dx + dy * dy) + dz * dz);
    const float dir[3] = {dx * inv, dy * inv, dz * inv};
    const float st = k.zMin / dir[2];
    float o[3], d[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        d[r] = (k.c2g[4 * r] * dir[0] + k.c2g[4 * r + 1] * dir[1]) + k.c2g[4 * r + 2] * dir[2];
        o[r] = d[r] * st + k.c2g[4 * r + 3];
    }
    const float dn = 1.0f / sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
    const float lo[3] = {0.f, 0.f, (float)k.z0}, hi[3] = {(float)k.Nx, (float)k.Ny, (float)k.z1};
    const float cx = k.camg[0] - o[0], cy = k.camg[1] - o[1], cz = k.camg[2] - o[2];
    float t0 = sqrtf((cx * cx + cy * cy) + cz * cz), t1 = 3.0e38f;          // nothing is sampled behind the camera
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        d[r] *= dn;
        const float iv = 1.0f / d[r];                                       // +-inf for an axis-parallel ray: the slab test still works
        const float a = iv * (lo[r] - o[r]), b = iv * (hi[r] - o[r]);
        t0 = fmaxf(t0, fminf(a, b)); t1 = fminf(t1, fmaxf(a, b));
    }
    float c = 0.f;
    if (t0 < t1) {
        const float t = t0 + (t1 - t0) * (((float)lane + 0.5f) * (1.0f / 64.0f));
        const int gx = (int)floorf(fmaf(t, d[0], o[0])), gy = (int)floorf(fmaf(t, d[1], o[1])), gz = (int)floorf(fmaf(t, d[2], o[2]));
        const bool occ = gx >= 0 && gx < k.Nx && gy >= 0 && gy < k.Ny && gz >= k.z0 && gz < k.z1 &&
                         brick_index[(gz * k.Ny + gy) * k.Nx + gx] >= 0;
        c = (float)__popcll(__ballot(occ)) * (t1 - t0);
    }
    if (lane == 0) part[wave] = c;
    __syncthreads();
    if (threadIdx.x == 0) cost_out[sti] = (part[0] + part[1]) + (part[2] + part[3]);
}

// rank[yy * Nx + xx] = position of the column (xx, yy) in the ASCENDING list of squared distances of its zz = 0 metavoxel from the camera -- the
// reference's List<MetavoxelSortData>.Sort by distance (VPR.cs:613-632; stable on ties, the list is built yy-major): phase A (OVER) walks the
// ranks descending, phase B (UNDER) ascending.  Ranked here, in k_rm_prepare's last workgroups (the host used to sort and upload it: one
// pageable copy command per frame in front of the frame's launches): the threads of column i count the columns that sort in front of it.
#define RM_RANK_LDS 4096
#define RM_RANK_COLS 16          // columns ranked per workgroup: its 256 threads form 16 groups, group g counts over the g-th sixteenth of the keys
__device__ __forceinline__ void column_rank(const RmConsts& k, const float* __restrict__ mvPos, int* __restrict__ rank_out, const int blk)
{
    __shared__ __attribute__((aligned(16))) float keys[RM_RANK_LDS];
    __shared__ int partial[RM_RANK_COLS];
    const int nxy = k.Nx * k.Ny;
    auto key_of = [&](int j) -> float {
        const float dx = mvPos[3 * j] - k.cam_world[0], dy = mvPos[3 * j + 1] - k.cam_world[1], dz = mvPos[3 * j + 2] - k.cam_world[2];
        return (dx * dx + dy * dy) + dz * dz;
    };
    const int i = blk * RM_RANK_COLS + ((int)threadIdx.x & (RM_RANK_COLS - 1)), chunk = (int)threadIdx.x / RM_RANK_COLS;
    if (nxy > RM_RANK_LDS) {                              // a grid wider than the LDS table: group 0 recomputes the keys from global memory
        if (chunk == 0 && i < nxy) {
            const float ki = key_of(i);
            int r = 0;
            for (int j = 0; j < nxy; ++j) { const float kj = key_of(j); r += (kj < ki || (kj == ki && j < i)) ? 1 : 0; }
            rank_out[i] = r;
        }
        return;
    }
    const int npad = (nxy + 3) & ~3;
    for (int j = threadIdx.x; j < npad; j += 256) keys[j] = j < nxy ? key_of(j) : 3.0e38f;      // padding never sorts in front of a column
    if (threadIdx.x < RM_RANK_COLS) partial[threadIdx.x] = 0;
    __syncthreads();
    if (i < nxy) {
        constexpr int NCHUNK = 256 / RM_RANK_COLS;
        const int per = ((npad + 4 * NCHUNK - 1) / (4 * NCHUNK)) * 4, j0 = chunk * per, j1 = min(npad, j0 + per);
        const float ki = keys[i];
        int r = 0;
        for (int j = j0; j < j1; j += 4) {
            const float4 kj = *reinterpret_cast<const float4*>(&keys[j]);
            r += (kj.x < ki || (kj.x == ki && j < i)) ? 1 : 0;
            r += (kj.y < ki || (kj.y == ki && j + 1 < i)) ? 1 : 0;
            r += (kj.z < ki || (kj.z == ki && j + 2 < i)) ? 1 : 0;
            r += (kj.w < ki || (kj.w == ki && j + 3 < i)) ? 1 : 0;
        }
        atomicAdd(&partial[threadIdx.x & (RM_RANK_COLS - 1)], r);
    }
    __syncthreads();
    if (chunk == 0 && i < nxy) rank_out[i] = partial[threadIdx.x];
}

// Everything the march needs per frame in ONE launch over all N^3 cells (it was four memsets and a kernel over the occupied metavoxels):
// the translation column of _CameraToMetavoxel = TRS(mvPos, lightRot, s).inverse * cameraToWorld of every occupied metavoxel (VPR.cs:774-778,
// same operation order as the matrix product the reference does per draw) into mvtrans[slot] and into the cell's record, "empty" records for the other cells, the
// occupancy bitmask rows (the thread of a row's first cell builds the word), brick_hit[slot] = 0, the sample counter = 0.
__global__ void __launch_bounds__(256)
k_rm_prepare(RmConsts k, const int* __restrict__ brick_index, const float* __restrict__ mvPos, int n3, float4* __restrict__ mvtrans,
             float4* __restrict__ cellinfo /* nullable */, uint32_t* __restrict__ occmask /* nullable */, int* __restrict__ brick_hit,
             unsigned long long* __restrict__ samples, int nprep, int sgx, float* __restrict__ cost_out, int ncost, int* __restrict__ rank_out)
{
    if ((int)blockIdx.x >= nprep + ncost) { column_rank(k, mvPos, rank_out, (int)blockIdx.x - nprep - ncost); return; }
    // workgroups past the cells' own: one super-tile's cost estimate each (tile_cost; it reads brick_index only -- nothing this launch writes).
    // One launch instead of two: a frame of the reference's scene is 0.24 ms and a launch is 4 us of it.
    if ((int)blockIdx.x >= nprep) { tile_cost(k, brick_index, sgx, cost_out, (int)blockIdx.x - nprep); return; }
    const int mi = blockIdx.x * 256 + threadIdx.x;
    if (mi == 0) *samples = 0ull;
    if (mi >= n3) return;
    const int slot = brick_index[mi];
    if (slot >= 0) {
        const float mx = mvPos[3 * mi], my = mvPos[3 * mi + 1], mz = mvPos[3 * mi + 2];
        float tr[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float a = k.inv_rows[r * 3], b = k.inv_rows[r * 3 + 1], c = k.inv_rows[r * 3 + 2];
            const float t = -((a * mx + b * my) + c * mz);
            tr[r] = ((a * k.c2w_t[0] + b * k.c2w_t[1]) + c * k.c2w_t[2]) + t * k.c2w_t[3];
        }
        mvtrans[slot] = make_float4(tr[0], tr[1], tr[2], 0.f);
        if (cellinfo) cellinfo[mi] = make_float4(tr[0], tr[1], tr[2], __int_as_float(slot));
        brick_hit[slot] = 0;
    } else if (cellinfo) {
        const float e = __int_as_float(-1);
        cellinfo[mi] = make_float4(e, e, e, e);                          // the walk reads the record of every cell it crosses
    }
    if (occmask && mi % k.Nx == 0) {
        uint32_t word = 0;
        for (int x = 0; x < k.Nx; ++x) word |= (brick_index[mi + x] >= 0 ? 1u : 0u) << x;
        occmask[mi / k.Nx] = word;                                       // row (zz, yy)
    }
}

// _OrderIndex of every occupied metavoxel = its position in the submission order of RenderMetavoxels (mvCount, VPR.cs:650-706):
// phase A (zz <= zBoundary) zz ascending, cells far -> near (rank descending); phase B zz ascending, cells near -> far.  Brick
// slots are z-major, so the MVs of slice zz occupy consecutive slots and "occupied MVs in front of slice zz" = own slot minus the
// occupied cells of the slice with a smaller cell index.  Only run for the VP_RM_SHOW_DRAW_ORDER view; stored in mvtrans[].w.
__global__ void __launch_bounds__(256)
k_order_index(RmConsts k, const int* __restrict__ occ_list, const int* __restrict__ brick_index, const int* __restrict__ rank, int n,
              float4* __restrict__ mvtrans)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int nxy = k.Nx * k.Ny, mi = occ_list[i], zz = mi / nxy, cell = mi % nxy;
    const int* occ = brick_index + zz * nxy;
    const int r = rank[cell];
    const bool phaseA = zz <= k.zB;
    int before_in_slice = 0, lower_cells = 0;
    for (int j = 0; j < nxy; ++j) {
        if (occ[j] < 0) continue;
        lower_cells += j < cell ? 1 : 0;
        before_in_slice += (phaseA ? rank[j] > r : rank[j] < r) ? 1 : 0;
    }
    mvtrans[i].w = __int_as_float((i - lower_cells) + before_in_slice);
}

// Ordered blend of partial images (slab granularity of VPR.cs:652-711): kinds[i] 0 = OVER, 1 = UNDER.
#define MAX_PARTIALS 32
struct BlendArgs { const float4* img[MAX_PARTIALS]; int kind[MAX_PARTIALS]; int n; };

__global__ void __launch_bounds__(256)
k_blend(BlendArgs a, size_t npix, float4* __restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npix) return;
    float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = 0; j < a.n; ++j) {
        const float4 s = a.img[j][i];
        if (a.kind[j] == 0) { const float ia = 1.0f - s.w; d.x = s.x + d.x * ia; d.y = s.y + d.y * ia; d.z = s.z + d.z * ia; d.w = s.w + d.w * ia; }
        else { const float ia = 1.0f - d.w; d.x = s.x * ia + d.x; d.y = s.y * ia + d.y; d.z = s.z * ia + d.z; d.w = s.w * ia + d.w; }
    }
    out[i] = d;
}

// CompositeParticles.shader: Blend One OneMinusSrcAlpha, One One                           Comp.shader:10
__global__ void __launch_bounds__(256)
k_composite(const float4* __restrict__ particles, float4* __restrict__ scene, size_t npix)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npix) return;
    const float4 s = particles[i];
    float4 d = scene[i];
    const float ia = 1.0f - s.w;
    d.x = s.x + d.x * ia; d.y = s.y + d.y * ia; d.z = s.z + d.z * ia; d.w = s.w + d.w;
    scene[i] = d;
}

}  // namespace

int launch_raymarch(vp_ctx* c, const RmConsts& k_in, float* d_over, float* d_under, const RmHandoff* handoff)
{
    RmConsts k = k_in;
    k.occ_lds = (VPFX_RM_OCC_LDS && VPFX_RM_CELLINFO && k.Nx <= 32 && k.Nz * k.Ny <= VPFX_RM_OCC_WORDS) ? 1 : 0;
    const RmHandoff ho = (handoff && d_under) ? *handoff : RmHandoff{};          // slab (partial-image) kernels only
    const int early_out = (c->cfg.no_early_out == 1 || (k.flags & (VP_RM_NO_EARLY_OUT | VP_RM_SHOW_NUM_SAMPLES | VP_RM_SHOW_BLEND_FUNC | VP_RM_SHOW_DRAW_ORDER))) ? 0 : 1;
    k.flags &= ~VP_RM_NO_EARLY_OUT;                     // a launch switch, not a path of the FLAGS kernels
    const int nocc = c->h_meta.occupied;
    if ((size_t)nocc > c->mvtrans_cap) {
        if (c->d_mvtrans) VP_HIP(hipFree(c->d_mvtrans));
        c->d_mvtrans = nullptr; c->mvtrans_cap = 0;
        VP_HIP(hipMalloc((void**)&c->d_mvtrans, ((size_t)nocc + nocc / 8 + 16) * sizeof(float4)));
        c->mvtrans_cap = (size_t)nocc + nocc / 8 + 16;
    }
    if ((size_t)nocc > c->brick_hit_cap) {
        if (c->d_brick_hit) VP_HIP(hipFree(c->d_brick_hit));
        c->d_brick_hit = nullptr; c->brick_hit_cap = 0;
        VP_HIP(hipMalloc((void**)&c->d_brick_hit, ((size_t)nocc + nocc / 8 + 16) * sizeof(int)));
        c->brick_hit_cap = (size_t)nocc + nocc / 8 + 16;
    }
#if VPFX_RM_CELLINFO
    if (!c->d_cellinfo) VP_HIP(hipMalloc((void**)&c->d_cellinfo, c->n3 * sizeof(float4)));
    if (k.occ_lds && !c->d_occmask) VP_HIP(hipMalloc((void**)&c->d_occmask, VPFX_RM_OCC_WORDS * sizeof(uint32_t)));
#endif
    c->brick_hit_n = nocc;
    // one launch prepares the frame: translations, per-cell records, occupancy rows, cleared hit flags and sample counter, and -- its trailing
    // workgroups -- the super-tiles' cost estimates for the dispatch order (k_rm_prepare).  The stage's start event sits in front of it:
    // the estimate used to be a launch of its own inside the timed stage.
    VP_HIP(hipEventRecord(c->ev[2][0], c->stream));
    {
        const int nprep = (int)((c->n3 + 255) / 256), nsuper = rm_num_super_tiles(k.W, k.H);
        const bool ordered = rm_ordered(c, k);
        float* cost = reinterpret_cast<float*>(c->d_tile_order + rm_order_ints(nsuper));
        const int ncost = ordered ? nsuper : 0, nrank = (k.Nx * k.Ny + RM_RANK_COLS - 1) / RM_RANK_COLS;
        hipLaunchKernelGGL(k_rm_prepare, dim3((unsigned)(nprep + ncost + nrank)), dim3(256), 0, c->stream, k, c->d_brick_index, c->d_mvPos,
                           (int)c->n3, c->d_mvtrans, c->d_cellinfo, k.occ_lds ? c->d_occmask : (uint32_t*)nullptr, c->d_brick_hit, c->d_samples,
                           nprep, rm_super_tiles_x(k.W), cost, ncost, c->d_rank);
    }
    if (nocc > 0) {
        if (k.flags & VP_RM_SHOW_DRAW_ORDER)
            hipLaunchKernelGGL(k_order_index, dim3((nocc + 255) / 256), dim3(256), 0, c->stream, k, c->d_occ_list, c->d_brick_index, c->d_rank,
                               nocc, c->d_mvtrans);
    }
    switch (k.nv) {
    case 16: launch_rm_nv<16>(c, k, d_over, d_under, early_out, ho); break;
    case 32: launch_rm_nv<32>(c, k, d_over, d_under, early_out, ho); break;
    case 64: launch_rm_nv<64>(c, k, d_over, d_under, early_out, ho); break;
    default: { int rc = launch_raymarch_generic(c, k, d_over, d_under, early_out, ho); if (rc) return rc; } break;   // any other voxel count: raymarch_generic.hip
    }
    VP_HIP(hipGetLastError());
    VP_HIP(hipEventRecord(c->ev[2][1], c->stream));
    c->ev_valid[2] = true;
    return VP_OK;
}

int launch_raymarch_one(vp_ctx* c, const RmConsts& k, int bi, int mi, int blend_over, int order_index, float* d_img)
{
    // translation column of this metavoxel's _CameraToMetavoxel, same operation order as k_rm_prepare (VPR.cs:774-778)
    const float* mp = c->h_mvPos + 3 * (size_t)mi;
    float tr[3];
    for (int r = 0; r < 3; ++r) {
        const float a = k.inv_rows[r * 3], b = k.inv_rows[r * 3 + 1], cc = k.inv_rows[r * 3 + 2];
        const float t = -((a * mp[0] + b * mp[1]) + cc * mp[2]);
        tr[r] = ((a * k.c2w_t[0] + b * k.c2w_t[1]) + cc * k.c2w_t[2]) + t * k.c2w_t[3];
    }
    const float4 trv = make_float4(tr[0], tr[1], tr[2], 0.f);
    const dim3 grid((k.W + 15) / 16, (k.H + 15) / 16), block(256);
    const bool wrap = c->g.b < 1;
    const size_t nv3 = (size_t)k.nv * k.nv * k.nv;
    const uint2* brick = c->d_bricks + (size_t)bi * nv3;
#define VPFX_RM_ONE(NV)                                                                                                        \
    do {                                                                                                                        \
        if (c->bricks_grey) hipLaunchKernelGGL((k_raymarch_one<NV, false, true>), grid, block, 0, c->stream, k, brick, trv,      \
                                     c->d_scene_depth, (float4*)d_img, blend_over, order_index, c->d_samples);                  \
        else if (wrap) hipLaunchKernelGGL((k_raymarch_one<NV, true, false>), grid, block, 0, c->stream, k, brick, trv, c->d_scene_depth,    \
                                     (float4*)d_img, blend_over, order_index, c->d_samples);                                    \
        else      hipLaunchKernelGGL((k_raymarch_one<NV, false, false>), grid, block, 0, c->stream, k, brick, trv, c->d_scene_depth,   \
                                     (float4*)d_img, blend_over, order_index, c->d_samples);                                    \
    } while (0)
    switch (k.nv) {
    case 16: VPFX_RM_ONE(16); break;
    case 32: VPFX_RM_ONE(32); break;
    case 64: VPFX_RM_ONE(64); break;
    default: { int rc = launch_raymarch_one_generic(c, k, brick, tr, blend_over, order_index, d_img); if (rc) return rc; } break;
    }
#undef VPFX_RM_ONE
    VP_HIP(hipGetLastError());
    return VP_OK;
}

int launch_blend(vp_ctx* c, const void* const* d_partials, const int32_t* kinds, int n, float* d_out, size_t npix)
{
    if (n > MAX_PARTIALS) return vp_fail(c, VP_ERR_BAD_ARG, "at most %d partial images", MAX_PARTIALS);
    BlendArgs a{};
    a.n = n;
    for (int i = 0; i < n; ++i) { a.img[i] = (const float4*)d_partials[i]; a.kind[i] = kinds[i]; }
    if (npix == 0) return VP_OK;
    hipLaunchKernelGGL(k_blend, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, c->stream, a, npix, (float4*)d_out);
    VP_HIP(hipGetLastError());
    return VP_OK;
}

int launch_composite(vp_ctx* c, const float* d_particles, float* d_scene)
{
    const size_t npix = (size_t)c->cfg.width * c->cfg.height;
    hipLaunchKernelGGL(k_composite, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, c->stream,
                       (const float4*)d_particles, (float4*)d_scene, npix);
    VP_HIP(hipGetLastError());
    return VP_OK;
}
