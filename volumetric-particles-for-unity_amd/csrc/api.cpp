// api.cpp -- the C ABI of include/vpfx.h: argument checking, device-memory ownership, stage sequencing.
// No compute happens on the host beyond the per-frame uniforms of host_logic.cpp; there is no CPU fallback.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "vpfx_internal.h"

thread_local std::string g_vp_create_error;

int vp_fail(vp_ctx* c, int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf; else g_vp_create_error = buf;
    return code;
}

namespace {

template <typename T>
int dev_alloc(vp_ctx* c, T** p, size_t count)
{
    *p = nullptr;
    if (count == 0) count = 1;
    VP_HIP(hipMalloc((void**)p, count * sizeof(T)));
    return VP_OK;
}

int ensure_device(vp_ctx* c)
{
    VP_HIP(hipSetDevice(c->device));
    return VP_OK;
}

// The chained fill's watchdog flag (a unit that waited ~seconds for its column's light sets it and carries on with whatever it read: an
// error the caller sees instead of a hung GPU).  A set flag invalidates the fill: the bricks must not be composited.
int check_chain_error(vp_ctx* c)
{
    if (c->h_chain_err && *(volatile int*)c->h_chain_err) {
        *(volatile int*)c->h_chain_err = 0;
        c->filled = c->local_done = false;
        return vp_fail(c, VP_ERR_HIP, "fill: a light hand-off between metavoxel units timed out (results of the last fill are invalid)");
    }
    return VP_OK;
}

// Stream sync of the host-facing entry points + the watchdog flag.
int stream_sync(vp_ctx* c)
{
    VP_HIP(hipStreamSynchronize(c->stream));
    return check_chain_error(c);
}

size_t lightmap_elems(const vp_ctx* c) { return (size_t)c->g.Nx * c->g.nv * c->g.Ny * c->g.nv; }
size_t image_elems(const vp_ctx* c) { return (size_t)c->cfg.width * c->cfg.height * 4; }
size_t nv3(const vp_ctx* c) { return (size_t)c->g.nv * c->g.nv * c->g.nv; }

int ensure_bricks(vp_ctx* c, bool need_scratch)
{
    const size_t need = (size_t)c->h_meta.occupied;
    if (need > c->brick_cap) {
        if (c->d_bricks) VP_HIP(hipFree(c->d_bricks));
        c->d_bricks = nullptr;
        c->brick_cap = 0;
        size_t cap = need + need / 8 + 16;               // head-room so that a slowly growing cloud does not reallocate every frame
        if (hipMalloc((void**)&c->d_bricks, cap * nv3(c) * sizeof(uint2)) != hipSuccess) {
            (void)hipGetLastError();
            c->d_bricks = nullptr;
            cap = need;                                    // ... but an exact fit beats failing (config 5 is 157 GiB of bricks)
            VP_HIP(hipMalloc((void**)&c->d_bricks, cap * nv3(c) * sizeof(uint2)));
        }
        c->brick_cap = cap;
    }
    if (need_scratch && need > c->dens_cap) {
        if (c->d_dens_ao) VP_HIP(hipFree(c->d_dens_ao));
        c->d_dens_ao = nullptr;
        c->dens_cap = 0;
        const size_t cap = need + need / 8 + 16;
        VP_HIP(hipMalloc((void**)&c->d_dens_ao, cap * nv3(c) * sizeof(float2)));
        c->dens_cap = cap;
    }
    return VP_OK;
}

// upload the fill pass inputs that live behind pointers (cubemap, light depth map) and build the uniforms
int stage_fill_inputs(vp_ctx* c, const vp_fill_params* p)
{
    // netDisplacement = D * raw + (1 - D) must stay >= 0 (Fill.shader:119-126; the ao max and the smoothstep edges rely on it):
    // the reference's slider is [0, 1] (scene:8103-8111).  Out-of-range inputs are refused, not silently mis-shaded.
    if (!(p->displacement_scale >= 0.f && p->displacement_scale <= 1.f))
        return vp_fail(c, VP_ERR_BAD_ARG, "vp_fill: displacement_scale %g outside [0, 1]", (double)p->displacement_scale);
    if (p->cubemap) {
        const int S = p->cubemap_size;
        if (S < 1 || S > 1024) return vp_fail(c, VP_ERR_BAD_ARG, "cubemap_size %d out of range", S);
        if (p->cubemap_format != VP_CUBEMAP_F32 && p->cubemap_format != VP_CUBEMAP_R8)
            return vp_fail(c, VP_ERR_BAD_ARG, "cubemap_format %d (VP_CUBEMAP_F32 or VP_CUBEMAP_R8)", p->cubemap_format);
        if (S != c->cubeS) {
            if (c->d_cubequads) VP_HIP(hipFree(c->d_cubequads));
            c->d_cubequads = nullptr;
            c->cubeS = 0;                                      // no table resident until the new one is allocated AND built
            VP_HIP(hipMalloc((void**)&c->d_cubequads, (size_t)6 * (S + 1) * (S + 2) * sizeof(float2) + 16));
        }
        c->cubeS = 0;                                          // the resident table is being overwritten: invalid until it is rebuilt
        void* d_cube = nullptr;
        const size_t bytes = (size_t)6 * S * S * (p->cubemap_format == VP_CUBEMAP_R8 ? 1 : sizeof(float));
        VP_HIP(hipMalloc(&d_cube, bytes));
        int bad = 0;
        int* d_bad = c->d_onecol + 1;
        hipError_t e = hipMemsetAsync(d_bad, 0, sizeof(int), c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d_cube, p->cubemap, bytes, hipMemcpyHostToDevice, c->stream);
        int rc = VP_OK;
        if (e == hipSuccess) rc = launch_build_cubequads(c, d_cube, p->cubemap_format, S, d_bad);
        // R8 maps that fit LDS (6 (S+2)^2 bytes <= 160 KB, i.e. S <= 162) also get the padded byte table of the persistent LDS fill
        c->cube_u8_S = 0;
        if (e == hipSuccess && !rc && p->cubemap_format == VP_CUBEMAP_R8 && cube_u8_bytes(S) <= (size_t)160 * 1024 - 1024 /* minus k_fill_lds static LDS */) {
            const size_t need = cube_u8_bytes(S);
            if (need > c->cube_u8_cap) {
                if (c->d_cube_u8) (void)hipFree(c->d_cube_u8);
                c->d_cube_u8 = nullptr; c->cube_u8_cap = 0;
                if (hipMalloc((void**)&c->d_cube_u8, need) == hipSuccess) c->cube_u8_cap = need; else (void)hipGetLastError();
            }
            if (c->d_cube_u8 && launch_build_cube_u8(c, d_cube, S) == VP_OK) c->cube_u8_S = S;     // failure here only loses the fast path
        }
        if (e == hipSuccess && !rc) e = hipMemcpyAsync(&bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, c->stream);
        hipError_t e2 = hipStreamSynchronize(c->stream);       // the caller's cubemap pointer is not retained
        (void)hipFree(d_cube);
        if (e != hipSuccess || e2 != hipSuccess) return vp_fail(c, VP_ERR_HIP, "cubemap upload failed");
        if (rc) return rc;
        if (bad) return vp_fail(c, VP_ERR_BAD_ARG, "vp_fill: displacement cubemap has texels outside [0, 1] (or NaN)");
        c->cubeS = S;
    } else if (!c->d_cubequads || c->cubeS == 0) {
        return vp_fail(c, VP_ERR_BAD_ARG, "vp_fill: no displacement cubemap given and none resident");
    }
    if (p->light_depth_map) {
        if (!c->d_depthmap) { int rc = dev_alloc(c, &c->d_depthmap, lightmap_elems(c)); if (rc) return rc; }
        VP_HIP(hipMemcpyAsync(c->d_depthmap, p->light_depth_map, lightmap_elems(c) * sizeof(float), hipMemcpyHostToDevice, c->stream));
        { int rcs = stream_sync(c); if (rcs) return rcs; }
        c->have_depthmap = true;
        c->light_depth_gen = 0;                                 // the buffer holds the caller's map now
    } else if (c->n_occluders > 0) {
        // no map given but occluder solids are set: render the light depth map on the GPU (VPR.cs:184) -- unless the buffer already holds exactly that
        if (!c->d_depthmap) { int rc = dev_alloc(c, &c->d_depthmap, lightmap_elems(c)); if (rc) return rc; c->light_depth_gen = 0; }
        const float planes[3] = {p->light_near, p->light_far, p->light_cam_distance};
        if (c->light_depth_gen != c->occl_gen || c->light_depth_frame != c->frame_gen || memcmp(planes, c->light_depth_planes, sizeof planes) != 0) {
            int rc = launch_light_depth(c, p->light_near, p->light_far, p->light_cam_distance, c->d_depthmap); if (rc) return rc;
            c->light_depth_gen = c->occl_gen; c->light_depth_frame = c->frame_gen; memcpy(c->light_depth_planes, planes, sizeof planes);
        }
        c->have_depthmap = true;
    } else {
        c->have_depthmap = false;                               // NULL = no occluders (depth 1.0 everywhere)
    }
    hl_build_fill_consts(c, p);
    return VP_OK;
}

// the eye depth rendered from the solids: once per (camera, solids); see vp_ctx::eye_depth_gen
int ensure_eye_depth(vp_ctx* c, const vp_camera* cam)
{
    if (!c->d_scene_depth) { int rc = dev_alloc(c, &c->d_scene_depth, (size_t)c->cfg.width * c->cfg.height); if (rc) return rc; c->eye_depth_gen = 0; }
    if (c->eye_depth_gen == c->occl_gen && memcmp(cam, &c->eye_depth_cam, sizeof *cam) == 0) return VP_OK;
    int rc = launch_scene_depth(c, cam, c->d_scene_depth); if (rc) return rc;
    c->eye_depth_gen = c->occl_gen; c->eye_depth_cam = *cam;
    return VP_OK;
}

int check_fill_ready(vp_ctx* c, const char* who)
{
    if (!c->have_frame) return vp_fail(c, VP_ERR_STATE, "%s before vp_set_frame", who);
    if (!c->binned) return vp_fail(c, VP_ERR_STATE, "%s before vp_bin", who);
    return VP_OK;
}

int stage_raymarch(vp_ctx* c, const vp_camera* cam, const vp_raymarch_params* rp, RmConsts* k)
{
    if (!cam || !rp) return vp_fail(c, VP_ERR_BAD_ARG, "null camera / params");
    { int rce = check_chain_error(c); if (rce) return rce; }   // a fill that already reported a timed-out hand-off must not be composited
    if (!c->filled) return vp_fail(c, VP_ERR_STATE, "vp_raymarch before vp_fill");
    if (rp->steps_per_mv < 1 || rp->soft_distance < 1) return vp_fail(c, VP_ERR_BAD_ARG, "steps_per_mv and soft_distance must be >= 1");
    if ((rp->flags & VP_RM_SHOW_DRAW_ORDER) && (c->g.z0 != 0 || c->g.z1 != c->g.Nz))
        return vp_fail(c, VP_ERR_UNSUPPORTED, "VP_RM_SHOW_DRAW_ORDER needs a whole-grid context (mvCount runs over every slab)");
    hl_build_rm_consts(c, cam, rp, k);
    // (the draw order of the (yy, xx) columns, VPR.cs:613-632, is ranked on the device by k_rm_prepare: no per-frame copy command)
    if (rp->scene_depth) {
        if (!c->d_scene_depth) { int rc = dev_alloc(c, &c->d_scene_depth, (size_t)c->cfg.width * c->cfg.height); if (rc) return rc; }
        VP_HIP(hipMemcpyAsync(c->d_scene_depth, rp->scene_depth, (size_t)c->cfg.width * c->cfg.height * sizeof(float),
                              hipMemcpyHostToDevice, c->stream));
        c->eye_depth_gen = 0;                                   // the buffer holds the caller's depth now
    } else if (c->n_occluders > 0) {
        // no depth buffer given but occluder solids are set: render the eye depth on the GPU (VPR.cs:204) -- unless the buffer already holds it for this camera
        int rc = ensure_eye_depth(c, cam); if (rc) return rc;
    }
    // scene_depth is pageable: the async copy above has already consumed it on return
    return VP_OK;
}

}  // namespace

// the single-device building blocks the fan-out (multi.cpp) runs on its slab contexts
int api_stream_sync(vp_ctx* c) { return stream_sync(c); }
int api_stage_fill_inputs(vp_ctx* c, const vp_fill_params* p) { return stage_fill_inputs(c, p); }
int api_ensure_bricks(vp_ctx* c, bool need_scratch) { return ensure_bricks(c, need_scratch); }
int api_stage_raymarch(vp_ctx* c, const vp_camera* cam, const vp_raymarch_params* rp, RmConsts* k) { return stage_raymarch(c, cam, rp, k); }
int api_set_slab(vp_ctx* c, int z0, int z1)
{
    if (z0 < 0 || z1 > c->g.Nz || z0 >= z1) return vp_fail(c, VP_ERR_BAD_ARG, "bad slab [%d,%d)", z0, z1);
    c->cfg.slab_z0 = z0; c->cfg.slab_z1 = z1;
    c->g.z0 = z0; c->g.z1 = z1;
    c->binned = c->filled = c->local_done = c->fill_begun = false;
    return VP_OK;
}

#define VP_NO_FANOUT(c, what)                                                                                                   \
    do { if ((c)->multi) return vp_fail((c), VP_ERR_UNSUPPORTED, what ": not available on a fan-out (multi-GPU) context"); } while (0)

// --------------------------------------------------------------------------------------------------
VP_EXPORT int vp_abi_version(void) { return VPFX_ABI_VERSION; }

VP_EXPORT const char* vp_last_error(const vp_ctx* c) { return c ? c->err.c_str() : g_vp_create_error.c_str(); }

VP_EXPORT int vp_create(const vp_config* cfg, vp_ctx** out)
{
    if (!cfg || !out) return vp_fail(nullptr, VP_ERR_BAD_ARG, "vp_create: null argument");
    *out = nullptr;
    if (cfg->num_devices < 0 || cfg->num_devices > VP_MAX_LOCAL_DEVICES || cfg->world_size < 0 || cfg->world_size > VP_MAX_RANKS)
        return vp_fail(nullptr, VP_ERR_BAD_ARG, "vp_create: num_devices %d / world_size %d out of range", cfg->num_devices, cfg->world_size);
    if (cfg->num_devices > 1 || cfg->world_size > 1 || (cfg->multi_flags & VP_MULTI_FORCE)) return multi_create(cfg, out);
    if (cfg->num_devices == 1) {                              // a one-entry device list is just that device
        vp_config one = *cfg;
        one.device = cfg->devices[0];
        return vp_create_single(&one, out);
    }
    return vp_create_single(cfg, out);
}

int vp_create_single(const vp_config* cfg, vp_ctx** out)
{
    vp_ctx* c = nullptr;
    *out = nullptr;
    const int nv = cfg->num_voxels;
    if (cfg->num_mv[0] < 1 || cfg->num_mv[1] < 1 || cfg->num_mv[2] < 1 || !(cfg->mv_scale > 0.f) || cfg->num_border < 0 ||
        2 * cfg->num_border >= nv || cfg->width < 1 || cfg->height < 1)
        return vp_fail(nullptr, VP_ERR_BAD_ARG, "vp_create: bad grid/screen configuration");
    // Any voxel count the reference's inspector field can hold (VPR.cs:84; its shader's column array stops at NUM_VOXELS = 32, Fill.shader:16,
    // libvpfx at 64): 16 / 32 / 64 run kernels with the count as a compile-time constant, every other value the run-time-nv kernels
    // (fill_generic.hip, raymarch_generic.hip).
    if (nv < 2 || nv > 64)
        return vp_fail(nullptr, VP_ERR_UNSUPPORTED, "vp_create: num_voxels %d outside [2, 64]", nv);
    if ((cfg->reserved[0] != 0 && cfg->reserved[0] != 1) || (cfg->reserved[1] != 0 && cfg->reserved[1] != 1) || cfg->reserved[2] != 0)
        return vp_fail(nullptr, VP_ERR_BAD_ARG, "vp_create: vp_config.reserved = {%d, %d, %d} (0 or the documented switches; an uninitialised struct?)",
                       cfg->reserved[0], cfg->reserved[1], cfg->reserved[2]);
    if ((size_t)cfg->num_mv[0] * cfg->num_mv[1] * cfg->num_mv[2] > ((size_t)1 << 28))
        return vp_fail(nullptr, VP_ERR_UNSUPPORTED, "vp_create: grid too large");
    int z0 = cfg->slab_z0, z1 = cfg->slab_z1;
    if (z0 == 0 && z1 == 0) z1 = cfg->num_mv[2];
    if (z0 < 0 || z1 > cfg->num_mv[2] || z0 >= z1) return vp_fail(nullptr, VP_ERR_BAD_ARG, "vp_create: bad slab [%d,%d)", z0, z1);

    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev < 1)
        return vp_fail(nullptr, VP_ERR_NO_DEVICE, "vp_create: no HIP device (%s); libvpfx has no CPU fallback",
                       e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    int dev = cfg->device;
    if (dev < 0) { if (hipGetDevice(&dev) != hipSuccess) dev = 0; }
    if (dev >= ndev) return vp_fail(nullptr, VP_ERR_BAD_ARG, "vp_create: device %d of %d", dev, ndev);

    c = new (std::nothrow) vp_ctx();
    if (!c) return vp_fail(nullptr, VP_ERR_OOM, "vp_create: host allocation failed");
    c->cfg = *cfg;
    c->device = dev;
    // test hooks are honoured only by contexts that opt in through vp_config (VP_MULTI_TEST_HOOKS): a stray environment variable never
    // changes what a production context does
    if (cfg->multi_flags & VP_MULTI_TEST_HOOKS) { const char* hook = getenv("VPFX_TEST_CHAIN_TIMEOUT"); c->test_chain_timeout = hook && hook[0] == '1'; }
#if VPFX_AB    // measurement switches of A/B builds (make EXTRA=-DVPFX_AB=1); the shipped library never reads them
    { const char* sw = getenv("VPFX_NO_ZPROFILE"); c->no_zprofile = sw && sw[0] == '1'; }
    { const char* sw = getenv("VPFX_RM_FLAT"); c->rm_flat = sw && sw[0] == '1'; }
    { const char* sw = getenv("VPFX_RM_XCD_AFFINE"); c->rm_xcd_affine = sw && sw[0] == '1'; }
#endif
    c->n3 = (size_t)cfg->num_mv[0] * cfg->num_mv[1] * cfg->num_mv[2];
    // identity frame until vp_set_frame
    memset(c->L, 0, sizeof c->L); c->L[0] = c->L[5] = c->L[10] = c->L[15] = 1.f;
    c->h_mvPos = (float*)malloc(c->n3 * 3 * sizeof(float));
    int rc = VP_OK;
    auto fail = [&](int code) { g_vp_create_error = c->err; vp_destroy_single(c); return code; };
    if (!c->h_mvPos) { c->err = "vp_create: host allocation failed"; return fail(VP_ERR_OOM); }
    hl_build_grid(c);
    if ((rc = ensure_device(c))) return fail(rc);
    if (hipDeviceGetAttribute(&c->num_cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c->num_cus < 1) c->num_cus = 256;
    const size_t nxy = (size_t)cfg->num_mv[0] * cfg->num_mv[1];
    if ((rc = dev_alloc(c, &c->d_mvPos, c->n3 * 3)) || (rc = dev_alloc(c, &c->d_count, c->n3)) ||
        (rc = dev_alloc(c, &c->d_offsets, c->n3 + 1)) || (rc = dev_alloc(c, &c->d_cursor, c->n3)) ||
        (rc = dev_alloc(c, &c->d_brick_index, c->n3)) || (rc = dev_alloc(c, &c->d_occ_list, c->n3)) ||
        (rc = dev_alloc(c, &c->d_onecol, 2)) || (rc = dev_alloc(c, &c->d_work_counter, 1)) || (rc = dev_alloc(c, &c->d_meta, 1)) ||
        (rc = dev_alloc(c, &c->d_ord, c->n3)) || (rc = dev_alloc(c, &c->d_colcount, nxy)) || (rc = dev_alloc(c, &c->d_chain, lightmap_elems(c))) ||
        (rc = dev_alloc(c, (int4**)&c->d_scan_totals, (c->n3 + 1023) / 1024 + 1)) ||
        (rc = dev_alloc(c, &c->d_lightmap, lightmap_elems(c))) || (rc = dev_alloc(c, &c->d_rank, nxy)) ||
        (rc = dev_alloc(c, &c->d_image, image_elems(c))) || (rc = dev_alloc(c, &c->d_samples, 1)) ||
        (rc = dev_alloc(c, &c->d_zsamples, (size_t)VPFX_ZPROF_COPIES * cfg->num_mv[2])) ||
        (rc = dev_alloc(c, &c->d_tile_order, (size_t)rm_order_ints(rm_num_super_tiles(cfg->width, cfg->height)) + rm_num_super_tiles(cfg->width, cfg->height) + 8)))
        return fail(rc);
#if VPFX_AB
    {
        // Hilbert order of the super-tile grid (k_tile_regions cuts it into one compact screen region per XCD): cells of the enclosing
        // 2^k x 2^k square in curve order, those outside the grid skipped
        const int sgx = rm_super_tiles_x(cfg->width), sgy = rm_super_tiles_y(cfg->height), ns = sgx * sgy;
        int side = 1; while (side < sgx || side < sgy) side <<= 1;
        std::vector<int> curve; curve.reserve(ns);
        for (long long d = 0; d < (long long)side * side; ++d) {
            long long tt = d; int x = 0, y = 0;
            for (int sdim = 1; sdim < side; sdim <<= 1) {
                const int rx = 1 & (int)(tt / 2), ry = 1 & ((int)tt ^ rx);
                if (ry == 0) { if (rx == 1) { x = sdim - 1 - x; y = sdim - 1 - y; } const int tmp = x; x = y; y = tmp; }
                x += sdim * rx; y += sdim * ry; tt /= 4;
            }
            if (x < sgx && y < sgy) curve.push_back(y * sgx + x);
        }
        if ((int)curve.size() == ns && (rc = dev_alloc(c, &c->d_tile_curve, (size_t)ns)) == VP_OK) {
            if (hipMemcpy(c->d_tile_curve, curve.data(), (size_t)ns * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) { c->err = "hipMemcpy failed"; return fail(VP_ERR_HIP); }
        } else if (rc) return fail(rc);
    }
#endif
    if (hipHostMalloc((void**)&c->h_meta_host, 2 * sizeof(DevMeta), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
        hipHostGetDevicePointer((void**)&c->d_meta_host, c->h_meta_host, 0) != hipSuccess) { c->err = "hipHostMalloc failed"; return fail(VP_ERR_HIP); }
    c->h_meta_host[0] = c->h_meta_host[1] = DevMeta{};        // [0] the totals, [1].occupied = the sequence word the scan writes behind them
    if (hipHostMalloc((void**)&c->h_chain_err, sizeof(int), hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void**)&c->d_chain_err, c->h_chain_err, 0) != hipSuccess) { c->err = "hipHostMalloc failed"; return fail(VP_ERR_HIP); }
    *c->h_chain_err = 0;
    if (hipMemset(c->d_chain, 0, lightmap_elems(c) * sizeof(unsigned long long)) != hipSuccess) { c->err = "hipMemset failed"; return fail(VP_ERR_HIP); }
    for (int s = 0; s < 4; ++s)
        for (int j = 0; j < 2; ++j)
            if (hipEventCreate(&c->ev[s][j]) != hipSuccess) { c->err = "hipEventCreate failed"; return fail(VP_ERR_HIP); }
    *out = c;
    return VP_OK;
}

VP_EXPORT void vp_destroy(vp_ctx* c)
{
    if (!c) return;
    if (c->multi) { multi_destroy(c); return; }
    vp_destroy_single(c);
}

void vp_destroy_single(vp_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream); else (void)hipDeviceSynchronize();
    void* dev[] = {c->d_mvPos, c->d_raw, c->d_ws, c->d_rec, c->d_count, c->d_offsets, c->d_cursor, c->d_brick_index,
                   c->d_occ_list, c->d_ids_tmp, c->d_ids, c->d_onecol, c->d_work_counter, c->d_ord, c->d_colcount, c->d_chain, c->d_cube_u8, c->d_meta, c->d_scan_totals, c->d_bricks, c->d_dens_ao,
                   c->d_lightmap, c->d_cubequads, c->d_depthmap, c->d_occluders, c->d_mvtrans, c->d_brick_hit, c->d_rank, c->d_tile_order, c->d_tile_curve, c->d_image, c->d_scene_depth, c->d_samples, c->d_zsamples, c->d_cellinfo, c->d_occmask};
    for (void* p : dev) if (p) (void)hipFree(p);
    if (c->h_chain_err) (void)hipHostFree(c->h_chain_err);
    if (c->h_meta_host) (void)hipHostFree(c->h_meta_host);
    if (c->copy_stream) { (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamDestroy(c->copy_stream); }
    if (c->ev_image_ready) (void)hipEventDestroy(c->ev_image_ready);
    if (c->ev_image_copied) (void)hipEventDestroy(c->ev_image_copied);
    for (int s = 0; s < 4; ++s) for (int j = 0; j < 2; ++j) if (c->ev[s][j]) (void)hipEventDestroy(c->ev[s][j]);
    free(c->h_mvPos);
    delete c;
}

VP_EXPORT int vp_pin_host_buffer(vp_ctx* c, void* ptr, uint64_t bytes)
{
    if (!c) return VP_ERR_BAD_ARG;
    if (!ptr || !bytes) return vp_fail(c, VP_ERR_BAD_ARG, "vp_pin_host_buffer: null buffer");
    if (c->multi) { VP_HIP(hipHostRegister(ptr, (size_t)bytes, hipHostRegisterPortable)); return VP_OK; }     // visible to every device
    int rc = ensure_device(c); if (rc) return rc;
    VP_HIP(hipHostRegister(ptr, (size_t)bytes, hipHostRegisterDefault));
    return VP_OK;
}

VP_EXPORT int vp_unpin_host_buffer(vp_ctx* c, void* ptr)
{
    if (!c) return VP_ERR_BAD_ARG;
    if (!ptr) return vp_fail(c, VP_ERR_BAD_ARG, "vp_unpin_host_buffer: null buffer");
    if (c->multi) { int rcm = multi_sync(c); if (rcm) return rcm; VP_HIP(hipHostUnregister(ptr)); return VP_OK; }
    int rc = ensure_device(c); if (rc) return rc;
    if (c->stream) (void)hipStreamSynchronize(c->stream); else (void)hipDeviceSynchronize();
    // a vp_raymarch_async read-back runs on the copy stream, which the caller's stream does not wait for: the buffer must not lose its page
    // lock (and be freed by the caller) under a copy in flight
    if (c->image_copy_pending) { VP_HIP(hipEventSynchronize(c->ev_image_copied)); c->image_copy_pending = false; }
    VP_HIP(hipHostUnregister(ptr));
    return VP_OK;
}

VP_EXPORT int vp_set_stream(vp_ctx* c, void* hip_stream)
{
    if (!c) return VP_ERR_BAD_ARG;
    VP_NO_FANOUT(c, "vp_set_stream (a fan-out context owns one stream per device)");
    c->stream = (hipStream_t)hip_stream;
    return VP_OK;
}

VP_EXPORT int vp_sync(vp_ctx* c)
{
    if (!c) return VP_ERR_BAD_ARG;
    if (c->multi) return multi_sync(c);
    int rc = ensure_device(c); if (rc) return rc;
    { int rcs = stream_sync(c); if (rcs) return rcs; }
    if (c->image_copy_pending) { VP_HIP(hipEventSynchronize(c->ev_image_copied)); c->image_copy_pending = false; }
    return VP_OK;
}

VP_EXPORT int vp_set_frame(vp_ctx* c, const float light_to_world[16], const float grid_center[3])
{
    if (!c) return VP_ERR_BAD_ARG;
    if (!light_to_world || !grid_center) return vp_fail(c, VP_ERR_BAD_ARG, "vp_set_frame: null argument");
    if (c->multi) return multi_set_frame(c, light_to_world, grid_center);
    int rc = ensure_device(c); if (rc) return rc;
    memcpy(c->L, light_to_world, sizeof c->L);
    memcpy(c->gc, grid_center, sizeof c->gc);
    hl_build_grid(c);
    VP_HIP(hipMemcpyAsync(c->d_mvPos, c->h_mvPos, c->n3 * 3 * sizeof(float), hipMemcpyHostToDevice, c->stream));
    { int rcs = stream_sync(c); if (rcs) return rcs; }
    c->have_frame = true;
    c->binned = c->filled = c->local_done = false;
    if (++c->frame_gen == 0) c->frame_gen = 1;
    return VP_OK;
}

VP_EXPORT int vp_upload_particles(vp_ctx* c, const void* particles, int32_t count, const vp_particle_layout* lay,
                                  const float psys_local_to_world[16])
{
    if (!c) return VP_ERR_BAD_ARG;
    if (c->multi) return multi_upload_particles(c, particles, count, lay, psys_local_to_world);
    return api_upload_particles(c, particles, count, lay, psys_local_to_world, true);
}

int api_upload_particles(vp_ctx* c, const void* particles, int32_t count, const vp_particle_layout* lay, const float* psys_local_to_world, bool sync)
{
    if ((!particles && count > 0) || count < 0 || !lay || !psys_local_to_world)
        return vp_fail(c, VP_ERR_BAD_ARG, "vp_upload_particles: null/negative argument");
    const int32_t offs[5] = {lay->off_position + 8, lay->off_size, lay->off_rotation, lay->off_lifetime, lay->off_start_lifetime};
    if (lay->stride < 4) return vp_fail(c, VP_ERR_BAD_ARG, "vp_upload_particles: stride %d", lay->stride);
    for (int32_t o : offs)
        if (o < 0 || o + 4 > lay->stride) return vp_fail(c, VP_ERR_BAD_ARG, "vp_upload_particles: field offset outside the record");
    if (lay->off_position < 0) return vp_fail(c, VP_ERR_BAD_ARG, "vp_upload_particles: field offset outside the record");
    int rc = ensure_device(c); if (rc) return rc;
    const size_t bytes = (size_t)count * lay->stride;
    if (bytes > c->raw_cap) {
        if (c->d_raw) VP_HIP(hipFree(c->d_raw));
        c->d_raw = nullptr; c->raw_cap = 0;
        VP_HIP(hipMalloc((void**)&c->d_raw, bytes + bytes / 4 + 256));
        c->raw_cap = bytes + bytes / 4 + 256;
    }
    if (count > c->P_cap) {
        if (c->d_ws) VP_HIP(hipFree(c->d_ws));
        if (c->d_rec) VP_HIP(hipFree(c->d_rec));
        c->d_ws = nullptr; c->d_rec = nullptr; c->P_cap = 0;
        const int cap = count + count / 4 + 64;
        VP_HIP(hipMalloc((void**)&c->d_ws, (size_t)cap * sizeof(float4)));
        VP_HIP(hipMalloc((void**)&c->d_rec, (size_t)cap * 16 * sizeof(float)));
        c->P_cap = cap;
    }
    c->P = count;
    c->lay = *lay;
    hl_build_psys(c, psys_local_to_world);
    if (count > 0) {
        VP_HIP(hipMemcpyAsync(c->d_raw, particles, bytes, hipMemcpyHostToDevice, c->stream));
        rc = launch_extract(c); if (rc) return rc;
        if (sync) { int rcs = stream_sync(c); if (rcs) return rcs; }     // caller's array is not retained past return (the fan-out syncs all devices itself)
    }
    c->have_particles = true;
    c->binned = c->filled = c->local_done = false;
    return VP_OK;
}

VP_EXPORT int vp_bin_resident(vp_ctx* c)
{
    if (!c) return VP_ERR_BAD_ARG;
    if (c->multi) return multi_bin_resident(c);
    if (!c->have_frame) return vp_fail(c, VP_ERR_STATE, "vp_bin before vp_set_frame");
    if (!c->have_particles) return vp_fail(c, VP_ERR_STATE, "vp_bin_resident before vp_upload_particles");
    int rc = ensure_device(c); if (rc) return rc;
    rc = launch_bin(c); if (rc) return rc;
    c->binned = true;
    c->filled = c->local_done = c->fill_begun = false;
    return VP_OK;
}

VP_EXPORT int vp_z_histogram(vp_ctx* c, int64_t* pairs_per_z)
{
    if (!c) return VP_ERR_BAD_ARG;
    if (!pairs_per_z) return vp_fail(c, VP_ERR_BAD_ARG, "vp_z_histogram: null output");
    VP_NO_FANOUT(c, "vp_z_histogram");
    if (!c->have_frame || !c->have_particles) return vp_fail(c, VP_ERR_STATE, "vp_z_histogram needs vp_set_frame and vp_upload_particles");
    int rc = ensure_device(c); if (rc) return rc;
    // d_cursor is free between bins (k_scan rewrites it); Nz <= N^3 ints
    rc = launch_z_histogram(c, c->d_cursor); if (rc) return rc;
    const int nz = c->g.Nz;
    int* h = (int*)malloc((size_t)nz * sizeof(int));
    if (!h) return vp_fail(c, VP_ERR_OOM, "host allocation failed");
    hipError_t e = hipMemcpyAsync(h, c->d_cursor, (size_t)nz * sizeof(int), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    for (int i = 0; i < nz; ++i) pairs_per_z[i] = h[i];
    free(h);
    if (e != hipSuccess) return vp_fail(c, VP_ERR_HIP, "vp_z_histogram: %s", hipGetErrorString(e));
    c->binned = c->filled = c->local_done = false;        // the cursor scratch was reused
    return VP_OK;
}

VP_EXPORT int vp_bin(vp_ctx* c, const void* particles, int32_t count, const vp_particle_layout* lay, const float psys_local_to_world[16])
{
    int rc = vp_upload_particles(c, particles, count, lay, psys_local_to_world);
    if (rc) return rc;
    return vp_bin_resident(c);
}

VP_EXPORT int vp_fill(vp_ctx* c, const vp_fill_params* p)
{
    if (!c) return VP_ERR_BAD_ARG;
    if (!p) return vp_fail(c, VP_ERR_BAD_ARG, "vp_fill: null params");
    if (c->multi) return multi_fill(c, p);
    int rc = check_fill_ready(c, "vp_fill"); if (rc) return rc;
    if ((rc = ensure_device(c)) || (rc = stage_fill_inputs(c, p)) || (rc = ensure_bricks(c, false))) return rc;
    rc = launch_fill(c, 0, nullptr, c->d_lightmap); if (rc) return rc;
    c->filled = true;
    return VP_OK;
}

// ---- the reference's per-metavoxel entry points ----------------------------------------------------------------------------
VP_EXPORT int vp_fill_begin(vp_ctx* c, const vp_fill_params* p)
{
    if (!c) return VP_ERR_BAD_ARG;
    if (!p) return vp_fail(c, VP_ERR_BAD_ARG, "vp_fill_begin: null params");
    VP_NO_FANOUT(c, "vp_fill_begin");
    int rc = check_fill_ready(c, "vp_fill_begin"); if (rc) return rc;
    const size_t cap0 = c->brick_cap;
    if ((rc = ensure_device(c)) || (rc = stage_fill_inputs(c, p)) || (rc = ensure_bricks(c, false))) return rc;
    // a freshly (re)allocated pool holds no textures yet: RenderTexture contents start out cleared.  So does a pool whose storage format
    // changes with this fill (grey z-pair entries <-> RGBA16F, decided by the ambient colour): per-metavoxel fills only rewrite the bricks
    // they are called for, and the ray-march reads every brick in ONE format -- a brick left over in the other one would render as garbage.
    const bool grey = c->fc.grey != 0;
    if (c->brick_cap != cap0 || grey != c->bricks_grey) VP_HIP(hipMemsetAsync(c->d_bricks, 0, c->brick_cap * nv3(c) * sizeof(uint2), c->stream));
    c->bricks_grey = grey;
    // GL.Clear(false, true, Color.red) on lightPropogationTex: 1.0 in the R channel                       VPR.cs:498-499
    rc = launch_fill_value(c, c->d_lightmap, lightmap_elems(c), 1.0f); if (rc) return rc;
    c->fill_begun = true;
    c->filled = false;
    return VP_OK;
}

VP_EXPORT int vp_fill_metavoxel(vp_ctx* c, int32_t xx, int32_t yy, int32_t zz)
{
    if (!c) return VP_ERR_BAD_ARG;
    VP_NO_FANOUT(c, "vp_fill_metavoxel");
    if (!c->fill_begun || !c->binned) return vp_fail(c, VP_ERR_STATE, "vp_fill_metavoxel before vp_fill_begin");
    const GridConsts& g = c->g;
    if (xx < 0 || yy < 0 || zz < 0 || xx >= g.Nx || yy >= g.Ny || zz >= g.Nz) return vp_fail(c, VP_ERR_BAD_ARG, "MV index out of range");
    if (zz < g.z0 || zz >= g.z1) return vp_fail(c, VP_ERR_BAD_ARG, "metavoxel slice %d is outside the owned slab [%d,%d)", zz, g.z0, g.z1);
    int rc = ensure_device(c); if (rc) return rc;
    rc = launch_fill_one(c, xx, yy, zz); if (rc) return rc;      // empty MV: the kernel skips it (VPR.cs:511)
    c->filled = true;
    return VP_OK;
}

VP_EXPORT int vp_clear_particles_rt(vp_ctx* c)
{
    if (!c) return VP_ERR_BAD_ARG;
    VP_NO_FANOUT(c, "vp_clear_particles_rt");
    int rc = ensure_device(c); if (rc) return rc;
    VP_HIP(hipMemsetAsync(c->d_image, 0, image_elems(c) * sizeof(float), c->stream));                     // VPR.cs:171-172
    VP_HIP(hipMemsetAsync(c->d_samples, 0, sizeof(unsigned long long), c->stream));
    return VP_OK;
}

VP_EXPORT int vp_render_metavoxel(vp_ctx* c, const vp_camera* cam, const vp_raymarch_params* rp, int32_t xx, int32_t yy, int32_t zz,
                                  int32_t blend_over, int32_t order_index)
{
    if (!c) return VP_ERR_BAD_ARG;
    if (!cam || !rp) return vp_fail(c, VP_ERR_BAD_ARG, "vp_render_metavoxel: null camera / params");
    VP_NO_FANOUT(c, "vp_render_metavoxel");
    if (!c->filled) return vp_fail(c, VP_ERR_STATE, "vp_render_metavoxel before vp_fill");
    if (rp->steps_per_mv < 1 || rp->soft_distance < 1) return vp_fail(c, VP_ERR_BAD_ARG, "steps_per_mv and soft_distance must be >= 1");
    const GridConsts& g = c->g;
    if (xx < 0 || yy < 0 || zz < 0 || xx >= g.Nx || yy >= g.Ny || zz >= g.Nz) return vp_fail(c, VP_ERR_BAD_ARG, "MV index out of range");
    int rc = ensure_device(c); if (rc) return rc;
    const size_t mi = ((size_t)zz * g.Ny + yy) * g.Nx + xx;
    int bi = -1;
    VP_HIP(hipMemcpyAsync(&bi, c->d_brick_index + mi, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    { int rcs = stream_sync(c); if (rcs) return rcs; }
    if (bi < 0) return VP_OK;                                    // empty / not owned: never submitted (VPR.cs:674, 703)
    RmConsts k;
    hl_build_rm_consts(c, cam, rp, &k);
    float* keep = c->d_scene_depth;
    if (rp->scene_depth) {
        if (!c->d_scene_depth) { rc = dev_alloc(c, &c->d_scene_depth, (size_t)c->cfg.width * c->cfg.height); if (rc) return rc; }
        keep = c->d_scene_depth;
        VP_HIP(hipMemcpyAsync(c->d_scene_depth, rp->scene_depth, (size_t)c->cfg.width * c->cfg.height * sizeof(float),
                              hipMemcpyHostToDevice, c->stream));
        c->eye_depth_gen = 0;
    } else if (c->n_occluders > 0) {
        rc = ensure_eye_depth(c, cam); if (rc) return rc;
        keep = c->d_scene_depth;
    } else {
        c->d_scene_depth = nullptr;
    }
    rc = launch_raymarch_one(c, k, bi, (int)mi, blend_over ? 1 : 0, order_index, c->d_image);
    c->d_scene_depth = keep;
    return rc;
}

VP_EXPORT int vp_read_particles_rt(vp_ctx* c, float* rgba_out)
{
    if (!c) return VP_ERR_BAD_ARG;
    if (!rgba_out) return vp_fail(c, VP_ERR_BAD_ARG, "vp_read_particles_rt: null output");
    VP_NO_FANOUT(c, "vp_read_particles_rt");
    int rc = ensure_device(c); if (rc) return rc;
    VP_HIP(hipMemcpyAsync(rgba_out, c->d_image, image_elems(c) * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    { int rcs = stream_sync(c); if (rcs) return rcs; }
    return VP_OK;
}

VP_EXPORT int vp_fill_local(vp_ctx* c, const vp_fill_params* p, void* d_tau_out)
{
    if (!c) return VP_ERR_BAD_ARG;
    if (!p || !d_tau_out) return vp_fail(c, VP_ERR_BAD_ARG, "vp_fill_local: null argument");
    VP_NO_FANOUT(c, "vp_fill_local");
    int rc = check_fill_ready(c, "vp_fill_local"); if (rc) return rc;
    if ((rc = ensure_device(c)) || (rc = stage_fill_inputs(c, p)) || (rc = ensure_bricks(c, true))) return rc;
    rc = launch_fill(c, 1, nullptr, (float*)d_tau_out); if (rc) return rc;
    c->local_done = true;
    c->filled = false;
    return VP_OK;
}

VP_EXPORT int vp_fill_finish(vp_ctx* c, const void* d_light_in)
{
    if (!c) return VP_ERR_BAD_ARG;
    VP_NO_FANOUT(c, "vp_fill_finish");
    if (!c->local_done) return vp_fail(c, VP_ERR_STATE, "vp_fill_finish before vp_fill_local");
    int rc = ensure_device(c); if (rc) return rc;
    rc = launch_fill(c, 2, (const float*)d_light_in, c->d_lightmap); if (rc) return rc;
    c->filled = true;
    return VP_OK;
}

VP_EXPORT int vp_fill_finish_gathered(vp_ctx* c, const void* d_tau_all, int32_t rank, int32_t world)
{
    if (!c) return VP_ERR_BAD_ARG;
    if (!d_tau_all || world < 1 || rank < 0 || rank >= world) return vp_fail(c, VP_ERR_BAD_ARG, "vp_fill_finish_gathered: bad argument");
    VP_NO_FANOUT(c, "vp_fill_finish_gathered");
    if (!c->local_done) return vp_fail(c, VP_ERR_STATE, "vp_fill_finish_gathered before vp_fill_local");
    int rc = ensure_device(c); if (rc) return rc;
    c->finish_tau_all = rank > 0 ? (const float*)d_tau_all : nullptr;      // rank 0: nothing nearer the light, T_in = 1
    c->finish_n_before = rank;
    rc = launch_fill(c, 2, nullptr, c->d_lightmap);
    c->finish_tau_all = nullptr; c->finish_n_before = 0;
    if (rc) return rc;
    c->filled = true;
    return VP_OK;
}

VP_EXPORT int vp_raymarch_device(vp_ctx* c, const vp_camera* cam, const vp_raymarch_params* rp, void* d_rgba_out)
{
    if (!c) return VP_ERR_BAD_ARG;
    if (!d_rgba_out) return vp_fail(c, VP_ERR_BAD_ARG, "vp_raymarch_device: null output");
    if (c->multi) return multi_raymarch(c, cam, rp, nullptr, d_rgba_out);
    int rc = ensure_device(c); if (rc) return rc;
    RmConsts k;
    rc = stage_raymarch(c, cam, rp, &k); if (rc) return rc;
    float* keep = c->d_scene_depth;
    if (!rp->scene_depth && c->n_occluders == 0) c->d_scene_depth = nullptr;
    rc = launch_raymarch(c, k, (float*)d_rgba_out, nullptr);
    c->d_scene_depth = keep;
    return rc;
}

VP_EXPORT int vp_raymarch(vp_ctx* c, const vp_camera* cam, const vp_raymarch_params* rp, float* rgba_out)
{
    if (!c) return VP_ERR_BAD_ARG;
    if (c->multi) return multi_raymarch(c, cam, rp, rgba_out, nullptr);   // (rgba_out may be NULL on processes that do not hold rank 0)
    if (!rgba_out) return vp_fail(c, VP_ERR_BAD_ARG, "vp_raymarch: null output");
    if (c->image_copy_pending) VP_HIP(hipStreamWaitEvent(c->stream, c->ev_image_copied, 0));    // an earlier vp_raymarch_async still reads d_image
    int rc = vp_raymarch_device(c, cam, rp, c->d_image); if (rc) return rc;
    VP_HIP(hipMemcpyAsync(rgba_out, c->d_image, image_elems(c) * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    { int rcs = stream_sync(c); if (rcs) return rcs; }
    return VP_OK;
}

// vp_raymarch without the wait: the ray-march is queued, a second stream copies the image to the host as soon as it is complete, and the
// call returns.  The next frame's vp_bin* / vp_fill run beside that copy (33 MB at 1080p, ~1 ms over PCIe); vp_wait_image blocks until the
// image has landed.  One image in flight per context: a later vp_raymarch / vp_raymarch_async waits ON THE DEVICE for the copy before it
// overwrites the context's image.  rgba_out should be page-locked (vp_pin_host_buffer); a pageable buffer is staged by the driver and the
// call then blocks for the copy.  Fan-out contexts: the synchronous vp_raymarch (the image is assembled on rank 0's stream anyway).
VP_EXPORT int vp_raymarch_async(vp_ctx* c, const vp_camera* cam, const vp_raymarch_params* rp, float* rgba_out)
{
    if (!c) return VP_ERR_BAD_ARG;
    if (c->multi) return multi_raymarch(c, cam, rp, rgba_out, nullptr);
    if (!rgba_out) return vp_fail(c, VP_ERR_BAD_ARG, "vp_raymarch_async: null output");
    int rc = ensure_device(c); if (rc) return rc;
    if (!c->copy_stream) {
        VP_HIP(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
        VP_HIP(hipEventCreateWithFlags(&c->ev_image_ready, hipEventDisableTiming));
        VP_HIP(hipEventCreateWithFlags(&c->ev_image_copied, hipEventDisableTiming));
    }
    if (c->image_copy_pending) VP_HIP(hipStreamWaitEvent(c->stream, c->ev_image_copied, 0));
    rc = vp_raymarch_device(c, cam, rp, c->d_image); if (rc) return rc;
    VP_HIP(hipEventRecord(c->ev_image_ready, c->stream));
    VP_HIP(hipStreamWaitEvent(c->copy_stream, c->ev_image_ready, 0));
    VP_HIP(hipMemcpyAsync(rgba_out, c->d_image, image_elems(c) * sizeof(float), hipMemcpyDeviceToHost, c->copy_stream));
    VP_HIP(hipEventRecord(c->ev_image_copied, c->copy_stream));
    c->image_copy_pending = true;
    return VP_OK;
}

VP_EXPORT int vp_wait_image(vp_ctx* c)
{
    if (!c) return VP_ERR_BAD_ARG;
    if (c->multi || !c->image_copy_pending) return VP_OK;
    int rc = ensure_device(c); if (rc) return rc;
    VP_HIP(hipEventSynchronize(c->ev_image_copied));
    c->image_copy_pending = false;
    return check_chain_error(c);          // (the fill whose image this is may have reported a timed-out hand-off)
}

VP_EXPORT int vp_raymarch_partial_device(vp_ctx* c, const vp_camera* cam, const vp_raymarch_params* rp, void* d_over, void* d_under,
                                         int32_t* phase_mask)
{
    return vp_raymarch_partial_handoff_device(c, cam, rp, d_over, d_under, phase_mask, nullptr, 0, nullptr, nullptr);
}

VP_EXPORT int vp_raymarch_partial_handoff_device(vp_ctx* c, const vp_camera* cam, const vp_raymarch_params* rp, void* d_over, void* d_under,
                                                 int32_t* phase_mask, const void* d_t_in, int32_t n_in, void* d_t_out0, void* d_t_out1)
{
    if (!c) return VP_ERR_BAD_ARG;
    if (!d_over || !d_under) return vp_fail(c, VP_ERR_BAD_ARG, "vp_raymarch_partial_device: null output");
    if (n_in < 0 || n_in > VP_MAX_RANKS || (n_in > 0 && !d_t_in)) return vp_fail(c, VP_ERR_BAD_ARG, "vp_raymarch_partial_handoff_device: bad t_in");
    VP_NO_FANOUT(c, "vp_raymarch_partial_device");
    int rc = ensure_device(c); if (rc) return rc;
    RmConsts k;
    rc = stage_raymarch(c, cam, rp, &k); if (rc) return rc;
    k.partial = 1;
    if (phase_mask) {
        // bit0: the slab has OVER-phase slices (zz <= zBoundary), bit1: UNDER-phase slices
        *phase_mask = (k.z0 <= k.zB ? 1 : 0) | (k.z1 - 1 > k.zB ? 2 : 0);
    }
    float* keep = c->d_scene_depth;
    if (!rp->scene_depth && c->n_occluders == 0) c->d_scene_depth = nullptr;
    RmHandoff ho{};
    ho.t_in = n_in > 0 ? (const uint8_t*)d_t_in : nullptr; ho.n_in = n_in; ho.plane = (size_t)c->cfg.width * c->cfg.height;
    ho.t_out0 = (uint8_t*)d_t_out0; ho.t_out1 = (uint8_t*)d_t_out1; ho.zsamples = c->no_zprofile ? nullptr : c->d_zsamples;
    VP_HIP(hipMemsetAsync(c->d_zsamples, 0, (size_t)VPFX_ZPROF_COPIES * c->g.Nz * sizeof(unsigned), c->stream));
    rc = launch_raymarch(c, k, (float*)d_over, (float*)d_under, &ho);
    c->d_scene_depth = keep;
    return rc;
}

VP_EXPORT int vp_read_zsamples(vp_ctx* c, int64_t* samples_per_z)
{
    if (!c) return VP_ERR_BAD_ARG;
    if (!samples_per_z) return vp_fail(c, VP_ERR_BAD_ARG, "vp_read_zsamples: null output");
    VP_NO_FANOUT(c, "vp_read_zsamples");
    int rc = ensure_device(c); if (rc) return rc;
    long long* tmp = (long long*)malloc((size_t)c->g.Nz * sizeof(long long));
    if (!tmp) return vp_fail(c, VP_ERR_OOM, "host allocation failed");
    rc = api_read_zsamples(c, tmp, true);
    if (!rc) for (int i = 0; i < c->g.Nz; ++i) samples_per_z[i] = tmp[i];        // (a failed read leaves the caller's buffer alone)
    free(tmp);
    return rc;
}

int api_read_zsamples(vp_ctx* c, long long* out, bool sync)
{
    if (sync) { int rcs = stream_sync(c); if (rcs) return rcs; }
    const int nz = c->g.Nz;
    const size_t n = (size_t)VPFX_ZPROF_COPIES * nz;
    unsigned* h = (unsigned*)malloc(n * sizeof(unsigned));
    if (!h) return vp_fail(c, VP_ERR_OOM, "host allocation failed");
    hipError_t e = hipMemcpyAsync(h, c->d_zsamples, n * sizeof(unsigned), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    for (int z = 0; z < nz; ++z) out[z] = 0;
    for (size_t i = 0; i < n && e == hipSuccess; ++i) out[i % nz] += h[i];
    free(h);
    if (e != hipSuccess) return vp_fail(c, VP_ERR_HIP, "zsamples read-back: %s", hipGetErrorString(e));
    return VP_OK;
}

VP_EXPORT int vp_blend_partials_device(vp_ctx* c, const void* const* d_partials, const int32_t* kinds, int32_t n, void* d_rgba_out)
{
    if (!c) return VP_ERR_BAD_ARG;
    if (!d_partials || !kinds || n < 0 || !d_rgba_out) return vp_fail(c, VP_ERR_BAD_ARG, "vp_blend_partials_device: bad argument");
    VP_NO_FANOUT(c, "vp_blend_partials_device");
    int rc = ensure_device(c); if (rc) return rc;
    return launch_blend(c, d_partials, kinds, n, (float*)d_rgba_out, (size_t)c->cfg.width * c->cfg.height);
}

VP_EXPORT int vp_blend_partials_range_device(vp_ctx* c, const void* const* d_partials, const int32_t* kinds, int32_t n, void* d_rgba_out,
                                             int64_t num_pixels)
{
    if (!c) return VP_ERR_BAD_ARG;
    if (!d_partials || !kinds || n < 0 || !d_rgba_out || num_pixels < 0)
        return vp_fail(c, VP_ERR_BAD_ARG, "vp_blend_partials_range_device: bad argument");
    VP_NO_FANOUT(c, "vp_blend_partials_range_device");
    int rc = ensure_device(c); if (rc) return rc;
    return launch_blend(c, d_partials, kinds, n, (float*)d_rgba_out, (size_t)num_pixels);
}

VP_EXPORT int vp_composite_device(vp_ctx* c, const void* d_particles_rgba, void* d_scene_rgba)
{
    if (!c) return VP_ERR_BAD_ARG;
    if (!d_particles_rgba || !d_scene_rgba) return vp_fail(c, VP_ERR_BAD_ARG, "vp_composite_device: null argument");
    if (c->multi) c = multi_owner_of_slice(c, -1);               // the display rank's context (images live on its device)
    if (!c) return VP_ERR_STATE;
    int rc = ensure_device(c); if (rc) return rc;
    return launch_composite(c, (const float*)d_particles_rgba, (float*)d_scene_rgba);
}

VP_EXPORT int vp_z_boundary(vp_ctx* c, const vp_camera* cam, int32_t* zb)
{
    if (!c) return VP_ERR_BAD_ARG;
    if (!cam || !zb) return vp_fail(c, VP_ERR_BAD_ARG, "vp_z_boundary: null argument");
    if (c->multi) c = multi_owner_of_slice(c, -2);               // any local child: the frame is the same everywhere
    if (!c->have_frame) return vp_fail(c, VP_ERR_STATE, "vp_z_boundary before vp_set_frame");
    *zb = hl_z_boundary(c, cam);
    return VP_OK;
}

// ---- scene occluders -----------------------------------------------------------------------------------
VP_EXPORT int vp_set_occluders2(vp_ctx* c, const vp_occluder* solids, int32_t n)
{
    if (!c) return VP_ERR_BAD_ARG;
    if (n < 0 || (n > 0 && !solids)) return vp_fail(c, VP_ERR_BAD_ARG, "vp_set_occluders2: bad argument");
    for (int i = 0; i < n; ++i) {
        if (solids[i].type < VP_OCC_BOX || solids[i].type > VP_OCC_ELLIPSOID) return vp_fail(c, VP_ERR_BAD_ARG, "vp_set_occluders2: unknown solid type");
        if (solids[i].type != VP_OCC_BOX)                              // boxes keep ABI 5's behaviour (a flat box is a plane)
            for (int k = 0; k < 3; ++k)
                if (!(solids[i].half_extent[k] > 0.f)) return vp_fail(c, VP_ERR_BAD_ARG, "vp_set_occluders2: half_extent must be > 0");
    }
    if (c->multi) return multi_set_occluders(c, solids, n);
    int rc = ensure_device(c); if (rc) return rc;
    if (n > c->occluders_cap) {
        if (c->d_occluders) VP_HIP(hipFree(c->d_occluders));
        c->d_occluders = nullptr; c->occluders_cap = 0;
        VP_HIP(hipMalloc((void**)&c->d_occluders, (size_t)(n + 8) * sizeof(vp_occluder)));
        c->occluders_cap = n + 8;
    }
    if (n > 0) {
        VP_HIP(hipMemcpyAsync(c->d_occluders, solids, (size_t)n * sizeof(vp_occluder), hipMemcpyHostToDevice, c->stream));
        { int rcs = stream_sync(c); if (rcs) return rcs; }
    }
    c->n_occluders = n;
    if (++c->occl_gen == 0) c->occl_gen = 1;
    return VP_OK;
}

VP_EXPORT int vp_set_occluders(vp_ctx* c, const vp_obb* boxes, int32_t n)
{
    if (!c) return VP_ERR_BAD_ARG;
    if (n < 0 || (n > 0 && !boxes)) return vp_fail(c, VP_ERR_BAD_ARG, "vp_set_occluders: bad argument");
    vp_occluder* v = n ? (vp_occluder*)calloc((size_t)n, sizeof(vp_occluder)) : nullptr;
    if (n && !v) return vp_fail(c, VP_ERR_OOM, "vp_set_occluders: out of host memory");
    for (int i = 0; i < n; ++i) { memcpy(&v[i], &boxes[i], sizeof(vp_obb)); v[i].type = VP_OCC_BOX; }
    const int rc = vp_set_occluders2(c, v, n);
    free(v);
    return rc;
}

VP_EXPORT int vp_render_light_depth(vp_ctx* c, float light_near, float light_far, float light_cam_distance, float* out)
{
    if (!c) return VP_ERR_BAD_ARG;
    if (!out || !(light_far > light_near)) return vp_fail(c, VP_ERR_BAD_ARG, "vp_render_light_depth: bad argument");
    if (c->multi) c = multi_owner_of_slice(c, -2);
    if (!c->have_frame) return vp_fail(c, VP_ERR_STATE, "vp_render_light_depth before vp_set_frame");
    int rc = ensure_device(c); if (rc) return rc;
    float* d_tmp = nullptr;
    VP_HIP(hipMalloc((void**)&d_tmp, lightmap_elems(c) * sizeof(float)));
    rc = launch_light_depth(c, light_near, light_far, light_cam_distance, d_tmp);
    hipError_t e = hipSuccess;
    if (!rc) e = hipMemcpyAsync(out, d_tmp, lightmap_elems(c) * sizeof(float), hipMemcpyDeviceToHost, c->stream);
    hipError_t e2 = hipStreamSynchronize(c->stream);
    (void)hipFree(d_tmp);
    if (rc) return rc;
    if (e != hipSuccess || e2 != hipSuccess) return vp_fail(c, VP_ERR_HIP, "vp_render_light_depth: copy failed");
    return VP_OK;
}

VP_EXPORT int vp_render_scene_depth(vp_ctx* c, const vp_camera* cam, float* out)
{
    if (!c) return VP_ERR_BAD_ARG;
    if (!cam || !out) return vp_fail(c, VP_ERR_BAD_ARG, "vp_render_scene_depth: null argument");
    if (c->multi) c = multi_owner_of_slice(c, -2);
    int rc = ensure_device(c); if (rc) return rc;
    const size_t n = (size_t)c->cfg.width * c->cfg.height;
    float* d_tmp = nullptr;
    VP_HIP(hipMalloc((void**)&d_tmp, n * sizeof(float)));
    rc = launch_scene_depth(c, cam, d_tmp);
    hipError_t e = hipSuccess;
    if (!rc) e = hipMemcpyAsync(out, d_tmp, n * sizeof(float), hipMemcpyDeviceToHost, c->stream);
    hipError_t e2 = hipStreamSynchronize(c->stream);
    (void)hipFree(d_tmp);
    if (rc) return rc;
    if (e != hipSuccess || e2 != hipSuccess) return vp_fail(c, VP_ERR_HIP, "vp_render_scene_depth: copy failed");
    return VP_OK;
}

// ---- probes ----------------------------------------------------------------------------------------
VP_EXPORT int vp_get_mv_positions(vp_ctx* c, float* out)
{
    if (!c) return VP_ERR_BAD_ARG;
    if (!out) return vp_fail(c, VP_ERR_BAD_ARG, "null output");
    if (c->multi) c = multi_owner_of_slice(c, -2);
    if (!c->have_frame) return vp_fail(c, VP_ERR_STATE, "vp_get_mv_positions before vp_set_frame");
    memcpy(out, c->h_mvPos, c->n3 * 3 * sizeof(float));
    return VP_OK;
}

VP_EXPORT int vp_read_bincounts(vp_ctx* c, int32_t* counts)
{
    if (!c) return VP_ERR_BAD_ARG;
    if (!counts) return vp_fail(c, VP_ERR_BAD_ARG, "null output");
    if (c->multi) return multi_read_bincounts(c, counts);
    if (!c->binned) return vp_fail(c, VP_ERR_STATE, "vp_read_bincounts before vp_bin");
    int rc = ensure_device(c); if (rc) return rc;
    { int rcs = stream_sync(c); if (rcs) return rcs; }
    VP_HIP(hipMemcpy(counts, c->d_count, c->n3 * sizeof(int), hipMemcpyDeviceToHost));
    return VP_OK;
}

VP_EXPORT int vp_read_binlist(vp_ctx* c, int32_t xx, int32_t yy, int32_t zz, int32_t* ids, int32_t cap, int32_t* n)
{
    if (!c) return VP_ERR_BAD_ARG;
    if (!n || (cap > 0 && !ids)) return vp_fail(c, VP_ERR_BAD_ARG, "null output");
    if (c->multi) { vp_ctx* k = multi_owner_of_slice(c, zz); if (!k) return vp_fail(c, VP_ERR_BAD_ARG, "slice %d is not on this process", zz); c = k; }
    if (!c->binned) return vp_fail(c, VP_ERR_STATE, "vp_read_binlist before vp_bin");
    const GridConsts& g = c->g;
    if (xx < 0 || yy < 0 || zz < 0 || xx >= g.Nx || yy >= g.Ny || zz >= g.Nz) return vp_fail(c, VP_ERR_BAD_ARG, "MV index out of range");
    int rc = ensure_device(c); if (rc) return rc;
    { int rcs = stream_sync(c); if (rcs) return rcs; }
    const size_t mi = ((size_t)zz * g.Ny + yy) * g.Nx + xx;
    int off[2];
    VP_HIP(hipMemcpy(off, c->d_offsets + mi, 2 * sizeof(int), hipMemcpyDeviceToHost));
    *n = off[1] - off[0];
    const int m = *n < cap ? *n : cap;
    if (m > 0) VP_HIP(hipMemcpy(ids, c->d_ids + off[0], (size_t)m * sizeof(int), hipMemcpyDeviceToHost));
    return VP_OK;
}

VP_EXPORT int vp_read_brick(vp_ctx* c, int32_t xx, int32_t yy, int32_t zz, uint16_t* half_rgba)
{
    if (!c) return VP_ERR_BAD_ARG;
    if (!half_rgba) return vp_fail(c, VP_ERR_BAD_ARG, "null output");
    if (c->multi) { vp_ctx* k = multi_owner_of_slice(c, zz); if (!k) return vp_fail(c, VP_ERR_BAD_ARG, "slice %d is not on this process", zz); c = k; }
    if (!c->filled) return vp_fail(c, VP_ERR_STATE, "vp_read_brick before vp_fill");
    const GridConsts& g = c->g;
    if (xx < 0 || yy < 0 || zz < 0 || xx >= g.Nx || yy >= g.Ny || zz >= g.Nz) return vp_fail(c, VP_ERR_BAD_ARG, "MV index out of range");
    int rc = ensure_device(c); if (rc) return rc;
    { int rcs = stream_sync(c); if (rcs) return rcs; }
    const size_t mi = ((size_t)zz * g.Ny + yy) * g.Nx + xx;
    int bi = -1;
    VP_HIP(hipMemcpy(&bi, c->d_brick_index + mi, sizeof(int), hipMemcpyDeviceToHost));
    if (bi < 0) return vp_fail(c, VP_ERR_STATE, "metavoxel (%d,%d,%d) is empty / not owned: no brick", xx, yy, zz);
    if (c->bricks_grey) {
        // (luminance, density) storage: expand to the reference's RGBA16F texel (r = g = b = luminance, a = density)
        // an entry is texel(z), texel(z + 1): the first word is this voxel's
        const size_t n = nv3(c), words = 2;
        uint32_t* tmp = (uint32_t*)malloc(n * words * sizeof(uint32_t));
        if (!tmp) return vp_fail(c, VP_ERR_OOM, "host allocation failed");
        hipError_t e = hipMemcpy(tmp, reinterpret_cast<const uint32_t*>(c->d_bricks) + (size_t)bi * n * words, n * words * sizeof(uint32_t),
                                 hipMemcpyDeviceToHost);
        for (size_t i = 0; i < n && e == hipSuccess; ++i) {
            const uint32_t t = tmp[i * words];
            const uint16_t lum = (uint16_t)(t & 0xffffu), den = (uint16_t)(t >> 16);
            half_rgba[4 * i] = lum; half_rgba[4 * i + 1] = lum; half_rgba[4 * i + 2] = lum; half_rgba[4 * i + 3] = den;
        }
        free(tmp);
        if (e != hipSuccess) return vp_fail(c, VP_ERR_HIP, "hipMemcpy failed: %s", hipGetErrorString(e));
        return VP_OK;
    }
    VP_HIP(hipMemcpy(half_rgba, c->d_bricks + (size_t)bi * nv3(c), nv3(c) * sizeof(uint2), hipMemcpyDeviceToHost));
    return VP_OK;
}

VP_EXPORT int vp_read_lightmap(vp_ctx* c, float* out)
{
    if (!c) return VP_ERR_BAD_ARG;
    if (!out) return vp_fail(c, VP_ERR_BAD_ARG, "null output");
    if (c->multi) return multi_read_lightmap(c, out);
    if (!c->filled) return vp_fail(c, VP_ERR_STATE, "vp_read_lightmap before vp_fill");
    int rc = ensure_device(c); if (rc) return rc;
    { int rcs = stream_sync(c); if (rcs) return rcs; }
    VP_HIP(hipMemcpy(out, c->d_lightmap, lightmap_elems(c) * sizeof(float), hipMemcpyDeviceToHost));
    return VP_OK;
}

VP_EXPORT int vp_get_stats(vp_ctx* c, vp_stats* st)
{
    if (!c) return VP_ERR_BAD_ARG;
    if (!st) return vp_fail(c, VP_ERR_BAD_ARG, "null output");
    if (c->multi) return multi_get_stats(c, st);
    int rc = ensure_device(c); if (rc) return rc;
    memset(st, 0, sizeof *st);
    st->particles = c->P;
    if (c->binned) {
        st->occupied_mv = c->h_meta.occupied;
        st->pairs = c->h_meta.pairs;
        st->max_pairs_per_mv = c->h_meta.max_pairs;
        st->voxels_filled = (int64_t)c->h_meta.occupied * (int64_t)nv3(c);
    }
    st->brick_bytes = (int64_t)(c->brick_cap * nv3(c) * sizeof(uint2));      // pool as allocated (8 B/voxel in either storage format)
    st->brick_bytes_per_voxel = 8;
    st->brick_format = c->bricks_grey ? VP_BRICKS_GREY_ZPAIR : VP_BRICKS_RGBA16F;     // storage of the bricks as last filled
    { int rcs = stream_sync(c); if (rcs) return rcs; }
    unsigned long long s = 0;
    VP_HIP(hipMemcpy(&s, c->d_samples, sizeof s, hipMemcpyDeviceToHost));
    st->samples = (int64_t)s;
    if (c->d_brick_hit && c->brick_hit_n > 0 && c->ev_valid[2]) {
        // the hit flags of the LAST ray-march: its own metavoxel count, not the current bin's (a slab re-cut or a re-bin since then may have
        // changed the occupied count beyond what the flag array holds: found by the fan-out re-cut test once the profile survived, round 4)
        const int n = c->brick_hit_n;
        int* h = (int*)malloc((size_t)n * sizeof(int));
        if (!h) return vp_fail(c, VP_ERR_OOM, "host allocation failed");
        hipError_t e = hipMemcpy(h, c->d_brick_hit, (size_t)n * sizeof(int), hipMemcpyDeviceToHost);
        int64_t cnt = 0;
        for (int i = 0; i < n; ++i) cnt += h[i] ? 1 : 0;
        free(h);
        if (e != hipSuccess) return vp_fail(c, VP_ERR_HIP, "hipMemcpy failed: %s", hipGetErrorString(e));
        st->bricks_sampled = cnt;
    }
    return VP_OK;
}

VP_EXPORT int vp_last_kernel_ms(vp_ctx* c, int32_t stage, float* ms)
{
    if (!c) return VP_ERR_BAD_ARG;
    if (!ms || stage < 0 || stage > 3) return vp_fail(c, VP_ERR_BAD_ARG, "vp_last_kernel_ms: bad argument");
    if (c->multi) return multi_last_kernel_ms(c, stage, ms);
    if (!c->ev_valid[stage]) return vp_fail(c, VP_ERR_STATE, "stage %d has not run", stage);
    int rc = ensure_device(c); if (rc) return rc;
    VP_HIP(hipEventSynchronize(c->ev[stage][1]));
    VP_HIP(hipEventElapsedTime(ms, c->ev[stage][0], c->ev[stage][1]));
    return VP_OK;
}

// ---- host-only planners (no device) --------------------------------------------------------------------------------------------------
VP_EXPORT int vp_plan_slabs(int32_t nz, int32_t world, const double* fill_ms, const double* rm_ms, int32_t rm_groups, int32_t* cuts_out)
{
    if (nz < 1 || world < 1 || world > nz || world > VP_MAX_RANKS || !cuts_out)
        return vp_fail(nullptr, VP_ERR_BAD_ARG, "vp_plan_slabs: %d ranks for %d z-slices (at most one rank per slice, <= %d)", world, nz, VP_MAX_RANKS);
    hl_plan_slabs(nz, world, fill_ms, rm_ms, rm_groups, cuts_out);
    return VP_OK;
}

VP_EXPORT int vp_exchange_plan(int32_t world, int32_t rank, int32_t straddler, int32_t all_gather_exchange, int32_t phase, vp_xop* ops_out,
                               int32_t cap, int32_t* n_out)
{
    if (world < 1 || world > VP_MAX_RANKS || rank < 0 || rank >= world || straddler < -1 || straddler >= world || (phase != 0 && phase != 1) ||
        !n_out || (cap > 0 && !ops_out) || cap < 0)
        return vp_fail(nullptr, VP_ERR_BAD_ARG, "vp_exchange_plan: bad argument");
    *n_out = hl_exchange_plan(world, rank, straddler, all_gather_exchange ? 1 : 0, phase, ops_out, cap);
    return *n_out <= cap ? VP_OK : vp_fail(nullptr, VP_ERR_BAD_ARG, "vp_exchange_plan: %d operations, room for %d", *n_out, cap);
}

VP_EXPORT int vp_blend_plan(int32_t world, const int32_t* cuts, int32_t z_boundary, int32_t* chain_out, int32_t* plan_rank, int32_t* plan_which,
                            int32_t* plan_kind, int32_t* n_plan, int32_t* straddler)
{
    if (world < 1 || world > VP_MAX_RANKS || !cuts || !chain_out || !plan_rank || !plan_which || !plan_kind || !n_plan)
        return vp_fail(nullptr, VP_ERR_BAD_ARG, "vp_blend_plan: bad argument");
    for (int r = 0; r < world; ++r)
        if (cuts[r] >= cuts[r + 1]) return vp_fail(nullptr, VP_ERR_BAD_ARG, "vp_blend_plan: slab %d is empty", r);
    *n_plan = hl_blend_plan(world, cuts, z_boundary, chain_out, plan_rank, plan_which, plan_kind, straddler);
    return VP_OK;
}
