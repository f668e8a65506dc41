// vpfx_internal.h -- shared declarations of libvpfx (context, kernel-constant PODs, launchers).
// gfx950 only.  Compiled with -ffp-contract=off: FMAs appear only where written as fmaf()/__builtin_fmaf
// (arithmetic spec).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

#include "../../include/vpfx.h"

#define VP_EXPORT extern "C" __attribute__((visibility("default")))

// vp_config.reserved[0]: test / measurement switch -- 1 keeps an R8 cube map on the global f32 footprint table (A/B against the LDS path)
#define VPFX_CFG_NO_LDS_CUBEMAP 1
// vp_config.reserved[1]: 1 keeps RGBA16F bricks when the ambient colour is grey (A/B against the luminance|density format)
#define VPFX_CFG_NO_GREY_BRICKS 1
// Test hook of the chained fill's watchdog: with the ENVIRONMENT variable VPFX_TEST_CHAIN_TIMEOUT=1 set when vp_create runs, the units await
// tags nobody writes and give up after a few polls (error path).  Not reachable through the ABI structs.

// ---------------------------------------------------------------------------------------------------
// Kernel-constant PODs (passed by value as kernel arguments -> SGPRs / kernarg segment)
// ---------------------------------------------------------------------------------------------------
struct GridConsts {
    int Nx, Ny, Nz, nv, b, z0, z1, pad0;
    float s, sb, one, inv_sb;
    float Linv[12];   // light worldToLocal, rows r: Linv[r*4 + c], c = 3 is the translation
    float lsO[3];     // light-space grid centre                                   VPR.cs:380
    float Rl[9];      // light rotation, row-major Rl[r*3 + c]
    float Rsb[9];     // Rl * sb   (columns of TRS(mvPos, lightRot, sb))           VPR.cs:596
    float rowsb[9];   // rows of (TRS(mvPos, lightRot, sb))^-1 linear part: rowsb[k*3+j] = Rl[j*3+k]/sb
    float fwd[3];     // dirLight.transform.forward.normalized                     VPR.cs:535
    float gc[3];
};

struct PsysConsts {
    float L2W[12];    // particle system localToWorld rows
    float axis[3];    // particleSys.transform.forward (normalised)
    int   rot_in_radians;
};

struct FillConsts {
    float opacity_factor, D, one_minus_D, init_light;
    float amb[3];
    int   fade;
    float bq, inv_a;              // light depth decode: lsSceneDepth = (d - bq) * inv_a   Fill.shader:218-219
    float camp[3];                // light camera position (gridCenter - fwd * 200)        VPR.cs:365
    float dstep[3];               // _LightForward * oneVoxelSize                          Fill.shader:183
    int   cubeS;
    float half_s, half_s_m05;     // S/2, S/2 - 0.5
    int   border_index;           // nv - clamp(b, 0, nv-2)                                Fill.shader:229
    float D_over_255;             // displacement scale for byte texels (R8 cube map in LDS): net = (D/255) * bilinear(bytes) + (1 - D)
    int   lds_pitch;              // row pitch of the padded byte table (cube_u8_pitch: S + 2 rounded up for bank spread)
    int   d_is_one;               // displacement scale == 1 exactly (net displacement can be 0: see cube_shade)
    int   grey;                   // ambient r == g == b: bricks stored as (luminance, density) z-pair entries (raymarch.hip)
};

struct RmConsts {
    int W, H, Nx, Ny, Nz, nv, z0, z1;
    int zB, steps, soft, partial;
    int wave_lx;                  // log2 of the ray-march wave's pixel-block extent along the lane-fastest screen axis: 3 = 8 x 8, 4 = 16 x 4, 5 = 32 x 2 (k_raymarch)
    int flags, num_covered, lane_transpose, occ_lds;  // (occ_lds: the grid's occupancy bitmask fits k_raymarch's LDS copy, launch_raymarch)  VP_RM_* bits of vp_raymarch_params.flags; _NumMetavoxelsCovered (VPR.cs:755); lanes run down screen columns
    float aspect, neg_inv_tan, zMin, s;
    float mvStep, inv_mvStep, nearc, farc;
    float c2m_lin[9];             // linear part of _CameraToMetavoxel (identical for every MV), rows     VPR.cs:778
    float inv_rows[9];            // linear part of TRS(mvPos, lightRot, s).inverse, rows
    float c2w_t[4];               // translation column of cameraToWorld (x, y, z, w)
    float c2g[12];                // camera space -> grid space (MV (x,y,z) spans [x,x+1)...), rows (traversal only)
    float camg[3];                // camera position in grid space
    float cam_world[3];           // camera position in world space (vp_camera.cam_pos): the columns' draw-order keys, k_rm_prepare
    float texScale, texBias;      // texel coordinate = local * (nv - 2b) + (b - 0.5)
    float inv_soft;
    float alpha_cutoff;           // early-out once (1 - dst.a) <= cutoff in the UNDER phase (0 = exact only)
};

// Cross-slab saturation hand-off of the ray-march (slab kernels only; raymarch.hip).  The reference's single render target sees every
// metavoxel (VPR.cs:652-711), so one GPU stops a ray as soon as it is saturated; a slab on its own only knows its own metavoxels.
// A hand-off map is one BYTE per pixel: code = floor(-8 log2 t) (capped at 255) of a transmittance t = 1 - alpha, decoded as
// 2^(-code / 8) >= t -- a conservative bound within a factor 2^(1/8) down to 2^-31.9; the product of maps is the sum of their codes.
// (2 MB per map at 1080p instead of 8.3 MB as f32: the maps cross xGMI point to point between slab groups, on the critical path.)
// t_in = n maps [n][H][W] of slabs that are composited IN FRONT of this one (received from the other GPUs); their product bounds what this
// slab can still contribute, and a ray stops once (1 - dst.a) * prod(t_in) <= 2^-25 -- with no map that is exactly the single-GPU rule
// 1 - dst.a == 0 (1 - a is a multiple of 2^-24 near a = 1), with maps it skips contributions of at most 3e-8.  t_out0 / t_out1 = this slab's
// own maps for the slabs behind it: 1 - alpha(phase-A image) and (1 - alpha(A)) (1 - alpha(B)) (equal unless the slab straddles
// zBoundary: phase-A slabs behind it are only hidden by its phase-A part).  zsamples[zz] += samples executed in light-axis slice zz (the
// work profile the slab cut is balanced with).  All pointers nullable.
#ifndef VPFX_ZPROF_COPIES
#define VPFX_ZPROF_COPIES 64      // zsamples is [COPIES][Nz], one copy per workgroup index mod COPIES: every wave adds to ~12 slices, and ~30 000
#endif                            // waves hammering the same 32 addresses would serialise in the L2's atomic units; readers sum the copies
struct RmHandoff {
    const uint8_t* t_in; int n_in; size_t plane;
    uint8_t* t_out0; uint8_t* t_out1;
    unsigned* zsamples;
};

// ---------------------------------------------------------------------------------------------------
// Device-side buffers and context
// ---------------------------------------------------------------------------------------------------
struct DevMeta {                  // small device-resident result block, copied to host after the scan
    int occupied;
    int pairs;
    int max_pairs;
    int unsorted_lists;           // MVs whose list was too long for the in-LDS rank sort
};

struct vp_multi;                  // multi.cpp: the fan-out (slab contexts, worker threads, RCCL communicator)

struct vp_ctx {
    vp_config cfg{};
    vp_multi* multi = nullptr;    // non-null: a fan-out context (its slabs live in child contexts); every entry point forwards
    bool test_chain_timeout = false;
    bool rm_xcd_affine = false;   // dispatch order of k_raymarch: compact screen region per XCD (k_tile_regions) instead of round robin in cost order
    int* d_tile_curve = nullptr;  // [super-tiles] Hilbert order of the super-tile grid (host-built at vp_create)
    bool rm_flat = false;         // VPFX_RM_FLAT=1 at vp_create: the wave-coherent ray-march traversal (k_raymarch_flat) -- A/B switch
    bool no_zprofile = false;     // VPFX_NO_ZPROFILE=1 in the environment at vp_create: measurement switch, slab ray-march without the per-slice sample profile
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;                 // vp_raymarch_async: device-to-host copy of the image beside the next frame's kernels
    hipEvent_t ev_image_ready = nullptr, ev_image_copied = nullptr;
    bool image_copy_pending = false;
    GridConsts g{};
    bool have_frame = false, have_particles = false, binned = false, filled = false, local_done = false, fill_begun = false;

    // frame
    float L[16]{}, gc[3]{};
    float* h_mvPos = nullptr;     // [N^3][3]
    float* d_mvPos = nullptr;

    // particles
    int P = 0, P_cap = 0;
    size_t raw_cap = 0;
    uint8_t* d_raw = nullptr;     // caller records as uploaded
    float4* d_ws = nullptr;       // [P] (ws.xyz, size)
    float* d_rec = nullptr;       // [P][16]
    vp_particle_layout lay{};
    PsysConsts psys{};

    // bins
    size_t n3 = 0;
    int* d_count = nullptr;       // [N^3]
    int* d_offsets = nullptr;     // [N^3 + 1]
    int* d_cursor = nullptr;      // [N^3]
    int* d_brick_index = nullptr; // [N^3]
    int* d_occ_list = nullptr;    // [N^3] (first `occupied` valid): MV linear index of brick i
    int* d_ids_tmp = nullptr;     // [pairs_cap]
    int* d_ids = nullptr;         // [pairs_cap]
    size_t pairs_cap = 0;
    int* d_onecol = nullptr;      // [2] one MV column index (per-metavoxel fill) + the cube-map range flag
    DevMeta* d_meta = nullptr;
    DevMeta* h_meta_host = nullptr;   // the same four words in pinned (coherent) host memory + a sequence word behind them, written by the scan kernel itself ...
    DevMeta* d_meta_host = nullptr;   // ... through this device address (no copy command between the scan and the host's wait)
    void* d_scan_totals = nullptr; // [ceil(N^3 / 1024)] per-tile totals of the two-launch scan
    DevMeta h_meta{};
    int bin_seq = 0;              // tag of the totals published by this frame's scan (launch_bin polls it in pinned host memory)

    // fill
    uint2* d_bricks = nullptr;    // [brick_cap][nv^3] x 8 B: RGBA16F texels, or (bricks_grey) z-pair entries (luminance|density)(z), (luminance|density)(z + 1)
    bool bricks_grey = false;     // format of the bricks as last filled
    size_t brick_cap = 0;
    float2* d_dens_ao = nullptr;  // split-fill scratch [brick_cap][nv^3]
    size_t dens_cap = 0;
    float* d_lightmap = nullptr;  // [(Ny*nv)][(Nx*nv)]
    float4* d_cubequads = nullptr;// footprint table: float2 column pairs [6][S+1][S+2] (see k_build_cubequads)
    int cubeS = 0;
    uint32_t* d_cube_u8 = nullptr;// R8 cube maps only: bytes [6][S+2][S+2], clamp border replicated (the LDS image of k_fill_lds)
    size_t cube_u8_cap = 0;       // bytes allocated
    int cube_u8_S = 0;            // 0 = no byte table resident (f32 cube map, or too large for LDS)
    int* d_work_counter = nullptr;// tile counter of the persistent fill
    int* d_ord = nullptr;         // [n3] per occupied MV: occupied MVs of its (xx, yy) column in front of it (owned slab)   (bin.hip)
    int* d_colcount = nullptr;    // [nxy] occupied MVs per column (owned slab)
    unsigned long long* d_chain = nullptr;   // [LH][LW] light hand-off words of the chained fill: tag << 32 | float bits     (fill.hip)
    uint32_t chain_seq = 0;       // fill launches since the hand-off words were last cleared
    int* h_chain_err = nullptr;   // pinned + mapped: set by a fill unit whose hand-off word never arrived (watchdog, fill.hip); read at every sync
    int* d_chain_err = nullptr;   // the device view of it
    int num_cus = 0;
    float* d_depthmap = nullptr;
    bool have_depthmap = false;
    FillConsts fc{};
    const float* finish_tau_all = nullptr;   // vp_fill_finish_gathered: [world][LH][LW] transmittance maps, only during that call
    int finish_n_before = 0;

    // occluder solids (scene-occlusion inputs produced on the GPU)
    vp_occluder* d_occluders = nullptr;
    int n_occluders = 0, occluders_cap = 0;
    // What d_scene_depth / d_depthmap hold when they were RENDERED from the solids (round 6): the eye depth is a function of the camera and the solids,
    // the light depth map of the frame (light, grid centre), the solids and the light camera's planes -- a static camera / light does not pay the render
    // again (the reference's scene: 1 of 2 launches per frame).  Generation 0 = nothing rendered (or the buffer was overwritten by a caller's map).
    unsigned occl_gen = 1, frame_gen = 1;                      // bumped by vp_set_occluders2 / vp_set_frame
    unsigned eye_depth_gen = 0; vp_camera eye_depth_cam{};
    unsigned light_depth_gen = 0, light_depth_frame = 0; float light_depth_planes[3] = {0.f, 0.f, 0.f};

    // raymarch
    float4* d_mvtrans = nullptr;  // [brick_cap] per-brick translation column of _CameraToMetavoxel
    size_t mvtrans_cap = 0;
    uint32_t* d_occmask = nullptr; // [Nz][Ny] one bit per cell (Nx <= 32): occupied metavoxels, copied into LDS by k_raymarch for the cell walk
    float4* d_cellinfo = nullptr; // [N^3] (translation, brick slot | -1) per cell: VPFX_RM_CELLINFO A/B variant of the cell walk
    int* d_rank = nullptr;        // [Ny*Nx] draw-order rank of the (yy, xx) columns for this frame's camera (VPR.cs:613-632), written by k_rm_prepare
    int* d_tile_order = nullptr;  // [rm_order_ints + super-tiles] dispatch order of k_raymarch (most expensive first), then the float cost estimates
    float* d_image = nullptr;     // [H][W][4]
    float* d_scene_depth = nullptr;
    unsigned long long* d_samples = nullptr;
    unsigned* d_zsamples = nullptr; // [VPFX_ZPROF_COPIES][Nz] samples executed per light-axis slice by the last slab ray-march (RmHandoff::zsamples)
    int* d_brick_hit = nullptr;   // [brick_hit_cap] set to 1 by the ray-march when a brick contributes a sample
    size_t brick_hit_cap = 0;
    int brick_hit_n = 0;          // metavoxels of the last ray-march (entries of d_brick_hit that frame used)
    long long last_samples = 0;

    hipEvent_t ev[4][2]{};        // start/stop of the dominant kernel per stage: 0 bin, 1 fill (fused or local), 2 raymarch, 3 fill_finish
    bool ev_valid[4] = {false, false, false, false};

    std::string err;
};

extern thread_local std::string g_vp_create_error;

int vp_fail(vp_ctx* c, int code, const char* fmt, ...);

#define VP_HIP(call)                                                                                  \
    do {                                                                                              \
        hipError_t e_ = (call);                                                                       \
        if (e_ != hipSuccess)                                                                         \
            return vp_fail(c, e_ == hipErrorOutOfMemory ? VP_ERR_OOM : VP_ERR_HIP, "%s failed: %s (%s:%d)", \
                           #call, hipGetErrorString(e_), __FILE__, __LINE__);                         \
    } while (0)

// host_logic.cpp
void   hl_build_grid(vp_ctx* c);                       // GridConsts + mvPos               VPR.cs:139,370-394
void   hl_build_psys(vp_ctx* c, const float m[16]);
void   hl_build_fill_consts(vp_ctx* c, const vp_fill_params* p);
int    hl_z_boundary(const vp_ctx* c, const vp_camera* cam);             //           VPR.cs:642-648
// ray-march screen decomposition: wave = 8x8 px, tile = 16x16 px (4 waves), super-tile = (16 << LX) x (16 << LY) px
// A/B builds (make EXTRA=-DVPFX_AB=1) carry the measured-and-dropped variants and their environment switches; the shipped library does not
#ifndef VPFX_AB
#define VPFX_AB 0
#endif
#ifndef VPFX_RM_LX
#define VPFX_RM_LX 2
#endif
#ifndef VPFX_RM_LY
#define VPFX_RM_LY 1
#endif
inline int rm_super_tiles_x(int W) { return (((W + 15) / 16) + (1 << VPFX_RM_LX) - 1) >> VPFX_RM_LX; }
inline int rm_super_tiles_y(int H) { return (((H + 15) / 16) + (1 << VPFX_RM_LY) - 1) >> VPFX_RM_LY; }
inline int rm_num_super_tiles(int W, int H) { return rm_super_tiles_x(W) * rm_super_tiles_y(H); }
// dispatch-order buffer of the ray-march (d_tile_order): [8 x cap] positions (XCD-affine order: region r owns r, 8 + r, ...; cap = the most
// super-tiles a region can hold, k_tile_regions), then the float cost estimates [nsuper]
inline int rm_order_cap(int nsuper) { return nsuper / 4 + 2; }
inline int rm_order_ints(int nsuper) { return 8 * rm_order_cap(nsuper) + 8; }
// slab cut + compositing order of the slabs (host only)
void   hl_plan_slabs(int nz, int world, const double* fill_ms, const double* rm_ms, int rm_groups, int* cuts /* [world + 1] */);
int    hl_blend_plan(int world, const int* cuts, int zb, int* chain, int* plan_rank, int* plan_which, int* plan_kind, int* straddler);
int    hl_exchange_plan(int world, int rank, int strad, int all_gather, int phase, vp_xop* ops, int cap);     // message schedule of the image exchange (vp_exchange_plan)
void   hl_chain_groups(int world, int rm_groups, int* group_of_pos /* [world]: group of chain position p */);
void   hl_build_rm_consts(const vp_ctx* c, const vp_camera* cam, const vp_raymarch_params* rp, RmConsts* k);

// bin.hip
int  launch_extract(vp_ctx* c);
int  launch_bin(vp_ctx* c);
int  launch_z_histogram(vp_ctx* c, int* d_hist);
// fill.hip
int  launch_build_cubequads(vp_ctx* c, const void* d_cube, int format, int S, int* d_bad);
int  launch_build_cube_u8(vp_ctx* c, const void* d_cube_r8, int S);                    // padded byte table for the LDS path
int  cube_u8_pitch(int S);                                                            // row pitch of the padded byte table
size_t cube_u8_bytes(int S);                                                          // 6 (S+2) rows of that pitch, rounded up to 16
int  launch_fill_one(vp_ctx* c, int xx, int yy, int zz);
int  launch_fill_value(vp_ctx* c, float* d, size_t n, float v, int* zero_word = nullptr);   // (zero_word: an int the same launch resets)                             // FillMetavoxel(xx, yy, zz)   VPR.cs:559
int  launch_fill(vp_ctx* c, int mode, const float* d_light_in, float* d_light_out);  // mode 0 fused, 1 local, 2 finish
// fill_generic.hip: the same for a voxel count that is not 16 / 32 / 64 (run-time nv; called by launch_fill / launch_fill_one)
int  launch_fill_generic(vp_ctx* c, int mode, const float* d_light_in, float* d_light_out, int math, bool lds);
int  launch_fill_one_generic(vp_ctx* c, const GridConsts& g, int math);
// raymarch.hip
int  launch_raymarch(vp_ctx* c, const RmConsts& k, float* d_over, float* d_under, const RmHandoff* handoff = nullptr);
int  launch_raymarch_one(vp_ctx* c, const RmConsts& k, int bi, int mi, int blend_over, int order_index, float* d_img);  // RenderMetavoxel VPR.cs:766
// raymarch_generic.hip: the march kernels for a voxel count that is not 16 / 32 / 64 (called by launch_raymarch / launch_raymarch_one)
int  launch_raymarch_generic(vp_ctx* c, const RmConsts& k, float* d_over, float* d_under, int early_out, const RmHandoff& ho);
int  launch_raymarch_one_generic(vp_ctx* c, const RmConsts& k, const void* brick, const float* tr, int blend_over, int order_index, float* d_img);
int    launch_blend(vp_ctx* c, const void* const* d_partials, const int32_t* kinds, int n, float* d_out, size_t npix);
int  launch_composite(vp_ctx* c, const float* d_particles, float* d_scene);
// api.cpp: the single-device entry points the fan-out calls on its slab contexts
int  api_read_zsamples(vp_ctx* c, long long* out /* [Nz] */, bool sync);   // sum of the copies; sync = wait for the stream first
int  vp_create_single(const vp_config* cfg, vp_ctx** out);
void vp_destroy_single(vp_ctx* c);
int  api_stream_sync(vp_ctx* c);
int  api_upload_particles(vp_ctx* c, const void* particles, int32_t count, const vp_particle_layout* lay, const float* psys_l2w, bool sync);
int  api_stage_fill_inputs(vp_ctx* c, const vp_fill_params* p);
int  api_ensure_bricks(vp_ctx* c, bool need_scratch);
int  api_stage_raymarch(vp_ctx* c, const vp_camera* cam, const vp_raymarch_params* rp, RmConsts* k);
int  api_set_slab(vp_ctx* c, int z0, int z1);
// multi.cpp
int  multi_create(const vp_config* cfg, vp_ctx** out);
void multi_destroy(vp_ctx* c);
int  multi_set_frame(vp_ctx* c, const float* l2w, const float* gc);
int  multi_upload_particles(vp_ctx* c, const void* particles, int32_t count, const vp_particle_layout* lay, const float* psys_l2w);
int  multi_bin_resident(vp_ctx* c);
int  multi_fill(vp_ctx* c, const vp_fill_params* p);
int  multi_raymarch(vp_ctx* c, const vp_camera* cam, const vp_raymarch_params* rp, float* host_out, void* d_out);
int  multi_sync(vp_ctx* c);
int  multi_set_occluders(vp_ctx* c, const vp_occluder* solids, int32_t n);
int  multi_get_stats(vp_ctx* c, vp_stats* st);
int  multi_last_kernel_ms(vp_ctx* c, int stage, float* ms);
int  multi_read_bincounts(vp_ctx* c, int32_t* counts);
int  multi_read_lightmap(vp_ctx* c, float* out);
vp_ctx* multi_owner_of_slice(vp_ctx* c, int zz);       // local child owning light-axis slice zz, or nullptr
// unity_plugin.cpp
int  vp_read_last_image(vp_ctx* c, const void* d_img, float* h_out);
// occluders.hip
int  launch_light_depth(vp_ctx* c, float nearz, float farz, float cam_dist, float* d_out);
int  launch_scene_depth(vp_ctx* c, const vp_camera* cam, float* d_out);
