/*
 * vpfx.h -- C ABI of libvpfx: the MI355X-native sparse-volumetric-particle hot path
 * (particle binning -> FillVolume + light propagation -> RayMarch + inter-metavoxel blend).
 *
 * This is the drop-in boundary for the path that, in the reference
 * (rajabala/Volumetric-Particles-For-Unity, file = Assets/Main Scene/VolumetricParticleRenderer.cs,
 * "VPR.cs" below), is driven by VolumetricParticleRenderer.OnPostRender (VPR.cs:181-220).
 * The reference has no native boundary (no DllImport anywhere); the calls being replaced are the
 * Unity Graphics calls of that path.  Each entry point cites the reference code it replaces.
 *
 * Conventions (same as the reference):
 *   - every float[16] is a Unity Matrix4x4 in memory order = COLUMN-major (m00,m10,m20,m30,m01,...);
 *     vectors are column vectors (HLSL mul(M,v), Fill.shader:106).
 *   - metavoxel ("MV") arrays are indexed [zz][yy][xx] (VPR.cs:287); zz = 0 is nearest the light.
 *   - a brick (one MV's 3D texture, RenderTextureFormat.ARGBHalf, VPR.cs:312) is
 *     [slice][py][px][rgba] IEEE binary16, rgb = lit colour, a = density.
 *   - light-propagation / light-depth maps are [py + yy*nv][px + xx*nv] float32 (VPR.cs:259-275).
 *   - images are [row][col][rgba] float32 premultiplied alpha, row 0 = bottom of the view.
 *   - the caller owns every pointer it passes; the library copies during the call and keeps none.
 *   - one context is NOT thread-safe; calls on one context must be serialised by the caller
 *     (the reference calls everything from Unity's main thread).
 *   - every function returns VP_OK (0) or a negative vp_status; no C++ exception crosses the ABI.
 *     The reference logs and carries on (Debug.LogError, VPR.cs:352,790); callers may do the same.
 *   - there is NO CPU fallback: without a HIP device vp_create fails with VP_ERR_NO_DEVICE.
 */
#ifndef VPFX_H
#define VPFX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VPFX_ABI_VERSION 6   /* 2: vp_fill_params.cubemap_format (R8 cube maps), per-metavoxel entry points, draw-order view
                                3: vp_config device list (multi-GPU fan-out inside the library, RCCL), VP_ERR_RCCL, Unity plugin entry points
                                4: same struct layouts; new: vp_config.reserved[2] = exchange time-out of a fan-out context (abort instead of hang),
                                   VP_MULTI_TEST_HOOKS / VP_MULTI_TEST_DROP_SEND, VP_RM_NO_EARLY_OUT, vp_exchange_plan, vp_unity_clear_slot, vp_unity_register_output_fd
                                5: same struct layouts; new: the particle source (vp_emitter_*)
                                6: same struct layouts; new: vp_occluder / vp_set_occluders2 (cylinder and ellipsoid occluders beside boxes).
                                   Stricter since 5: vp_emitter_config.reserved[] must be 0 and .lifetime finite, <= 1e5 s (VP_ERR_BAD_ARG otherwise) */

typedef enum vp_status {
    VP_OK = 0,
    VP_ERR_BAD_ARG = -1,     /* null pointer / out-of-range value                               */
    VP_ERR_HIP = -2,         /* a HIP runtime call failed; see vp_last_error                    */
    VP_ERR_OOM = -3,         /* device or host allocation failed                                */
    VP_ERR_STATE = -4,       /* call order violated (e.g. raymarch before fill)                 */
    VP_ERR_NO_DEVICE = -5,   /* no HIP device visible (there is no CPU fallback)                */
    VP_ERR_UNSUPPORTED = -6, /* configuration outside what the kernels are built for            */
    VP_ERR_RCCL = -7         /* librccl missing, or an RCCL call failed; see vp_last_error      */
} vp_status;

typedef struct vp_ctx vp_ctx;

/* Grid / screen configuration.  Counterpart of the inspector fields VPR.cs:82-86 and of
 * CreateResources (VPR.cs:224-281).  Metavoxels are cubes (the reference assumes it, VPR.cs:422). */
typedef struct vp_config {
    int32_t num_mv[3];        /* numMetavoxelsX, Y, Z                                  VPR.cs:83 */
    int32_t num_voxels;       /* numVoxelsInMetavoxel (nv): any value in [2, 64], odd ones included (the reference's shader stops at
                                 NUM_VOXELS = 32, Fill.shader:16); 16 / 32 / 64 run the specialised kernels          VPR.cs:85 */
    int32_t num_border;       /* numBorderVoxels per end                               VPR.cs:86 */
    float   mv_scale;         /* mvScale.x: world size of one metavoxel                VPR.cs:84 */
    int32_t width, height;    /* Screen.width / height (particlesRT extent)            VPR.cs:228 */
    int32_t device;           /* HIP device ordinal, -1 = current device                          */
    int32_t slab_z0, slab_z1; /* owned light-axis slab zz in [z0,z1); 0,0 = whole grid (1 GPU)    */
    int32_t exact_math;       /* 1: IEEE divisions in the fill kernel (bit-parity test builds)     */
    int32_t no_early_out;     /* 1: the ray-march never stops early (sample-count parity tests)    */
    int32_t reserved[3];      /* 0.  (Measurement switches: [0] = 1 keeps an R8 cube map out of LDS,
                                 [1] = 1 keeps RGBA16F bricks when the ambient colour is grey.  [2]: single-device context: must be
                                 0; fan-out context: time-out in ms of every inter-rank exchange and of every wait for one
                                 (0 = 120 000: RCCL sets its peer connections up lazily inside the first exchanges, which can take many seconds on 8 GPUs; <= 3 600 000) -- when it expires the context ABORTS: ncclCommAbort on every local
                                 communicator, every local rank returns VP_ERR_RCCL, every later call fails fast until vp_destroy.
                                 Any other value is refused with VP_ERR_BAD_ARG.)                    */
    /* ---- ABI 3: multi-GPU fan-out INSIDE the library ("device list", "multi-GPU fan-out is internal").
     * num_devices <= 1 and world_size == 0: one GPU (`device`), everything above.  Otherwise the context is a FAN-OUT context: the grid
     * is cut into world_size contiguous light-axis slabs (cost-balanced, see vp_rebalance), slab r lives on the GPU of rank r, and
     * vp_set_frame / vp_bin / vp_fill / vp_raymarch stay the only calls the host makes: the library runs one host thread, one HIP stream
     * and one RCCL rank per local device, all-gathers the slab transmittance maps (fill), hands the saturation of the slabs in front to
     * the slabs behind and exchanges + blends the partial images (ray-march) over RCCL / xGMI.  The image is delivered on rank 0.
     *   one process drives all GPUs (the C# / C host):  num_devices = N, devices[] = the HIP ordinals, world_size = 0
     *   one process per GPU (torchrun-style launch):    num_devices = 1, devices[0] = local GPU, world_size = N, first_rank = rank,
     *                                                    rccl_unique_id = the 128 bytes vp_rccl_unique_id() returned on rank 0 */
    int32_t num_devices;      /* local GPUs this process drives (0 = 1 = `device` alone unless world_size > 1)       */
    int32_t devices[8];       /* HIP ordinals, slab order; entries may only repeat with VP_MULTI_PEER_COPY            */
    int32_t world_size;       /* total ranks (= slabs) of a multi-process job; 0 = num_devices                        */
    int32_t first_rank;       /* rank of devices[0] (ranks of one process are consecutive)                            */
    int32_t multi_flags;      /* VP_MULTI_* bits                                                                       */
    int32_t rm_groups;        /* saturation hand-off: the slabs, front to back, form this many groups; a group's slabs march
                                 concurrently knowing the opacity of every earlier group.  0 / 1 = no hand-off (default: every slab
                                 marches at once -- shortest frame, most samples), world_size = a fully serial chain (exactly the
                                 single GPU's samples, longest frame: the groups run one after the other)             */
    uint8_t rccl_unique_id[128];
} vp_config;

#define VP_MAX_LOCAL_DEVICES 8
#define VP_MAX_RANKS 16
/* vp_config.multi_flags */
#define VP_MULTI_PEER_COPY           1  /* TEST HOOK: no RCCL; every exchange is a device-to-device copy between the local contexts (all ranks
                                           must be local).  devices[] may then repeat, e.g. {0,0,0,0}: four slabs on one GPU                */
#define VP_MULTI_EXCHANGE_ALL_GATHER 2  /* image exchange = ONE all-gather of whole partial images, blended on the display rank (the north-star
                                           form); default = all-to-all of screen pieces, sharded blend, gather (8x less xGMI traffic)        */
#define VP_MULTI_UNIFORM_SLABS       4  /* equal-thickness slabs; default: balanced from the work histograms                                  */
#define VP_MULTI_FORCE               8  /* take the fan-out path (threads, RCCL communicator, collectives) even with one rank                 */
#define VP_MULTI_TEST_DROP_SEND     16  /* TEST HOOK (needs VP_MULTI_TEST_HOOKS + VP_MULTI_PEER_COPY): the last rank silently skips the first message it
                                           should send -- its peers must time out, the context must abort and every call return VP_ERR_RCCL     */
#define VP_MULTI_TEST_SHARED_DEVICE  32  /* TEST HOOK (needs VP_MULTI_TEST_HOOKS): devices[] may repeat on the RCCL path.  The real librccl refuses a duplicate device;
                                           the tests' checking stand-in (tests/tools/fake_rccl.cpp, first on LD_LIBRARY_PATH) runs N ranks on one GPU with RCCL's
                                           in-issue-order matching enforced                                                                     */
#define VP_MULTI_TEST_HOOKS 0x40000000  /* opt-in for test hooks (any context, single-device ones too): only with this bit does vp_create read the
                                           VPFX_TEST_* environment switches (VPFX_TEST_CHAIN_TIMEOUT=1: the fill's chain watchdog test)           */

/* Byte layout of one caller-side particle record (ParticleSystem.Particle[], VPR.cs:412-413).
 * Offsets are explicit because the managed struct layout is Unity-version specific. */
typedef struct vp_particle_layout {
    int32_t stride;               /* bytes per record                                             */
    int32_t off_position;         /* 3 x f32, particle-system local space       .position :418    */
    int32_t off_size;             /* f32, diameter                              .size     :425    */
    int32_t off_rotation;         /* f32                                        .rotation :583    */
    int32_t off_lifetime;         /* f32, remaining seconds                     .lifetime :586    */
    int32_t off_start_lifetime;   /* f32                                        .startLifetime    */
    int32_t rotation_in_radians;  /* 0: degrees (the C# property), 1: radians (the raw field)     */
    int32_t reserved;
} vp_particle_layout;

/* Uniforms of the fill pass: SetFillPassConstants (VPR.cs:523-554). */
typedef struct vp_fill_params {
    float   opacity_factor;       /* _OpacityFactor                                               */
    float   displacement_scale;   /* _DisplacementScale                                           */
    int32_t fade_out_particles;   /* _FadeOutParticles                                            */
    float   ambient[3];           /* _AmbientColor                                                */
    float   init_light_intensity; /* _InitLightIntensity (reference passes 1.0)                   */
    float   light_near, light_far;/* light camera clip planes (0.3 / 1000, VPR.cs:340-342)        */
    float   light_cam_distance;   /* light camera sits gridCenter - fwd*200 (VPR.cs:365)          */
    int32_t cubemap_size;         /* S: edge of one displacement cubemap face                     */
    int32_t cubemap_format;       /* VP_CUBEMAP_F32 (0) or VP_CUBEMAP_R8 (1)                       */
    const void* cubemap;          /* 6*S*S texels = .x channel of _DisplacementTexture; faces in D3D
                                     order +X,-X,+Y,-Y,+Z,-Z; row 0 = top; bilinear, clamp.
                                     F32: floats in [0,1].  R8: bytes, texel = byte/255 (UNORM) -- the
                                     reference's own asset is 8-bit (ARGB32, Assets/Textures/
                                     DisplacementTexture.cubemap:10-23, bound FillVolume.mat:23-29).
                                     NULL = keep the cube map of the previous vp_fill resident.
                                     displacement_scale must lie in [0,1] (the reference's slider,
                                     scene:8103-8111) and every texel in [0,1]: netDisplacement >= 0
                                     is what Fill.shader:119-126 assumes; anything else is refused
                                     with VP_ERR_BAD_ARG                                           */
    const float* light_depth_map; /* optional [(Ny*nv)][(Nx*nv)] f32 in [0,1] (_LightDepthMap,
                                     D3D ortho depth); NULL = 1.0 everywhere (no occluders)       */
} vp_fill_params;

#define VP_CUBEMAP_F32 0
#define VP_CUBEMAP_R8  1

/* Main camera: SetRaymarchPassConstants (VPR.cs:733-737) + RenderMetavoxel (VPR.cs:778). */
typedef struct vp_camera {
    float world_to_camera[16];    /* Camera.worldToCameraMatrix (view space looks down -Z)        */
    float camera_to_world[16];    /* Camera.cameraToWorldMatrix                                   */
    float cam_pos[3];             /* Camera.main.transform.position                               */
    float fov_y;                  /* vertical field of view in RADIANS (_Fov)                     */
    float near_clip, far_clip;    /* clip planes used by the rasteriser coverage rule             */
} vp_camera;

typedef struct vp_raymarch_params {
    int32_t steps_per_mv;         /* rayMarchSteps (_NumRaymarchStepsPerMV)                       */
    int32_t soft_distance;        /* softParticleStepDistance (_SoftDistance)                     */
    const float* scene_depth;     /* optional [H][W] f32 linear eye depth of opaque scene for the
                                     ZTest Less rejection (RM.shader:14); NULL = no occluders     */
    int32_t flags;                /* VP_RM_* bits                                                  */
    int32_t reserved[3];
} vp_raymarch_params;

/* vp_raymarch_params.flags */
#define VP_RM_QUANTIZE_UNORM8  1  /* emulate the reference's 8-bit particlesRT (ARGB32, VPR.cs:228): the render target is
                                     re-quantised to UNORM8 after every metavoxel blend.  Default: fp32 */
#define VP_RM_SHOW_NUM_SAMPLES 2  /* _ShowNumSamples debug view: colour-code the samples taken per metavoxel (RM.shader:283-299) */
#define VP_RM_SHOW_BLEND_FUNC  4  /* _ShowRayMarchBlendFunc debug view: yellow = OVER, cyan = UNDER (RM.shader:174-181)        */
#define VP_RM_SHOW_DRAW_ORDER  8  /* _ShowMetavoxelDrawOrder debug view: every metavoxel coloured by its position in the global
                                     submission order (DrawOrderColoring, RM.shader:123-138, 170-173; _OrderIndex = mvCount of
                                     VPR.cs:650-706, _NumMetavoxelsCovered VPR.cs:755).  Whole-grid contexts only               */

#define VP_RM_SHOW_RAY_SAMPLES 16  /* perf view (not in the reference): every pixel = the number of lattice samples its ray executed, as
                                     (n, n, n, 1) f32 -- the samples-per-ray heat map of SURVEY 8(f) row 4.  Unlike the reference's debug
                                     views it leaves the saturation early-out on: it shows the work the frame really did */

#define VP_RM_NO_EARLY_OUT     32  /* this call marches every lattice sample of SURVEY 8(d)'s formula (sum over pixels and metavoxels of
                                     max(0, tExit - tEntry + 1)) -- the saturation early-out off for ONE call, on the resident bricks (the
                                     per-context form is vp_config.no_early_out).  Same image up to rounding; vp_stats.samples then is the
                                     formula count.  Does not select the debug-view kernels */

/* An opaque occluder: oriented box (the demo scene's ground/back planes and cubes are boxes).  Used to PRODUCE the two
 * scene-occlusion inputs of the path on the GPU instead of reading them back from Unity render targets:
 *   light depth map  <- lightCamera.RenderWithShader(GenerateLightDepthMap)   VPR.cs:184, 320-367, LDM.shader:6 (Cull Front:
 *                        the nearest BACK face is what lands in the depth buffer)
 *   eye scene depth  <- the main camera's depth buffer used by ZTest Less    VPR.cs:204, RM.shader:14 */
typedef struct vp_obb {
    float center[3];
    float axes[9];                /* rows = the box's unit axes in world space                     */
    float half_extent[3];
} vp_obb;

/* ABI 6: the general occluder record.  The first 60 bytes ARE a vp_obb; `type` picks the unit solid that
 * center / axes / half_extent place in the world (local point l = diag(1/half_extent) * axes * (p - center)):
 *   VP_OCC_BOX        |l.x|, |l.y|, |l.z| <= 1                                (Unity's Cube, mesh 10202: half_extent = scale / 2)
 *   VP_OCC_CYLINDER   l.x^2 + l.z^2 <= 1, |l.y| <= 1: capped, axis = axes row 1 (Unity's Cylinder, mesh 10206: radius 0.5, height 2
 *                     => half_extent = (scale.x / 2, scale.y, scale.z / 2); the mesh is a 20-sided prism, the analytic solid
 *                     differs from it by <= 0.62 % of the radius)
 *   VP_OCC_ELLIPSOID  |l|^2 <= 1                                                (Unity's Sphere, mesh 10207: half_extent = scale / 2)
 * The reference's scene holds two scaled cubes (ground, back wall), two unit cubes and four unit cylinders on the Default layer
 * (Assets/Volumetric_Particle_System.unity:1755, 5462, 8382, 8623 are the cylinders), all drawn into both depth inputs
 * (VPR.cs:184 cullingMask = Default, LDM.shader:6-31; RM.shader:14 ZTest against the main camera's depth). */
#define VP_OCC_BOX       0
#define VP_OCC_CYLINDER  1
#define VP_OCC_ELLIPSOID 2
typedef struct vp_occluder {
    float   center[3];
    float   axes[9];              /* rows = the solid's unit axes in world space                   */
    float   half_extent[3];
    int32_t type;                 /* VP_OCC_*                                                      */
} vp_occluder;

#define VP_BRICKS_RGBA16F    0
#define VP_BRICKS_GREY_ZPAIR 1

typedef struct vp_stats {
    int64_t particles;            /* particles uploaded                                           */
    int64_t occupied_mv;          /* numMetavoxelsCovered (VPR.cs:515), within the owned slab     */
    int64_t pairs;                /* sum over MVs of mParticlesCovered.Count                      */
    int64_t voxels_filled;        /* occupied_mv * nv^3                                           */
    int64_t samples;              /* ray-march samples of the last vp_raymarch* call              */
    int64_t brick_bytes;          /* resident brick pool bytes                                    */
    int64_t max_pairs_per_mv;
    int64_t bricks_sampled;       /* bricks that contributed >= 1 sample to the last vp_raymarch* call */
    int64_t brick_bytes_per_voxel;/* 8 (both storage formats below)                               */
    int64_t brick_format;         /* VP_BRICKS_RGBA16F, or VP_BRICKS_GREY_ZPAIR: (luminance, density) fp16 pairs of the voxel and of its
                                     z + 1 neighbour -- the storage the library picks when the ambient colour is grey (r = g = b in
                                     every voxel) and volume_border >= 1; vp_read_brick always returns RGBA16F */
    int64_t reserved[2];
} vp_stats;

/* ---- lifetime ------------------------------------------------------------------------------ */
/* Start() + CreateResources (VPR.cs:132-149, 224-317). */
int  vp_create(const vp_config* cfg, vp_ctx** out);
void vp_destroy(vp_ctx* ctx);
/* Message for the most recent failure on ctx (or on vp_create when ctx == NULL). */
const char* vp_last_error(const vp_ctx* ctx);
int  vp_abi_version(void);
/* Run all device work of this context on the given hipStream_t (NULL = default stream). */
int  vp_set_stream(vp_ctx* ctx, void* hip_stream);
/* Block until all device work queued by this context has finished. */
int  vp_sync(vp_ctx* ctx);
/* Optional: page-lock a caller-owned host buffer that is handed to vp_bin / vp_upload_particles (the Particle[] array) or
 * vp_raymarch (the RGBA read-back) every frame, so that the PCIe copies run at DMA speed instead of through a bounce
 * buffer (33 MB read-back at 1080p: 1.45 -> ~0.6 ms).  The caller keeps the buffer alive and unmoved until
 * vp_unpin_host_buffer (a pinned managed array in C#: GCHandle.Alloc(..., Pinned)) -- and must unpin it BEFORE freeing it: the page lock
 * belongs to the address range, and memory the allocator hands out again inside a stale range makes later copies fail (vp_destroy does not
 * know the caller's buffers).  Purely a speed hint: every entry point accepts unpinned memory. */
int  vp_pin_host_buffer(vp_ctx* ctx, void* ptr, uint64_t bytes);
int  vp_unpin_host_buffer(vp_ctx* ctx, void* ptr);

/* ---- per-frame host logic ------------------------------------------------------------------- */
/* UpdateMetavoxelPositions (VPR.cs:370-394).  light_to_world = dirLight.transform.localToWorldMatrix
 * (rigid), grid_center = gridCenter.transform.position. */
int  vp_set_frame(vp_ctx* ctx, const float light_to_world[16], const float grid_center[3]);

/* BinParticlesToMetavoxels (VPR.cs:397-457) = vp_upload_particles + vp_bin_resident.
 * `particles` is host memory (e.g. a pinned ParticleSystem.Particle[]).  A particle whose position or size is not finite
 * is skipped (the reference indexes its grid with (int)NaN there, VPR.cs:434-438). */
int  vp_bin(vp_ctx* ctx, const void* particles, int32_t count, const vp_particle_layout* layout,
            const float psys_local_to_world[16]);
int  vp_upload_particles(vp_ctx* ctx, const void* particles, int32_t count,
                         const vp_particle_layout* layout, const float psys_local_to_world[16]);
int  vp_bin_resident(vp_ctx* ctx);

/* FillMetavoxels / FillMetavoxel + FillVolume.shader (VPR.cs:495-609, Fill.shader:152-274). */
int  vp_fill(vp_ctx* ctx, const vp_fill_params* params);

/* The reference's per-metavoxel entry point FillMetavoxel(xx, yy, zz) (VPR.cs:559-609), for callers that drive the fill
 * themselves (debugging one metavoxel, partial refills):
 *   vp_fill_begin      = the head of FillMetavoxels: SetFillPassConstants + clear of lightPropogationTex to 1.0
 *                        (VPR.cs:497-503, 523-554); same params as vp_fill
 *   vp_fill_metavoxel  = FillMetavoxel: fills ONE metavoxel's brick from its particle list, reading the light that
 *                        reaches it from the light-propagation map and writing the transmitted light back
 *                        (Fill.shader:224, 250).  An empty metavoxel is a no-op (the reference never calls it for one,
 *                        VPR.cs:511).  Calling it for every occupied metavoxel in zz-major order after vp_fill_begin
 *                        reproduces vp_fill: bit for bit with an f32 cube map; within 1 fp16 ulp with an R8 map (vp_fill keeps
 *                        that one in LDS as bytes and folds 1/255 into the displacement scale, the per-metavoxel kernel filters
 *                        byte/255 floats from the global table: the same value up to rounding).
 *   A vp_fill_begin whose ambient colour changes the bricks' storage format (grey z-pair entries <-> RGBA16F) clears every
 *   brick first: a partial refill must never leave bricks of two formats in one pool. */
int  vp_fill_begin(vp_ctx* ctx, const vp_fill_params* params);
int  vp_fill_metavoxel(vp_ctx* ctx, int32_t xx, int32_t yy, int32_t zz);

/* RenderMetavoxels / RenderMetavoxel + RayMarchVoxel.shader + ROP blend
 * (VPR.cs:613-794, RM.shader:14-18,95-302).  rgba_out = particlesRT as [H][W][4] f32.
 * vp_raymarch synchronises and copies to host; the _device form writes device memory and is
 * asynchronous on the context's stream. */
int  vp_raymarch(vp_ctx* ctx, const vp_camera* cam, const vp_raymarch_params* params, float* rgba_out);
int  vp_raymarch_device(vp_ctx* ctx, const vp_camera* cam, const vp_raymarch_params* params,
                        void* d_rgba_out);
/* vp_raymarch without the wait ("or offers _async + vp_wait"): the ray-march is queued, a second stream copies the image to
 * rgba_out as soon as it is complete, and the call returns; the next frame's vp_bin* / vp_fill run beside the copy (33 MB at 1080p).
 * vp_wait_image blocks until the image has landed in rgba_out (also done by vp_sync).  One image in flight per context: a later
 * vp_raymarch / vp_raymarch_async waits on the device for the copy before it reuses the context's image.  rgba_out must stay valid until
 * vp_wait_image and should be page-locked (vp_pin_host_buffer): with a pageable buffer the call blocks for the copy.  On a fan-out
 * (multi-GPU) context this is the synchronous vp_raymarch and vp_wait_image returns at once. */
int  vp_raymarch_async(vp_ctx* ctx, const vp_camera* cam, const vp_raymarch_params* params, float* rgba_out);
int  vp_wait_image(vp_ctx* ctx);

/* The reference's per-metavoxel entry point RenderMetavoxel(xx, yy, zz, orderIndex) (VPR.cs:766-794) on the context's own
 * particlesRT (VPR.cs:228), with the blend state RenderMetavoxels sets around it (VPR.cs:659-662 / 688-691):
 *   vp_clear_particles_rt  = OnPreRender: GL.Clear(particlesRT, (0,0,0,0))                         VPR.cs:168-177
 *   vp_render_metavoxel    = one DrawMeshNow(cube): every pixel whose ray hits the metavoxel marches it and blends the
 *                            result into particlesRT; blend_over = 1: Blend One OneMinusSrcAlpha, 0: Blend OneMinusDstAlpha One.
 *                            order_index is only read by the VP_RM_SHOW_DRAW_ORDER view (_OrderIndex).  Empty MV: no-op.
 *   vp_read_particles_rt   = read particlesRT back, [H][W][4] f32.
 * Submitting every occupied metavoxel in the order of RenderMetavoxels reproduces vp_raymarch (up to fp32 rounding of the
 * blend association, ~1e-7; exactly with VP_RM_QUANTIZE_UNORM8 off and the literal order of the flags kernel). */
int  vp_clear_particles_rt(vp_ctx* ctx);
int  vp_render_metavoxel(vp_ctx* ctx, const vp_camera* cam, const vp_raymarch_params* params, int32_t xx, int32_t yy, int32_t zz,
                         int32_t blend_over, int32_t order_index);
int  vp_read_particles_rt(vp_ctx* ctx, float* rgba_out);

/* CompositeParticles.shader (Comp.shader:10, VPR.cs:210): scene.rgb = p.rgb + scene.rgb*(1-p.a),
 * scene.a += p.a; in place on device images [H][W][4] f32. */
int  vp_composite_device(vp_ctx* ctx, const void* d_particles_rgba, void* d_scene_rgba);

/* ---- scene occluders ---------------------------------------------------------- */
/* Keep `n` occluder boxes in the context (n = 0 removes them).  While set, a vp_fill whose params carry no
 * light_depth_map renders one from the boxes (ortho light camera of VPR.cs:320-367: extents = the grid's x/y size,
 * position gridCenter - fwd * light_cam_distance, D3D depth (z - near)/(far - near)), and a vp_raymarch* whose params
 * carry no scene_depth renders the eye depth (linear, nearest front face) from them. */
int  vp_set_occluders(vp_ctx* ctx, const vp_obb* boxes, int32_t n);
/* ABI 6: the same with typed solids (boxes, capped cylinders, ellipsoids); vp_set_occluders(boxes) == vp_set_occluders2 with
 * type = VP_OCC_BOX.  Both depth inputs take, per ray, the nearest BACK face under the light camera (Cull Front) and the nearest front
 * face under the main camera, exactly as for boxes.  An unknown type or a half_extent <= 0 is VP_ERR_BAD_ARG.
 * A map rendered from the solids is kept in the context and reused while what it depends on is unchanged (eye depth: the vp_camera and the
 * solids; light depth map: vp_set_frame's arguments, the solids and light_near / light_far / light_cam_distance); passing a map of your own
 * replaces it. */
int  vp_set_occluders2(vp_ctx* ctx, const vp_occluder* solids, int32_t n);
/* Parity probes: render and read back the two maps. */
int  vp_render_light_depth(vp_ctx* ctx, float light_near, float light_far, float light_cam_distance, float* out /* [(Ny*nv)][(Nx*nv)] */);
int  vp_render_scene_depth(vp_ctx* ctx, const vp_camera* cam, float* out /* [H][W] linear eye depth, 3e38 = nothing */);

/* ---- multi-GPU inside the library (fan-out contexts, vp_config.num_devices / world_size) ---------------------- */
typedef struct vp_multi_info {
    int32_t world_size, num_local, first_rank;
    int32_t rccl_ranks;           /* ncclCommCount of the communicator; 0 with VP_MULTI_PEER_COPY                              */
    int32_t exchange;             /* 0 tiles (all-to-all + gather), 1 all-gather                                               */
    int32_t rm_groups;
    int32_t slab_cuts[VP_MAX_RANKS + 1];   /* slab r = zz in [slab_cuts[r], slab_cuts[r + 1])                                  */
    int32_t chain[VP_MAX_RANKS];  /* ranks front to back as the last vp_raymarch composited them                               */
    int32_t group_of[VP_MAX_RANKS];        /* hand-off group of every rank in that frame                                       */
    int64_t samples[VP_MAX_RANKS];         /* lattice samples executed per LOCAL rank in the last vp_raymarch (0 for remote ones) */
    float   stage_ms[VP_MAX_RANKS][4];     /* per LOCAL rank: bin, fill (local pass; rank 0: its fused fill), ray-march, fill finish kernel times (rank 0: none) */
    float   exchange_ms[4];       /* rank-0-of-this-process stream time: tau all-gather, saturation hand-off, image exchange + blend, - */
} vp_multi_info;
int  vp_get_multi_info(vp_ctx* ctx, vp_multi_info* out);
/* Ask for a new slab cut: the NEXT vp_raymarch also records how many samples it executes per light-axis slice (+3 % on that frame), and
 * the vp_bin* after it re-cuts the slabs, weighting every slice with its (particle, metavoxel) pairs (fill) and those samples (ray-march),
 * scaled by the kernel times measured on the GPUs; collective over all ranks (every rank must call it in the same frame).  The first
 * vp_bin* of a context cuts from the pair histogram alone. */
int  vp_rebalance(vp_ctx* ctx);
/* 128 bytes identifying a new RCCL communicator (ncclGetUniqueId): call on ONE process, hand the bytes to every process of the job
 * (any transport: MPI, a TCP store, a file) and pass them in vp_config.rccl_unique_id. */
int  vp_rccl_unique_id(uint8_t out[128]);
/* The slab cut itself (host-only, no GPU needed): fill_ms[z] (the fill's local pass), rm_ms[z] = estimated milliseconds per light-axis
 * slice.  The slabs behind the first wait for the slowest slab's local pass (it ends in the all-gather of the transmittance maps) and then run
 * their finish pass (0.42 x the local pass) + ray-march (per hand-off group when rm_groups > 1); the slab nearest the light runs the fused fill,
 * needs nobody's light and marches as soon as its own fill is done (the all-gather runs on the library's exchange stream beside it): the frame
 * is the later of the two paths, and the cut minimises that.  cuts_out[world + 1]. */
int  vp_plan_slabs(int32_t nz, int32_t world, const double* fill_ms, const double* rm_ms, int32_t rm_groups, int32_t* cuts_out);
/* Front-to-back compositing order of the slabs for a given zBoundary (VPR.cs:652-711 at slab granularity) and the blend plan of their
 * partial images: chain_out[world] = ranks front to back (the slab straddling zBoundary first, then the phase-A slabs zz descending, then
 * the phase-B slabs zz ascending); plan_rank/plan_which/plan_kind [<= world + 1] = partial images in blend order (which 0 = the slab's
 * first image, 1 = the straddler's phase-B image; kind 0 OVER, 1 UNDER); returns the plan length in *n_plan, the straddler or -1. */
int  vp_blend_plan(int32_t world, const int32_t* cuts, int32_t z_boundary, int32_t* chain_out, int32_t* plan_rank, int32_t* plan_which,
                   int32_t* plan_kind, int32_t* n_plan, int32_t* straddler);

/* The message schedule of the image exchange (host-only, no GPU needed): what rank `rank` of `world` sends, receives and copies, in order.
 * The library's own fan-out executes exactly this list (csrc/multi.cpp: send / recv -> grouped ncclSend / ncclRecv or peer copies, all-gather
 * -> ncclAllGather), and the CPU tests execute it over gloo on host buffers (tests/test_fanout_gloo.py) -- one definition of the exchange.
 * Buffers are arrays of UNITS: one unit = one screen piece (ceil(W H / world) pixels, tiles exchange) or one whole padded image (all-gather
 * exchange).  VP_XBUF_PRIMARY: this rank's first partial image in blend order [world units / 1 unit]; VP_XBUF_SECOND: the straddling
 * slab's phase-B image, same shape; VP_XBUF_PIECES: [world + 1] received units, slot r = rank r's, slot world = the straddler's second;
 * VP_XBUF_PIECE_OUT: [1] the unit this rank blends (tiles); VP_XBUF_FINAL: [world] the finished image on rank 0.
 * phase 0 = everything before the ordered blend (the blend reads VP_XBUF_PIECES per vp_blend_plan and writes VP_XBUF_PIECE_OUT, or -- all-gather
 * exchange, rank 0 only -- VP_XBUF_FINAL); phase 1 = after it (gather of the finished pieces on rank 0; empty for the all-gather exchange).
 * Consecutive VP_XOP_SEND / VP_XOP_RECV entries form one batch (issued together, they complete together). */
typedef struct vp_xop {
    int32_t kind;                 /* VP_XOP_*                                                                         */
    int32_t peer;                 /* SEND / RECV: the other rank; otherwise -1                                        */
    int32_t buf, index;           /* SEND / COPY: source unit; RECV: destination unit; ALL_GATHER: buf, own unit      */
    int32_t dst_buf, dst_index;   /* COPY: destination unit                                                           */
} vp_xop;
#define VP_XOP_RECV 0
#define VP_XOP_SEND 1
#define VP_XOP_COPY 2
#define VP_XOP_ALL_GATHER 3       /* in-place all-gather of buf: every rank contributes unit `index` (= its rank)      */
#define VP_XBUF_PRIMARY 0
#define VP_XBUF_SECOND 1
#define VP_XBUF_PIECES 2
#define VP_XBUF_PIECE_OUT 3
#define VP_XBUF_FINAL 4
int  vp_exchange_plan(int32_t world, int32_t rank, int32_t straddler /* -1: none */, int32_t all_gather_exchange, int32_t phase,
                      vp_xop* ops_out, int32_t cap, int32_t* n_out);

/* ---- multi-GPU building blocks (one context per GPU, each owning a contiguous zz slab; what the fan-out is made of) ------------ */
/* Fill split at the only cross-slab dependency, the per-column transmitted light (Fill.shader:224,250):
 *   vp_fill_local   : density/ao of the slab + slab transmittance map tau (computed with T_in = 1)
 *                     written to d_tau_out [(Ny*nv)][(Nx*nv)] f32 (device);
 *   [caller all-gathers tau over RCCL and forms T_in = prod_{slabs before} tau]
 *   vp_fill_finish  : propagate with the true incoming light d_light_in (device, same shape;
 *                     NULL = 1.0) and store the bricks. */
int  vp_fill_local(vp_ctx* ctx, const vp_fill_params* params, void* d_tau_out);
int  vp_fill_finish(vp_ctx* ctx, const void* d_light_in);
/* The same with the product fused in: d_tau_all = the all-gathered transmittance maps [world][(Ny*nv)][(Nx*nv)] f32 (device),
 * this context being slab `rank` of `world`; T_in = tau[0] * tau[1] * ... * tau[rank-1] (slab order) is formed inside the kernel. */
int  vp_fill_finish_gathered(vp_ctx* ctx, const void* d_tau_all, int32_t rank, int32_t world);
/* Partial images of the owned slab: d_over = composite of its MVs drawn in the OVER phase (zz <= zBoundary),
 * d_under = composite of those in the UNDER phase; each [H][W][4] f32 device.  *phase_mask gets bit0 if
 * the OVER image is non-empty, bit1 for UNDER. */
int  vp_raymarch_partial_device(vp_ctx* ctx, const vp_camera* cam, const vp_raymarch_params* params,
                                void* d_over, void* d_under, int32_t* phase_mask);
/* The same with the cross-slab saturation hand-off (the reference's one render target sees every metavoxel, VPR.cs:652-711, so a single GPU
 * stops a ray once it is saturated; a slab alone only knows its own metavoxels):
 * A hand-off map holds ONE BYTE per pixel: code = floor(-8 log2 t), capped at 255, of a transmittance t = 1 - alpha; it decodes to
 * 2^(-code/8) >= t (a conservative bound within a factor 2^(1/8)), and the product of maps is the sum of their codes.
 *   d_t_in   n_in maps [n_in][H][W] u8 (device) of slabs composited IN FRONT of this one; a ray stops once
 *            (1 - dst.a) * prod(t_in) <= 2^-25 (no map: exactly the single-GPU rule).  NULL / 0 = nothing known.
 *   d_t_out0 [H][W] u8: this slab's own map, 1 - alpha of its phase-A image; d_t_out1: that times 1 - alpha of its phase-B image (what a
 *            phase-B slab behind it needs; the two differ only for the slab that straddles zBoundary).  NULL = not wanted.
 * Also accumulates the per-slice sample profile read by vp_read_zsamples. */
int  vp_raymarch_partial_handoff_device(vp_ctx* ctx, const vp_camera* cam, const vp_raymarch_params* params, void* d_over, void* d_under,
                                        int32_t* phase_mask, const void* d_t_in, int32_t n_in, void* d_t_out0, void* d_t_out1);
/* Lattice samples the last vp_raymarch_partial* call executed per light-axis slice zz [Nz] (0 outside the owned slab). */
int  vp_read_zsamples(vp_ctx* ctx, int64_t* samples_per_z);
/* Ordered final blend of gathered partial images (VPR.cs:652-711 restricted to slab granularity):
 * d_partials[i] (device pointers, host array) are applied in the given order, kinds[i] = 0 OVER, 1 UNDER. */
int  vp_blend_partials_device(vp_ctx* ctx, const void* const* d_partials, const int32_t* kinds, int32_t n,
                              void* d_rgba_out);
/* The same blend over num_pixels consecutive pixels only (all pointers already point at the first of them): the blend
 * is per pixel, so N ranks can each finish 1/N of the image after an all-to-all of the partial images' pieces. */
int  vp_blend_partials_range_device(vp_ctx* ctx, const void* const* d_partials, const int32_t* kinds, int32_t n,
                                    void* d_rgba_out, int64_t num_pixels);
/* (particle, metavoxel) pairs per light-axis slice zz over the WHOLE grid (independent of the owned slab): the
 * work histogram from which callers cut balanced slabs.  Needs vp_set_frame + vp_upload_particles; invalidates bins. */
int  vp_z_histogram(vp_ctx* ctx, int64_t* pairs_per_z /* [Nz] */);
/* zBoundary of RenderMetavoxels (VPR.cs:642-648) for this frame/camera. */
int  vp_z_boundary(vp_ctx* ctx, const vp_camera* cam, int32_t* z_boundary);

/* ---- Unity native-plugin hookup ----------------------------------------------------------------------------------
 * The reference does its GPU work from OnPostRender on Unity's main thread through Graphics.* calls (VPR.cs:181-220).  A native plugin works
 * on Unity's RENDER thread: the C# component fills a vp_unity_frame, calls GL.IssuePluginEvent(vp_unity_render_event_func(), slot), and the
 * callback runs [vp_set_frame] -> [vp_bin -> vp_fill] -> vp_raymarch there, writing particlesRT into the output registered for the slot.
 * UnityPluginLoad / UnityPluginUnload are the two names Unity's plugin loader looks up (IUnityInterface.h); the IUnityInterfaces* is kept
 * opaque.  Pointers inside the frame (particles, cube map, depth maps) must stay valid until the event has run; the struct itself is copied. */
#define VP_UNITY_MAX_SLOTS 8
#define VP_UNITY_SET_FRAME    1   /* light / grid moved: UpdateMetavoxelPositions first                  VPR.cs:188-195 */
#define VP_UNITY_BIN_AND_FILL 2   /* this frame re-bins and re-fills (Time.frameCount % updateInterval)  VPR.cs:186     */
typedef struct vp_unity_frame {
    vp_ctx* ctx;
    int32_t flags;                        /* VP_UNITY_* */
    int32_t particle_count;
    float light_to_world[16];
    float grid_center[3];
    float psys_local_to_world[16];
    const void* particles;                /* ParticleSystem.Particle[] (pinned), particle_count records */
    vp_particle_layout layout;
    vp_fill_params fill;
    vp_camera camera;
    vp_raymarch_params raymarch;
} vp_unity_frame;
typedef void (*vp_unity_render_event)(int event_id);
void UnityPluginLoad(void* unity_interfaces);
void UnityPluginUnload(void);
vp_unity_render_event vp_unity_render_event_func(void);            /* pass to GL.IssuePluginEvent with eventId = slot */
int  vp_unity_set_frame_desc(int32_t slot, const vp_unity_frame* frame);
/* Where the event writes particlesRT ([H][W][4] f32 premultiplied): a HIP device pointer (written by the ray-march itself) and / or a host
 * buffer (read back after it).  At least one must be non-NULL when the event runs. */
int  vp_unity_register_output(int32_t slot, void* d_rgba_out, float* h_rgba_out);
/* Texture interop, the native half: `fd` = the memory behind the render texture's linear RGBA32F buffer, exported by the graphics API as a POSIX
 * file descriptor (Vulkan: VkExportMemoryAllocateInfo + vkGetMemoryFdKHR; opaque fd / dma-buf), `bytes` its allocation size.  Imports it with HIP's
 * external-memory API on the context's display device, maps W*H*16 bytes at `offset` and registers the mapping as the slot's DEVICE output (replacing
 * d_rgba_out of vp_unity_register_output; the host output is kept): the ray-march writes the shared memory directly -- what VPR.cs:204-210 does on the
 * GPU, without the PCIe round trip.  HIP owns the fd after a successful import.  vp_unity_clear_slot / UnityPluginUnload drop the import. */
int  vp_unity_register_output_fd(int32_t slot, vp_ctx* ctx, int32_t fd, uint64_t bytes, uint64_t offset);
/* Status of the most recent event of the slot (Unity's callback returns void) and how many events have run. */
int  vp_unity_last_status(int32_t slot, uint64_t* events_run);
/* Detach the slot: waits for an event of the slot that is executing right now, then forgets its frame description and outputs, so that an
 * event Unity delivers LATER is a no-op (status VP_ERR_STATE; the event counter keeps counting).  Call it before freeing anything the frame
 * description points to and before vp_destroy -- the render thread runs an issued event whenever it gets to it.  A host that issues one
 * event per frame should also not touch the arrays of the description (particles, the read-back buffer) while
 * vp_unity_last_status().events_run is behind the number of events it has issued. */
int  vp_unity_clear_slot(int32_t slot);

/* ---- particle source -----------------------------------------------------------------------------------------------
 * The reference's input is a Unity ParticleSystem (scene "Particle System Demo", Assets/Volumetric_Particle_System.unity:2264-2620: cone
 * shape type 4 with angle 10 deg and radius 0.5, 10 particles / s, lifetime 6 s, speed 3, size 4, random start rotation, angular velocity
 * 0.0698 rad/s, at most 60 particles) whose GetParticles array BinParticlesToMetavoxels reads (VPR.cs:412-413).  For hosts without Unity
 * (examples/, bench.py --config DEMO, the tests) the library carries an emitter with those parameters: host code, deterministic (own
 * PCG32 stream -- Unity's is closed source), simulated in the system's LOCAL space like the reference's.  It needs no device and no vp_ctx. */
typedef struct vp_emitter vp_emitter;
typedef struct vp_emitter_config {
    uint64_t seed;
    float rate;                  /* particles per second                      EmissionModule scene:2512-2556 (10)    */
    float lifetime;              /* seconds = startLifetime of every particle InitialModule  scene:2272-2497 (6)     */
    float speed;                 /* start speed along the cone                                               (3)     */
    float size;                  /* start size = DIAMETER (VPR.cs:425)                                        (4)     */
    float cone_angle_deg;        /* ShapeModule type 4                        scene:2498-2511                (10)    */
    float cone_radius;           /* radius of the cone's base disc                                            (0.5)   */
    float angular_velocity_deg;  /* degrees per second                        RotationModule scene:2587-2621 (4)     */
    int32_t max_particles;       /* live particles never exceed this          numParticlesEmitted            (60)    */
    int32_t reserved[6];         /* [0] = 1: prewarm -- the reference's system has `prewarm: 1` (scene:2269) and starts in steady state; vp_emitter_create
                                    then simulates one lifetime in 1/30 s steps (the default config leaves it 0: callers that step the emitter
                                    themselves for `lifetime` seconds, as bench.py --config DEMO does, get the same cloud).  Others 0.   */
} vp_emitter_config;
void vp_emitter_default_config(vp_emitter_config* cfg);        /* the demo scene's values (the numbers in parentheses above), seed 7 */
int  vp_emitter_create(const vp_emitter_config* cfg, vp_emitter** out);
void vp_emitter_destroy(vp_emitter* em);
/* Advance by dt seconds: move / age / retire, then emit what the rate owes.  Returns the live count (>= 0) or a negative vp_status. */
int  vp_emitter_step(vp_emitter* em, float dt);
int  vp_emitter_count(const vp_emitter* em);
/* Write the live particles (oldest first) as records of `layout` (each zeroed, then position, size, rotation -- degrees or radians as the
 * layout says --, lifetime, startLifetime): the array vp_bin / vp_upload_particles take.  Returns the records written (<= capacity). */
int  vp_emitter_write_particles(const vp_emitter* em, void* particles_out, int32_t capacity, const vp_particle_layout* layout);

/* ---- parity probes / stats ------------------------------------------------------------------ */
int  vp_get_mv_positions(vp_ctx* ctx, float* pos_out /* [Nz][Ny][Nx][3] */);
int  vp_read_binlist(vp_ctx* ctx, int32_t xx, int32_t yy, int32_t zz, int32_t* ids, int32_t cap, int32_t* n);
int  vp_read_bincounts(vp_ctx* ctx, int32_t* counts /* [Nz][Ny][Nx] */);
int  vp_read_brick(vp_ctx* ctx, int32_t xx, int32_t yy, int32_t zz, uint16_t* half_rgba /* nv^3*4 */);
int  vp_read_lightmap(vp_ctx* ctx, float* out /* [(Ny*nv)][(Nx*nv)] */);
int  vp_get_stats(vp_ctx* ctx, vp_stats* out);
/* Device time in milliseconds of a stage over its most recent launch, measured with HIP events on the context's stream.  stage: 0 bin,
 * 1 fill (vp_fill / vp_fill_local), 2 raymarch -- since round 4 from in front of the frame preparation (k_rm_prepare: per-cell records, draw-order
 * ranks, tile costs; + the dispatch-order sort) to the end of the march, i.e. the whole stage, not the march kernel alone --, 3 vp_fill_finish. */
int  vp_last_kernel_ms(vp_ctx* ctx, int32_t stage, float* ms);

#ifdef __cplusplus
}
#endif
#endif /* VPFX_H */
