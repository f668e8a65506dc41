/* demo_scene.c -- the reference's own scene, headless, through the C ABI alone: the per-frame driver of
 * VolumetricParticleRenderer.OnPostRender (VPR.cs:181-220) with the library's particle source standing in for Unity's ParticleSystem.
 *   grid 10^3 metavoxels x 32^3 voxels of size 3, border 1, grid centre (0,-5,0)            Volumetric_Particle_System.unity:9013-9016
 *   "Particle System Demo" at (0,0,11.2), rotated 180 deg about Y (cone, 10 particles/s...)  :2264-2620  -> vp_emitter_*
 *   directional light quaternion (0.1856,0,0,0.9826) at (0,0,-44.34)                         :6792
 *   main camera at (-10,0,-20) looking down world +z, fov 60, 1024 x 768                     :8965-8967
 *   the eight Default-layer meshes under "Scene" (0,-5,0): ground / back (cubes scaled (50,1,50)), Cube, Cube 1 and the four cylinders
 *   (:8623, 1755, 8382, 5462) as occluder solids -> both depth inputs rendered by the library  VPR.cs:184, 204  -> vp_set_occluders2
 *   bin + fill every updateInterval = 2 frames, ray-march every frame                        VPR.cs:186,207
 *   cc -std=c99 -Iinclude examples/demo_scene.c -Lvolumetric-particles-for-unity_amd -lvpfx -lm -o demo_scene
 *   demo_scene [frames] [width] [height]
 * Prints, per refill frame, the particle count and what the library binned, and at the end the frame time and an image checksum. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "vpfx.h"

/* ParticleSystem.Particle of Unity 5 (84 bytes); the ABI takes the offsets explicitly */
typedef struct {
    float position[3], velocity[3], animated_velocity[3], axis_of_rotation[3];
    float rotation, angular_velocity, size; uint32_t color, random_seed; float lifetime, start_lifetime, emit_accumulator[2];
} unity_particle;

#define CHECK(call) do { int rc_ = (call); if (rc_ < 0) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, vp_last_error(ctx)); return 2; } } while (0)

static double now_ms(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec * 1e-6; }

/* column-major TRS of a unit quaternion (x, y, z, w) and a position: Matrix4x4.TRS with scale 1 */
static void trs(float m[16], const float q[4], const float p[3])
{
    const float x = q[0], y = q[1], z = q[2], w = q[3];
    const float r[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                        2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                        2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)};
    memset(m, 0, 16 * sizeof(float));
    for (int c = 0; c < 3; ++c) for (int rr = 0; rr < 3; ++rr) m[4 * c + rr] = r[3 * rr + c];
    m[12] = p[0]; m[13] = p[1]; m[14] = p[2]; m[15] = 1.0f;
}

int main(int argc, char** argv)
{
    const int frames = argc > 1 ? atoi(argv[1]) : 60, W = argc > 2 ? atoi(argv[2]) : 1024, H = argc > 3 ? atoi(argv[3]) : 768;
    const int update_interval = 2, S = 32, max_particles = 60;
    vp_ctx* ctx = NULL;
    vp_config cfg; memset(&cfg, 0, sizeof cfg);
    cfg.num_mv[0] = cfg.num_mv[1] = cfg.num_mv[2] = 10; cfg.num_voxels = 32; cfg.num_border = 1; cfg.mv_scale = 3.0f;
    cfg.width = W; cfg.height = H; cfg.device = -1;
    int rc = vp_create(&cfg, &ctx);
    if (rc != VP_OK) { fprintf(stderr, "vp_create -> %d: %s\n", rc, vp_last_error(NULL)); return rc == VP_ERR_NO_DEVICE ? 3 : 2; }

    const float light_q[4] = {0.185593992f, 0, 0, 0.982626557f}, light_p[3] = {0, 0, -44.34f};
    const float psys_q[4] = {0, 1, 0, -4.37113883e-08f}, psys_p[3] = {0, 0, 11.2f};
    float light_to_world[16], psys_to_world[16], grid_center[3] = {0, -5, 0};
    trs(light_to_world, light_q, light_p);
    trs(psys_to_world, psys_q, psys_p);
    CHECK(vp_set_frame(ctx, light_to_world, grid_center));

    /* the opaque scene: Unity primitives (Cube spans [-0.5,0.5]^3, Cylinder = radius 0.5, height 2 about local y) under their transforms */
    static const struct { int type; float px, py, pz, sx, sy, sz; int rot_x_90; } prim[8] = {
        {VP_OCC_BOX,      0.0f,  -1.52f,  0.0f,   50, 1, 50, 0},      /* ground                                  */
        {VP_OCC_BOX,      0.0f,  24.0f,  24.5f,   50, 1, 50, 1},      /* back: quaternion (0.7071,0,0,0.7071)    */
        {VP_OCC_BOX,     -5.34f,  6.18f, -11.89f,  1, 1, 1, 0},       /* Cube                                    */
        {VP_OCC_BOX,      0.0f,   4.57f, -11.0f,   1, 1, 1, 0},       /* Cube 1                                  */
        {VP_OCC_CYLINDER, -8.24f, 0.0f,   0.0f,    1, 1, 1, 0},       /* Cylinder                                */
        {VP_OCC_CYLINDER, 0.0f,   0.0f,   0.0f,    1, 1, 1, 0},       /* Cylinder 1                              */
        {VP_OCC_CYLINDER, 0.0f,   0.0f,   0.0f,    1, 1, 1, 0},       /* Cylinder 2 (coincides with Cylinder 1)  */
        {VP_OCC_CYLINDER, 1.45f, -1.05f,  9.11f,   1, 1, 1, 0}};      /* Cylinder 3                              */
    vp_occluder solids[8]; memset(solids, 0, sizeof solids);
    for (int i = 0; i < 8; ++i) {
        vp_occluder* o = &solids[i];
        o->type = prim[i].type;
        o->center[0] = prim[i].px; o->center[1] = prim[i].py - 5.0f; o->center[2] = prim[i].pz;        /* parent "Scene" at (0,-5,0) */
        /* rows = the local axes in world space; 90 deg about x: local y -> world z, local z -> world -y */
        o->axes[0] = 1.0f;
        if (prim[i].rot_x_90) { o->axes[5] = 1.0f; o->axes[7] = -1.0f; } else { o->axes[4] = 1.0f; o->axes[8] = 1.0f; }
        o->half_extent[0] = 0.5f * prim[i].sx; o->half_extent[2] = 0.5f * prim[i].sz;
        o->half_extent[1] = (prim[i].type == VP_OCC_CYLINDER ? 1.0f : 0.5f) * prim[i].sy;
    }
    CHECK(vp_set_occluders2(ctx, solids, 8));
    {
        const int LW = 10 * 32, LH = 10 * 32;
        float* ld = (float*)malloc(sizeof(float) * LW * LH);
        CHECK(vp_render_light_depth(ctx, 0.3f, 1000.0f, 200.0f, ld));
        int shadowed = 0;
        for (int i = 0; i < LW * LH; ++i) shadowed += ld[i] < 1.0f;
        printf("light_depth_shadowed %d\n", shadowed);
        free(ld);
    }

    vp_emitter* em = NULL;
    vp_emitter_config ec;
    vp_emitter_default_config(&ec);                     /* the scene's ParticleSystem modules */
    CHECK(vp_emitter_create(&ec, &em));
    for (int i = 0; i < 180; ++i) CHECK(vp_emitter_step(em, 1.0f / 30.0f));     /* 6 s: the plume has reached its steady state */
    unity_particle* parts = (unity_particle*)calloc((size_t)max_particles, sizeof(unity_particle));
    vp_particle_layout lay; memset(&lay, 0, sizeof lay);
    lay.stride = (int32_t)sizeof(unity_particle);
    lay.off_position = 0; lay.off_rotation = 48; lay.off_size = 56; lay.off_lifetime = 68; lay.off_start_lifetime = 72;

    /* displacement cube map: smooth procedural noise in [0.1, 0.95] (the reference's asset is not shipped) */
    float* cube = (float*)malloc(sizeof(float) * 6 * S * S);
    for (int f = 0; f < 6; ++f) for (int y = 0; y < S; ++y) for (int x = 0; x < S; ++x)
        cube[(f * S + y) * S + x] = 0.525f + 0.425f * sinf(0.7f * x + 1.3f * f) * cosf(0.5f * y - 0.9f * f);
    vp_fill_params fp; memset(&fp, 0, sizeof fp);
    fp.opacity_factor = 0.04f; fp.displacement_scale = 0.7f; fp.ambient[0] = fp.ambient[1] = fp.ambient[2] = 0.2f;   /* scene:9017-9026 */
    fp.init_light_intensity = 1.0f; fp.light_near = 0.3f; fp.light_far = 1000.0f; fp.light_cam_distance = 200.0f;
    fp.cubemap_size = S; fp.cubemap = cube;

    vp_camera cam; memset(&cam, 0, sizeof cam);
    const float c2w[16] = {1,0,0,0, 0,1,0,0, 0,0,-1,0, -10,0,-20,1};      /* Unity's camera looks down its +z; view space looks down -z */
    const float w2c[16] = {1,0,0,0, 0,1,0,0, 0,0,-1,0, 10,0,-20,1};       /* [R | -R t], R = diag(1,1,-1) */
    memcpy(cam.camera_to_world, c2w, sizeof c2w); memcpy(cam.world_to_camera, w2c, sizeof w2c);
    cam.cam_pos[0] = -10; cam.cam_pos[2] = -20; cam.fov_y = 60.0f * 3.14159265358979f / 180.0f; cam.near_clip = 0.3f; cam.far_clip = 1000.0f;
    vp_raymarch_params rp; memset(&rp, 0, sizeof rp);
    rp.steps_per_mv = 64; rp.soft_distance = 20;
    float* img = (float*)malloc(sizeof(float) * 4 * (size_t)W * H);

    double t0 = 0;
    for (int frame = 0; frame < frames; ++frame) {
        if (frame == 1) t0 = now_ms();                              /* frame 0 pays the allocations */
        CHECK(vp_emitter_step(em, 1.0f / 30.0f));
        if (frame % update_interval == 0) {                         /* VPR.cs:186 */
            const int n = vp_emitter_write_particles(em, parts, max_particles, &lay);
            CHECK(n);
            CHECK(vp_bin(ctx, parts, n, &lay, psys_to_world));
            CHECK(vp_fill(ctx, &fp));
            if (frame % 20 == 0) {
                vp_stats st; CHECK(vp_get_stats(ctx, &st));
                printf("frame %d: particles %d occupied_mv %lld pairs %lld\n", frame, n, (long long)st.occupied_mv, (long long)st.pairs);
            }
        }
        CHECK(vp_raymarch(ctx, &cam, &rp, img));                    /* VPR.cs:207 */
    }
    const double ms = frames > 1 ? (now_ms() - t0) / (frames - 1) : 0.0;
    double s_a = 0, s_rgb = 0;
    for (int i = 0; i < W * H; ++i) { s_rgb += img[4 * i] + img[4 * i + 1] + img[4 * i + 2]; s_a += img[4 * i + 3]; }
    printf("frames %d, %.3f ms per frame (host-buffer entry points, image copied back every frame)\n", frames, ms);
    printf("sum_rgb %.4f sum_alpha %.4f covered %.4f\n", s_rgb, s_a, s_a / ((double)W * H));
    vp_emitter_destroy(em);
    vp_destroy(ctx);
    free(parts); free(cube); free(img);
    return 0;
}
