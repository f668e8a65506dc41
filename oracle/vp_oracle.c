/*
 * vp_oracle.c -- CPU restatement (plain C, fp32) of the reference's sparse-volumetric-particle hot path.
 *
 *   *** TEST INFRASTRUCTURE ONLY ***  Nothing in the product (libvpfx, the package, bench.py's timed GPU
 *   region) may link, import or call this file.  It is the parity checker used by tests/, by
 *   __graft_entry__.smoke() and by bench.py's `cpu_baseline` leg.
 *
 *   *** PARITY UNPINNED ***  The reference (rajabala/Volumetric-Particles-For-Unity) ships no tests, golden
 *   images or known-answer vectors, and neither its C# (needs closed-source UnityEngine.dll, no mono/dotnet
 *   here) nor its HLSL (ShaderLab + D3D11 fixed-function units, no dxc/fxc here) can be built or run in this
 *   container.  This file is therefore a restatement written from the reference's source text; it is pinned
 *   only by (1) an independent float64 numpy twin (oracle/numpy_twin.py), (2) analytic known-answer tests
 *   (tests/test_oracle_kat.py), (3) committed golden fixtures generated from (1)+(this file), and (4) the
 *   one external anchor there is: eight pixels + whole-frame statistics of config C1 recorded in SURVEY.md
 *   App. C from the surveyor's own independent float64 probe of the shaders, rendered with the reference's
 *   displacement cubemap asset -- reproduced by this file to 6e-6 (tests/golden/survey_anchors_C1.npz).
 *   None of these is an output of the reference itself, hence "unpinned".
 *
 * Reference files restated (paths relative to the reference root):
 *   VPR.cs     = Assets/Main Scene/VolumetricParticleRenderer.cs
 *   MathUtil   = Assets/Main Scene/MathUtil.cs
 *   Fill       = Assets/Shaders/Metavoxel/FillVolume.shader
 *   RM         = Assets/Shaders/Metavoxel/RayMarchVoxel.shader
 *   Comp       = Assets/Shaders/Metavoxel/CompositeParticles.shader
 *
 * Arithmetic: fp32 throughout, no contraction (-ffp-contract=off); fused multiply-adds appear only where
 * written as fmaf().  Unity engine calls whose arithmetic is closed source (Matrix4x4.TRS/.inverse,
 * Quaternion.AngleAxis, Transform matrices) are restated from their documented definitions; the exact
 * operation order chosen here is the "arithmetic spec" of DESIGN.md section 4, which the HIP kernels
 * follow for every decision-critical quantity (bin acceptance, voxel-in-sphere hit test).
 *
 * The public structs come from include/vpfx.h (the interface, not product code).
 */
#include "../include/vpfx.h"

#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define VPO_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------------ */
/* small math helpers                                                                               */
/* ------------------------------------------------------------------------------------------------ */
typedef struct { float x, y, z; } v3;

static inline v3 v3_make(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 v3_add(v3 a, v3 b) { return v3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v3_sub(v3 a, v3 b) { return v3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v3_scale(v3 a, float s) { return v3_make(a.x * s, a.y * s, a.z * s); }
static inline float v3_dot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }

/* column-major 4x4 element (row r, column c) */
#define M(m, r, c) ((m)[(c) * 4 + (r)])

/* Matrix4x4.MultiplyPoint3x4: res.x = m00*x + m01*y + m02*z + m03 (left to right). */
static inline v3 mul_point(const float* m, v3 v)
{
    v3 r;
    r.x = ((M(m, 0, 0) * v.x + M(m, 0, 1) * v.y) + M(m, 0, 2) * v.z) + M(m, 0, 3);
    r.y = ((M(m, 1, 0) * v.x + M(m, 1, 1) * v.y) + M(m, 1, 2) * v.z) + M(m, 1, 3);
    r.z = ((M(m, 2, 0) * v.x + M(m, 2, 1) * v.y) + M(m, 2, 2) * v.z) + M(m, 2, 3);
    return r;
}
static inline v3 mul_dir(const float* m, v3 v)
{
    v3 r;
    r.x = (M(m, 0, 0) * v.x + M(m, 0, 1) * v.y) + M(m, 0, 2) * v.z;
    r.y = (M(m, 1, 0) * v.x + M(m, 1, 1) * v.y) + M(m, 1, 2) * v.z;
    r.z = (M(m, 2, 0) * v.x + M(m, 2, 1) * v.y) + M(m, 2, 2) * v.z;
    return r;
}
/* C = A * B (4x4, column-major), each element a left-to-right dot product (Matrix4x4.operator*). */
static void mat_mul(const float* A, const float* B, float* C)
{
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r)
            M(C, r, c) = ((M(A, r, 0) * M(B, 0, c) + M(A, r, 1) * M(B, 1, c)) + M(A, r, 2) * M(B, 2, c)) +
                         M(A, r, 3) * M(B, 3, c);
}
/* inverse of TRS(t, R, (sc,sc,sc)) with R orthonormal: linear = R^T / sc, translation = -(linear * t). */
static void trs_inverse(v3 t, const float R[9] /* row-major 3x3: R[r*3+c] */, float sc, float* out)
{
    float inv = 1.0f / sc;
    memset(out, 0, 16 * sizeof(float));
    for (int k = 0; k < 3; ++k) {
        float rx = R[0 * 3 + k] * inv, ry = R[1 * 3 + k] * inv, rz = R[2 * 3 + k] * inv;
        M(out, k, 0) = rx; M(out, k, 1) = ry; M(out, k, 2) = rz;
        M(out, k, 3) = -((rx * t.x + ry * t.y) + rz * t.z);
    }
    M(out, 3, 3) = 1.0f;
}

/* Mathf.RoundToInt = Math.Round = round-half-to-even (default FP rounding mode). */
static inline int round_to_int(float x) { return (int)rintf(x); }

/* sin/cos of an angle in DEGREES, arithmetic spec: exact reduction by quarter turns, then
 * cephes-style minimax polynomials on [-pi/4, pi/4], evaluated with fmaf only so that CPU and GPU agree
 * bit for bit.  Restates what Quaternion.AngleAxis needs (VPR.cs:583); Unity's own sincos is closed source. */
static void sincos_deg(float deg, float* s_out, float* c_out)
{
    float k = rintf(deg * (1.0f / 90.0f));
    float r = fmaf(-90.0f, k, deg);            /* [-45, 45] */
    float x = r * 0.017453292519943295f;       /* radians   */
    float z = x * x;
    float sp = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f), z * x, x);
    float cp = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f),
                    z * z, fmaf(-0.5f, z, 1.0f));
    int q = ((int)k) & 3;
    float s, c;
    switch (q) {
    case 0: s = sp; c = cp; break;
    case 1: s = cp; c = -sp; break;
    case 2: s = -sp; c = -cp; break;
    default: s = -cp; c = sp; break;
    }
    *s_out = s; *c_out = c;
}

/* IEEE binary16 conversions (typed UAV store to R16G16B16A16_FLOAT; round-to-nearest-even assumed). */
static uint16_t f32_to_f16(float f)
{
    uint32_t x; memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t em = x & 0x7fffffffu;
    if (em >= 0x7f800000u) return (uint16_t)(sign | (em > 0x7f800000u ? 0x7e00u : 0x7c00u));
    if (em >= 0x477ff000u) {            /* >= 65520 rounds to inf */
        return (uint16_t)(sign | 0x7c00u);
    }
    if (em < 0x38800000u) {             /* subnormal half or zero */
        if (em < 0x33000000u) return (uint16_t)sign;   /* < 2^-25 -> 0 */
        uint32_t e = em >> 23;
        uint32_t m = (em & 0x7fffffu) | 0x800000u;
        uint32_t shift = 126u - e;      /* 14..24 */
        uint32_t h = m >> shift;
        uint32_t rem = m & ((1u << shift) - 1u);
        uint32_t half = 1u << (shift - 1u);
        if (rem > half || (rem == half && (h & 1u))) h++;
        return (uint16_t)(sign | h);
    }
    uint32_t h = ((em - 0x38000000u) >> 13);
    uint32_t rem = em & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) h++;
    return (uint16_t)(sign | h);
}
static float f16_to_f32(uint16_t h)
{
    uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1fu, m = h & 0x3ffu, x;
    if (e == 0) {
        if (m == 0) x = sign;
        else {
            int sh = 0;
            while (!(m & 0x400u)) { m <<= 1; sh++; }
            m &= 0x3ffu;
            x = sign | ((uint32_t)(113 - sh) << 23) | (m << 13);
        }
    } else if (e == 31) x = sign | 0x7f800000u | (m << 13);
    else x = sign | ((e + 112u) << 23) | (m << 13);
    float f; memcpy(&f, &x, 4);
    return f;
}

/* ------------------------------------------------------------------------------------------------ */
/* context                                                                                          */
/* ------------------------------------------------------------------------------------------------ */
typedef struct vpo_ctx {
    vp_config cfg;
    int Nx, Ny, Nz, nv, b, W, H, z0, z1;
    float s, sb, one;
    int literal;                 /* 1: literal reference voxel stepping (Fill.shader:183,207), 0: spec closed form */
    int threads;
    /* frame */
    int have_frame;
    float L[16], Linv[16], Rl[9], gc[3];
    v3 lsO, fwd;
    float* mvPos;                /* [N^3][3] */
    /* particles */
    int P;
    float* ws;                   /* [P][3] world position */
    float* psize;                /* diameter */
    float* rec;                  /* [P][16]: rows k=0..2 -> (rx,ry,rz,t), [12]=opacity, [13]=radius */
    /* bins (CSR) */
    int* offsets;                /* [N^3+1] */
    int* ids;
    long pairs;
    /* fill products */
    int* brick_index;            /* [N^3] -> brick or -1 (only slab MVs get bricks) */
    int occupied;
    uint16_t* bricks;            /* [occupied][nv][nv][nv][4] */
    float* dens_ao;              /* scratch of vpo_fill_local: [occupied][nv^3][2] */
    vp_fill_params fp;           /* retained between fill_local and fill_finish */
    float* cubemap; int cube_s;
    float* depthmap;
    float* lightmap;             /* [(Ny*nv)][(Nx*nv)] */
    int filled;
    long samples;
    float* f16lut;               /* 65536 floats */
    vp_occluder* occluders; int n_occluders;
    char err[256];
} vpo_ctx;

static char g_err[256];

VPO_API const char* vpo_last_error(const vpo_ctx* c) { return c ? c->err : g_err; }

VPO_API int vpo_create(const vp_config* cfg, vpo_ctx** out)
{
    if (!cfg || !out) { snprintf(g_err, sizeof g_err, "null argument"); return VP_ERR_BAD_ARG; }
    if (cfg->num_mv[0] < 1 || cfg->num_mv[1] < 1 || cfg->num_mv[2] < 1 || cfg->num_voxels < 2 ||
        cfg->num_border < 0 || 2 * cfg->num_border >= cfg->num_voxels || !(cfg->mv_scale > 0.f)) {
        snprintf(g_err, sizeof g_err, "bad grid configuration"); return VP_ERR_BAD_ARG;
    }
    vpo_ctx* c = (vpo_ctx*)calloc(1, sizeof *c);
    if (!c) return VP_ERR_OOM;
    c->cfg = *cfg;
    c->Nx = cfg->num_mv[0]; c->Ny = cfg->num_mv[1]; c->Nz = cfg->num_mv[2];
    c->nv = cfg->num_voxels; c->b = cfg->num_border; c->W = cfg->width; c->H = cfg->height;
    c->z0 = cfg->slab_z0; c->z1 = cfg->slab_z1;
    if (c->z0 == 0 && c->z1 == 0) c->z1 = c->Nz;
    if (c->z0 < 0 || c->z1 > c->Nz || c->z0 >= c->z1) { free(c); snprintf(g_err, sizeof g_err, "bad slab"); return VP_ERR_BAD_ARG; }
    c->s = cfg->mv_scale;
    /* mvScaleWithBorder = mvScale * nv / (nv - 2b)                                   VPR.cs:139 */
    c->sb = c->s * (float)c->nv / (float)(c->nv - 2 * c->b);
    /* oneVoxelSize = _MetavoxelScaleZ / _NumVoxels                                   Fill.shader:160 */
    c->one = c->sb / (float)c->nv;
    size_t n3 = (size_t)c->Nx * c->Ny * c->Nz;
    c->mvPos = (float*)calloc(n3 * 3, sizeof(float));
    c->offsets = (int*)calloc(n3 + 1, sizeof(int));
    c->brick_index = (int*)malloc(n3 * sizeof(int));
    c->lightmap = (float*)calloc((size_t)c->Nx * c->nv * c->Ny * c->nv, sizeof(float));
    c->f16lut = (float*)malloc(65536 * sizeof(float));
    if (!c->mvPos || !c->offsets || !c->brick_index || !c->lightmap || !c->f16lut) return VP_ERR_OOM;
    for (size_t i = 0; i < n3; ++i) c->brick_index[i] = -1;
    for (int i = 0; i < 65536; ++i) c->f16lut[i] = f16_to_f32((uint16_t)i);
    c->threads = 0;
    *out = c;
    return VP_OK;
}

VPO_API void vpo_destroy(vpo_ctx* c)
{
    if (!c) return;
    free(c->mvPos); free(c->ws); free(c->psize); free(c->rec); free(c->offsets); free(c->ids);
    free(c->brick_index); free(c->bricks); free(c->dens_ao); free(c->cubemap); free(c->depthmap);
    free(c->lightmap); free(c->f16lut); free(c->occluders); free(c);
}

VPO_API int vpo_set_mode(vpo_ctx* c, int literal_stepping) { c->literal = literal_stepping; return VP_OK; }
VPO_API int vpo_set_threads(vpo_ctx* c, int n) { c->threads = n; return VP_OK; }
VPO_API int vpo_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
static int nthreads(const vpo_ctx* c)
{
#ifdef _OPENMP
    return c->threads > 0 ? c->threads : omp_get_max_threads();
#else
    (void)c; return 1;
#endif
}

static inline size_t mv_index(const vpo_ctx* c, int xx, int yy, int zz) { return ((size_t)zz * c->Ny + yy) * c->Nx + xx; }

/* ------------------------------------------------------------------------------------------------ */
/* F2: UpdateMetavoxelPositions                                                       VPR.cs:370-394 */
/* ------------------------------------------------------------------------------------------------ */
VPO_API int vpo_set_frame(vpo_ctx* c, const float L[16], const float gc[3])
{
    if (!c || !L || !gc) return VP_ERR_BAD_ARG;
    memcpy(c->L, L, sizeof c->L);
    memcpy(c->gc, gc, sizeof c->gc);
    for (int r = 0; r < 3; ++r) for (int k = 0; k < 3; ++k) c->Rl[r * 3 + k] = M(L, r, k);
    /* dirLight.transform.worldToLocalMatrix of a rigid transform */
    trs_inverse(v3_make(M(L, 0, 3), M(L, 1, 3), M(L, 2, 3)), c->Rl, 1.0f, c->Linv);
    /* lsWorldOrigin = worldToLocal.MultiplyPoint3x4(wsGridCenter)                    VPR.cs:380 */
    c->lsO = mul_point(c->Linv, v3_make(gc[0], gc[1], gc[2]));
    /* dirLight.transform.forward.normalized                                          VPR.cs:535 */
    v3 f = v3_make(c->Rl[0 * 3 + 2], c->Rl[1 * 3 + 2], c->Rl[2 * 3 + 2]);
    float mag = sqrtf(v3_dot(f, f));
    c->fwd = (mag > 1e-5f) ? v3_make(f.x / mag, f.y / mag, f.z / mag) : v3_make(0, 0, 0);
    for (int zz = 0; zz < c->Nz; ++zz)
        for (int yy = 0; yy < c->Ny; ++yy)
            for (int xx = 0; xx < c->Nx; ++xx) {
                /* lsOffset = Scale((Nx/2 - xx, Ny/2 - yy, Nz/2 - zz), mvScale), INTEGER N/2  VPR.cs:388 */
                v3 off = v3_make((float)(c->Nx / 2 - xx) * c->s, (float)(c->Ny / 2 - yy) * c->s, (float)(c->Nz / 2 - zz) * c->s);
                v3 p = mul_point(c->L, v3_sub(c->lsO, off));                         /* VPR.cs:389 */
                float* d = c->mvPos + 3 * mv_index(c, xx, yy, zz);
                d[0] = p.x; d[1] = p.y; d[2] = p.z;
            }
    c->have_frame = 1;
    c->filled = 0;
    return VP_OK;
}

VPO_API int vpo_get_mv_positions(vpo_ctx* c, float* out)
{
    if (!c || !out || !c->have_frame) return VP_ERR_STATE;
    memcpy(out, c->mvPos, (size_t)c->Nx * c->Ny * c->Nz * 3 * sizeof(float));
    return VP_OK;
}

/* ------------------------------------------------------------------------------------------------ */
/* F3/F4: BinParticlesToMetavoxels + DoesBoxIntersectSphere          VPR.cs:397-457, MathUtil:11-25 */
/* ------------------------------------------------------------------------------------------------ */
static inline float rd_f32(const unsigned char* p) { float f; memcpy(&f, p, 4); return f; }

/* per-particle record used by the fill (VPR.cs:580-586):
 *   mWorldToLocal = TRS(wsPos, AngleAxis(p.rotation, psys.forward), (size,size,size)).inverse
 *   mOpacity = lifetime / startLifetime */
static void build_particle_record(const vpo_ctx* c, v3 ws, float size, float rot_deg, float opacity, v3 axis, float* rec)
{
    (void)c;
    float sh, ch;
    sincos_deg(rot_deg * 0.5f, &sh, &ch);
    float qx = axis.x * sh, qy = axis.y * sh, qz = axis.z * sh, qw = ch;
    float x2 = qx + qx, y2 = qy + qy, z2 = qz + qz;
    float xx = qx * x2, yy = qy * y2, zz = qz * z2, xy = qx * y2, xz = qx * z2, yz = qy * z2;
    float wx = qw * x2, wy = qw * y2, wz = qw * z2;
    float R[9];
    R[0] = 1.0f - (yy + zz); R[1] = xy - wz;          R[2] = xz + wy;
    R[3] = xy + wz;          R[4] = 1.0f - (xx + zz); R[5] = yz - wx;
    R[6] = xz - wy;          R[7] = yz + wx;          R[8] = 1.0f - (xx + yy);
    float inv = 1.0f / size;
    for (int k = 0; k < 3; ++k) {
        float rx = R[0 * 3 + k] * inv, ry = R[1 * 3 + k] * inv, rz = R[2 * 3 + k] * inv;
        rec[k * 4 + 0] = rx; rec[k * 4 + 1] = ry; rec[k * 4 + 2] = rz;
        rec[k * 4 + 3] = -((rx * ws.x + ry * ws.y) + rz * ws.z);
    }
    rec[12] = opacity;
    rec[13] = size * 0.5f;
    rec[14] = 0.f; rec[15] = 0.f;
}

typedef struct { int lo[3], hi[3]; } cand_range;

/* candidate MV range of one particle                                                 VPR.cs:418-432 */
static cand_range particle_candidates(const vpo_ctx* c, v3 ws, float size)
{
    /* A particle with a non-finite position or size is skipped.  The reference's behaviour for it is undefined (C# (int)NaN in
     * :434-438 indexes the MV grid with int.MinValue); skipping is the defined behaviour of this implementation (vpfx.h). */
    if (!(isfinite(ws.x) && isfinite(ws.y) && isfinite(ws.z) && isfinite(size))) {
        cand_range none = { {0, 0, 0}, {-1, -1, -1} };
        return none;
    }
    v3 ls = mul_point(c->Linv, ws);                                  /* lsParticlePos  :419 */
    v3 pio = v3_make((ls.x - c->lsO.x) / c->s, (ls.y - c->lsO.y) / c->s, (ls.z - c->lsO.z) / c->s);  /* :422 */
    v3 pi = v3_make(pio.x + (float)c->Nx * 0.5f, pio.y + (float)c->Ny * 0.5f, pio.z + (float)c->Nz * 0.5f); /* :423 */
    int e = round_to_int((size / 2.0f) / c->s);                      /* pExtents       :425 */
    float fe = (float)e;
    float lo[3] = { fmaxf(0.f, pi.x - fe), fmaxf(0.f, pi.y - fe), fmaxf(0.f, pi.z - fe) };   /* Vector3.Max :431 */
    float hi[3] = { fminf((float)(c->Nx - 1), pi.x + fe), fminf((float)(c->Ny - 1), pi.y + fe), fminf((float)(c->Nz - 1), pi.z + fe) };
    cand_range r;
    for (int k = 0; k < 3; ++k) { r.lo[k] = (int)lo[k]; r.hi[k] = (int)hi[k]; }   /* C# (int) truncates toward zero :434-438 */
    return r;
}

/* exact sphere / bordered-box test of one candidate                     VPR.cs:440-450, MathUtil:11-25 */
static int particle_hits_mv(const vpo_ctx* c, const float rows[3][3] /* Rl^T / sb */, v3 ws, float size, const float* mvp)
{
    float m[3];
    for (int k = 0; k < 3; ++k) {
        float t = -((rows[k][0] * mvp[0] + rows[k][1] * mvp[1]) + rows[k][2] * mvp[2]);
        m[k] = ((rows[k][0] * ws.x + rows[k][1] * ws.y) + rows[k][2] * ws.z) + t;
    }
    float r = (size / 2.0f) / c->sb;                                /* mvParticleRadius :445 */
    float r2 = r * r;
    for (int k = 0; k < 3; ++k) {
        if (m[k] < -0.5f) { float d = m[k] - (-0.5f); r2 -= d * d; }
        else if (m[k] > 0.5f) { float d = m[k] - 0.5f; r2 -= d * d; }
    }
    return r2 > 0.f;
}

VPO_API int vpo_bin(vpo_ctx* c, const void* particles, int count, const vp_particle_layout* lay, const float psysL2W[16])
{
    if (!c || (!particles && count > 0) || !lay || !psysL2W || count < 0) return VP_ERR_BAD_ARG;
    if (!c->have_frame) { snprintf(c->err, sizeof c->err, "vpo_bin before vpo_set_frame"); return VP_ERR_STATE; }
    free(c->ws); free(c->psize); free(c->rec); free(c->ids);
    c->P = count;
    c->ws = (float*)malloc((size_t)(count + 1) * 3 * sizeof(float));
    c->psize = (float*)malloc((size_t)(count + 1) * sizeof(float));
    c->rec = (float*)malloc((size_t)(count + 1) * 16 * sizeof(float));
    /* particleSys.transform.forward = R*(0,0,1), normalised (Quaternion.AngleAxis normalises its axis) */
    v3 ax = v3_make(M(psysL2W, 0, 2), M(psysL2W, 1, 2), M(psysL2W, 2, 2));
    float am = sqrtf(v3_dot(ax, ax));
    ax = (am > 0.f) ? v3_make(ax.x / am, ax.y / am, ax.z / am) : v3_make(0, 0, 1);
    const unsigned char* base = (const unsigned char*)particles;
    for (int p = 0; p < count; ++p) {
        const unsigned char* q = base + (size_t)p * lay->stride;
        v3 lp = v3_make(rd_f32(q + lay->off_position), rd_f32(q + lay->off_position + 4), rd_f32(q + lay->off_position + 8));
        v3 ws = mul_point(psysL2W, lp);                              /* wsParticlePos :418 */
        float size = rd_f32(q + lay->off_size);
        float rot = rd_f32(q + lay->off_rotation);
        if (lay->rotation_in_radians) rot = rot * 57.29577951308232f;
        float life = rd_f32(q + lay->off_lifetime), life0 = rd_f32(q + lay->off_start_lifetime);
        c->ws[3 * p] = ws.x; c->ws[3 * p + 1] = ws.y; c->ws[3 * p + 2] = ws.z;
        c->psize[p] = size;
        build_particle_record(c, ws, size, rot, life / life0, ax, c->rec + 16 * (size_t)p);
    }
    size_t n3 = (size_t)c->Nx * c->Ny * c->Nz;
    float rows[3][3];
    float invsb = 1.0f / c->sb;
    for (int k = 0; k < 3; ++k) for (int j = 0; j < 3; ++j) rows[k][j] = c->Rl[j * 3 + k] * invsb;
    int* cnt = (int*)calloc(n3 + 1, sizeof(int));
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1) {
            c->offsets[0] = 0;
            for (size_t i = 0; i < n3; ++i) { c->offsets[i + 1] = c->offsets[i] + cnt[i]; cnt[i] = 0; }
            c->pairs = c->offsets[n3];
            c->ids = (int*)malloc((size_t)(c->pairs + 1) * sizeof(int));
        }
        for (int p = 0; p < count; ++p) {                            /* ascending particle index (list order, :452) */
            v3 ws = v3_make(c->ws[3 * p], c->ws[3 * p + 1], c->ws[3 * p + 2]);
            cand_range r = particle_candidates(c, ws, c->psize[p]);
            for (int zz = r.lo[2]; zz <= r.hi[2]; ++zz)
                for (int yy = r.lo[1]; yy <= r.hi[1]; ++yy)
                    for (int xx = r.lo[0]; xx <= r.hi[0]; ++xx) {
                        size_t mi = mv_index(c, xx, yy, zz);
                        if (!particle_hits_mv(c, rows, ws, c->psize[p], c->mvPos + 3 * mi)) continue;
                        if (pass == 1) c->ids[c->offsets[mi] + cnt[mi]] = p;
                        cnt[mi]++;
                    }
        }
    }
    free(cnt);
    c->filled = 0;
    return VP_OK;
}

VPO_API int vpo_read_bincounts(vpo_ctx* c, int* counts)
{
    size_t n3 = (size_t)c->Nx * c->Ny * c->Nz;
    for (size_t i = 0; i < n3; ++i) counts[i] = c->offsets[i + 1] - c->offsets[i];
    return VP_OK;
}
VPO_API int vpo_read_binlist(vpo_ctx* c, int xx, int yy, int zz, int* ids, int cap, int* n)
{
    size_t mi = mv_index(c, xx, yy, zz);
    int cnt = c->offsets[mi + 1] - c->offsets[mi];
    *n = cnt;
    for (int i = 0; i < cnt && i < cap; ++i) ids[i] = c->ids[c->offsets[mi] + i];
    return VP_OK;
}
/* per-particle fill records [P][16] (parity probe for the DisplacedParticle build, VPR.cs:575-588) */
VPO_API int vpo_read_particle_records(vpo_ctx* c, float* out) { memcpy(out, c->rec, (size_t)c->P * 16 * sizeof(float)); return VP_OK; }

/* ------------------------------------------------------------------------------------------------ */
/* F9: compute_voxel_color                                                         Fill.shader:110-135 */
/* ------------------------------------------------------------------------------------------------ */
/* texCUBE(_DisplacementTexture, dir).x : D3D cube face selection + per-face bilinear, clamp, no mips. */
static float sample_cubemap(const float* cube, int S, float dx, float dy, float dz)
{
    float ax = fabsf(dx), ay = fabsf(dy), az = fabsf(dz);
    int face; float ma, sc, tc;
    /* major axis: D3D leaves ties implementation-defined; the spec resolves them as the GCN/CDNA cube
     * instructions do (z before y before x), probed on gfx950 (scripts/probes/cube_probe.hip). */
    if (az >= ax && az >= ay) { ma = az; if (dz >= 0.f) { face = 4; sc = dx; tc = -dy; } else { face = 5; sc = -dx; tc = -dy; } }
    else if (ay >= ax)        { ma = ay; if (dy >= 0.f) { face = 2; sc = dx; tc = dz; } else { face = 3; sc = dx; tc = -dz; } }
    else                      { ma = ax; if (dx >= 0.f) { face = 0; sc = -dz; tc = -dy; } else { face = 1; sc = dz; tc = -dy; } }
    float u, v;
    if (ma > 0.f) { float inv = 1.0f / ma; u = sc * inv; v = tc * inv; } else { u = 0.f; v = 0.f; }
    float hs = 0.5f * (float)S;
    float fx = fmaf(u, hs, hs - 0.5f), fy = fmaf(v, hs, hs - 0.5f);
    float x0 = floorf(fx), y0 = floorf(fy);
    float tx = fx - x0, ty = fy - y0;
    int ix0 = (int)x0, iy0 = (int)y0, ix1 = ix0 + 1, iy1 = iy0 + 1;
    if (ix0 < 0) ix0 = 0; if (ix0 > S - 1) ix0 = S - 1;
    if (ix1 < 0) ix1 = 0; if (ix1 > S - 1) ix1 = S - 1;
    if (iy0 < 0) iy0 = 0; if (iy0 > S - 1) iy0 = S - 1;
    if (iy1 < 0) iy1 = 0; if (iy1 > S - 1) iy1 = S - 1;
    const float* f = cube + (size_t)face * S * S;
    float t00 = f[iy0 * S + ix0], t10 = f[iy0 * S + ix1], t01 = f[iy1 * S + ix0], t11 = f[iy1 * S + ix1];
    float a = fmaf(tx, t10 - t00, t00), bb = fmaf(tx, t11 - t01, t01);
    return fmaf(ty, bb - a, a);
}

static inline float saturate(float x) { return fminf(fmaxf(x, 0.f), 1.f); }

static inline void voxel_color(const vpo_ctx* c, float psx, float psy, float psz, float dist2, float opacity, float* density, float* ao)
{
    float raw = sample_cubemap(c->cubemap, c->cube_s, psx, psy, psz);      /* texCUBE(.., 2*ps).x       :116 */
    float D = c->fp.displacement_scale;
    float net = fmaf(D, raw, 1.0f - D);                                     /* netDisplacement           :119 */
    float d2 = 4.0f * dist2;                                                /* dot(2ps,2ps)              :121 */
    float t = saturate((d2 - net) / (0.7f * net - net));                    /* smoothstep(net,0.7net,d2) :126 */
    float base = t * t * (3.0f - 2.0f * t);
    float den = base * c->fp.opacity_factor;                                /* :127 */
    if (c->fp.fade_out_particles == 1) den *= opacity;                      /* :130-131 */
    *density = den; *ao = net;
}

/* ------------------------------------------------------------------------------------------------ */
/* Scene-occlusion inputs from occluder boxes.  Restates what Unity's rasteriser produces */
/* for boxes: light depth map = nearest BACK face under the ortho light camera (LDM.shader:6 Cull Front,      */
/* VPR.cs:184,320-367); eye depth = nearest front face of the main camera's opaque pass (VPR.cs:204).        */
/* ------------------------------------------------------------------------------------------------ */
static int ray_obb(const vp_obb* b, v3 o, v3 d, float* t0, float* t1)
{
    v3 p = v3_sub(o, v3_make(b->center[0], b->center[1], b->center[2]));
    *t0 = -3.0e38f; *t1 = 3.0e38f;
    for (int k = 0; k < 3; ++k) {
        v3 a = v3_make(b->axes[3 * k], b->axes[3 * k + 1], b->axes[3 * k + 2]);
        float lo = v3_dot(a, p), ld = v3_dot(a, d), h = b->half_extent[k];
        if (ld != 0.f) {
            float inv = 1.0f / ld;
            float ta = (-h - lo) * inv, tb = (h - lo) * inv;
            *t0 = fmaxf(*t0, fminf(ta, tb));
            *t1 = fminf(*t1, fmaxf(ta, tb));
        } else if (lo < -h || lo > h) return 0;
    }
    return *t0 <= *t1;
}

/* Capped cylinder / ellipsoid (ABI 6: the reference scene's four cylinders, scene:1755,5462,8382,8623, are drawn into both depth   */
/* inputs like every Default-layer mesh, VPR.cs:184, LDM.shader:6-31).  The ray is taken into the solid's normalised local space  */
/* l = diag(1/h) A (p - c) -- an affine map, so the ray parameter t is unchanged -- where the solid is the unit cylinder about y  */
/* (x^2 + z^2 <= 1, |y| <= 1) or the unit ball.  The quadric is solved about the ray's point of closest approach to the axis /   */
/* centre (foot point), not about the far-away origin: with the light camera 200 units out the textbook b^2 - ac loses ~4 digits.*/
static int ray_quadric(const vp_occluder* b, v3 o, v3 d, float* t0, float* t1)
{
    const int cyl = b->type == VP_OCC_CYLINDER;
    v3 w = v3_sub(o, v3_make(b->center[0], b->center[1], b->center[2]));
    float p[3], q[3];
    for (int k = 0; k < 3; ++k) {
        v3 a = v3_make(b->axes[3 * k], b->axes[3 * k + 1], b->axes[3 * k + 2]);
        p[k] = v3_dot(a, w) / b->half_extent[k];
        q[k] = v3_dot(a, d) / b->half_extent[k];
    }
    const float qq = cyl ? q[0] * q[0] + q[2] * q[2] : (q[0] * q[0] + q[1] * q[1]) + q[2] * q[2];
    *t0 = -3.0e38f; *t1 = 3.0e38f;
    if (qq > 0.f) {
        const float pq = cyl ? p[0] * q[0] + p[2] * q[2] : (p[0] * q[0] + p[1] * q[1]) + p[2] * q[2];
        const float tc = -(pq / qq);
        const float f0 = p[0] + tc * q[0], f1 = p[1] + tc * q[1], f2 = p[2] + tc * q[2];
        const float dist2 = cyl ? f0 * f0 + f2 * f2 : (f0 * f0 + f1 * f1) + f2 * f2;
        if (dist2 > 1.0f) return 0;
        const float half = sqrtf((1.0f - dist2) / qq);
        *t0 = tc - half; *t1 = tc + half;
    } else {
        const float dist2 = cyl ? p[0] * p[0] + p[2] * p[2] : (p[0] * p[0] + p[1] * p[1]) + p[2] * p[2];
        if (dist2 > 1.0f) return 0;
    }
    if (cyl) {
        if (q[1] != 0.f) {
            const float inv = 1.0f / q[1];
            const float ta = (-1.0f - p[1]) * inv, tb = (1.0f - p[1]) * inv;
            *t0 = fmaxf(*t0, fminf(ta, tb));
            *t1 = fminf(*t1, fmaxf(ta, tb));
        } else if (p[1] < -1.0f || p[1] > 1.0f) return 0;
    }
    return *t0 <= *t1;
}

static int ray_solid(const vp_occluder* b, v3 o, v3 d, float* t0, float* t1)
{
    return b->type == VP_OCC_BOX ? ray_obb((const vp_obb*)b, o, d, t0, t1) : ray_quadric(b, o, d, t0, t1);
}

VPO_API int vpo_set_occluders2(vpo_ctx* c, const vp_occluder* solids, int n)
{
    if (!c || n < 0 || (n > 0 && !solids)) return VP_ERR_BAD_ARG;
    for (int i = 0; i < n; ++i) {
        if (solids[i].type < VP_OCC_BOX || solids[i].type > VP_OCC_ELLIPSOID) return VP_ERR_BAD_ARG;
        if (solids[i].type != VP_OCC_BOX) for (int k = 0; k < 3; ++k) if (!(solids[i].half_extent[k] > 0.f)) return VP_ERR_BAD_ARG;
    }
    free(c->occluders); c->occluders = NULL; c->n_occluders = n;
    if (n > 0) { c->occluders = (vp_occluder*)malloc((size_t)n * sizeof(vp_occluder)); memcpy(c->occluders, solids, (size_t)n * sizeof(vp_occluder)); }
    return VP_OK;
}

VPO_API int vpo_set_occluders(vpo_ctx* c, const vp_obb* boxes, int n)
{
    if (!c || n < 0 || (n > 0 && !boxes)) return VP_ERR_BAD_ARG;
    free(c->occluders); c->occluders = NULL; c->n_occluders = n;
    if (n > 0) {
        c->occluders = (vp_occluder*)calloc((size_t)n, sizeof(vp_occluder));
        for (int i = 0; i < n; ++i) { memcpy(&c->occluders[i], &boxes[i], sizeof(vp_obb)); c->occluders[i].type = VP_OCC_BOX; }
    }
    return VP_OK;
}

VPO_API int vpo_render_light_depth(vpo_ctx* c, float nearz, float farz, float cam_dist, float* out)
{
    if (!c || !out || !c->have_frame) return VP_ERR_STATE;
    const int LW = c->Nx * c->nv, LH = c->Ny * c->nv;
    /* Ortho(-r, r, -t, t, 0.3, 1000), r = Nx*s/2, t = Ny*s/2                                 VPR.cs:338-342 */
    const float r = (float)c->Nx * c->s * 0.5f, t = (float)c->Ny * c->s * 0.5f;
    const float cx = c->gc[0] - c->fwd.x * cam_dist, cy = c->gc[1] - c->fwd.y * cam_dist, cz = c->gc[2] - c->fwd.z * cam_dist;   /* :365 */
    for (int Y = 0; Y < LH; ++Y)
        for (int X = 0; X < LW; ++X) {
            const float lx = -r + ((float)X + 0.5f) / (float)LW * (2.0f * r);
            const float ly = -t + ((float)Y + 0.5f) / (float)LH * (2.0f * t);
            v3 o = v3_make(cx + c->Rl[0] * lx + c->Rl[1] * ly, cy + c->Rl[3] * lx + c->Rl[4] * ly, cz + c->Rl[6] * lx + c->Rl[7] * ly);
            float zmin = 3.0e38f;
            for (int i = 0; i < c->n_occluders; ++i) {
                float t0, t1;
                if (!ray_solid(&c->occluders[i], o, c->fwd, &t0, &t1)) continue;
                if (t1 >= nearz && t1 <= farz) zmin = fminf(zmin, t1);           /* Cull Front: the back face is drawn */
            }
            out[(size_t)Y * LW + X] = zmin < 3.0e38f ? (zmin - nearz) / (farz - nearz) : 1.0f;
        }
    return VP_OK;
}

VPO_API int vpo_render_scene_depth(vpo_ctx* c, const vp_camera* cam, float* out)
{
    if (!c || !cam || !out) return VP_ERR_BAD_ARG;
    const int W = c->W, H = c->H;
    const float aspect = (float)W / (float)H;
    const float nit = -(1.0f / (float)tan((double)cam->fov_y * 0.5));
    const float* m = cam->camera_to_world;
    const float farc = cam->far_clip > 0.f ? cam->far_clip : 3.0e38f;
    v3 o = v3_make(M(m, 0, 3), M(m, 1, 3), M(m, 2, 3));
    for (int row = 0; row < H; ++row)
        for (int col = 0; col < W; ++col) {
            v3 d = v3_make((2.0f * ((float)col + 0.5f) / (float)W - 1.0f) * aspect, 2.0f * ((float)row + 0.5f) / (float)H - 1.0f, nit);
            v3 w = mul_dir(m, d);
            float best = 3.0e38f;
            for (int i = 0; i < c->n_occluders; ++i) {
                float t0, t1;
                if (!ray_solid(&c->occluders[i], o, w, &t0, &t1)) continue;
                float te = t0 > 0.f ? t0 : t1;
                float depth = te * (-nit);
                if (te > 0.f && depth >= cam->near_clip && depth <= farc) best = fminf(best, depth);
            }
            out[(size_t)row * W + col] = best;
        }
    return VP_OK;
}

/* ------------------------------------------------------------------------------------------------ */
/* F5-F12: FillMetavoxels / FillMetavoxel / frag                   VPR.cs:495-609, Fill.shader:152-274 */
/* ------------------------------------------------------------------------------------------------ */
static int retain_fill_params(vpo_ctx* c, const vp_fill_params* p)
{
    if (!p || !p->cubemap || p->cubemap_size < 1) { snprintf(c->err, sizeof c->err, "fill params: cubemap required"); return VP_ERR_BAD_ARG; }
    c->fp = *p;
    free(c->cubemap); free(c->depthmap); c->depthmap = NULL;
    size_t cn = (size_t)6 * p->cubemap_size * p->cubemap_size;
    c->cubemap = (float*)malloc(cn * sizeof(float));
    if (p->cubemap_format == VP_CUBEMAP_R8) {
        /* 8-bit UNORM texels (the reference's asset, DisplacementTexture.cubemap:10-23): D3D11 UNORM -> float = byte / 255 */
        const unsigned char* b = (const unsigned char*)p->cubemap;
        for (size_t i = 0; i < cn; ++i) c->cubemap[i] = (float)b[i] / 255.0f;
    } else {
        memcpy(c->cubemap, p->cubemap, cn * sizeof(float));
    }
    c->cube_s = p->cubemap_size;
    if (p->light_depth_map) {
        size_t dn = (size_t)c->Nx * c->nv * c->Ny * c->nv;
        c->depthmap = (float*)malloc(dn * sizeof(float));
        memcpy(c->depthmap, p->light_depth_map, dn * sizeof(float));
    } else if (c->n_occluders > 0) {
        size_t dn = (size_t)c->Nx * c->nv * c->Ny * c->nv;
        c->depthmap = (float*)malloc(dn * sizeof(float));
        vpo_render_light_depth(c, p->light_near, p->light_far, p->light_cam_distance, c->depthmap);
    }
    c->fp.cubemap = NULL; c->fp.light_depth_map = NULL;
    return VP_OK;
}

/* assign brick slots to the occupied MVs of the slab in draw (z-major) order */
static int assign_bricks(vpo_ctx* c)
{
    size_t n3 = (size_t)c->Nx * c->Ny * c->Nz;
    int occ = 0;
    for (size_t i = 0; i < n3; ++i) {
        int zz = (int)(i / ((size_t)c->Nx * c->Ny));
        int cnt = c->offsets[i + 1] - c->offsets[i];
        c->brick_index[i] = (cnt != 0 && zz >= c->z0 && zz < c->z1) ? occ++ : -1;   /* VPR.cs:511 */
    }
    c->occupied = occ;
    free(c->bricks);
    size_t nv3 = (size_t)c->nv * c->nv * c->nv;
    c->bricks = (uint16_t*)malloc(((size_t)occ * nv3 * 4 + 4) * sizeof(uint16_t));
    return c->bricks ? VP_OK : VP_ERR_OOM;
}

/* density/ao of one voxel column of one MV (the two coverage loops, Fill.shader:160-208). */
static void column_density(const vpo_ctx* c, const float* mvp, int px, int py, const int* ids, int n, float* dens, float* ao, v3* v0_out)
{
    const int nv = c->nv;
    const float fnv = (float)nv;
    /* get_voxel_world_pos(svPos, 0): normPos = ((svPos - nv/2)/nv, (z - nv/2)/nv, 1)   Fill.shader:96-107 */
    float nx = (((float)px + 0.5f) - fnv / 2.0f) / fnv, ny = (((float)py + 0.5f) - fnv / 2.0f) / fnv, nz = (0.0f - fnv / 2.0f) / fnv;
    /* _MetavoxelToWorld = TRS(mvPos, lightRot, sb)                                      VPR.cs:596 */
    v3 v0;
    v0.x = ((c->Rl[0] * c->sb * nx + c->Rl[1] * c->sb * ny) + c->Rl[2] * c->sb * nz) + mvp[0];
    v0.y = ((c->Rl[3] * c->sb * nx + c->Rl[4] * c->sb * ny) + c->Rl[5] * c->sb * nz) + mvp[1];
    v0.z = ((c->Rl[6] * c->sb * nx + c->Rl[7] * c->sb * ny) + c->Rl[8] * c->sb * nz) + mvp[2];
    *v0_out = v0;
    v3 dstep = v3_scale(c->fwd, c->one);                         /* _LightForward * oneVoxelSize  :183 */
    for (int s = 0; s < nv; ++s) { dens[s] = 0.f; ao[s] = 0.f; } /* "clear it" for particle 0     :178-181 */
    for (int i = 0; i < n; ++i) {
        const float* r = c->rec + 16 * (size_t)ids[i];
        if (c->literal) {
            /* literal reference stepping: voxelWorldPos += fwd*one per slice, full mat-vec per voxel */
            v3 v = v0;
            for (int s = 0; s < nv; ++s) {
                float psx = fmaf(r[2], v.z, fmaf(r[1], v.y, fmaf(r[0], v.x, r[3])));
                float psy = fmaf(r[6], v.z, fmaf(r[5], v.y, fmaf(r[4], v.x, r[7])));
                float psz = fmaf(r[10], v.z, fmaf(r[9], v.y, fmaf(r[8], v.x, r[11])));
                float d2 = fmaf(psz, psz, fmaf(psy, psy, psx * psx));
                if (d2 <= 0.25f) {                                                       /* :172 */
                    float den, a; voxel_color(c, psx, psy, psz, d2, r[12], &den, &a);
                    dens[s] += den; ao[s] = fmaxf(ao[s], a);                             /* :200-201 */
                }
                v = v3_add(v, dstep);
            }
        } else {
            /* arithmetic spec: ps(s) = A + s*B, A = W2P*(v0,1), B = W2P_linear*dstep */
            float Ax = fmaf(r[2], v0.z, fmaf(r[1], v0.y, fmaf(r[0], v0.x, r[3])));
            float Ay = fmaf(r[6], v0.z, fmaf(r[5], v0.y, fmaf(r[4], v0.x, r[7])));
            float Az = fmaf(r[10], v0.z, fmaf(r[9], v0.y, fmaf(r[8], v0.x, r[11])));
            float Bx = fmaf(r[2], dstep.z, fmaf(r[1], dstep.y, r[0] * dstep.x));
            float By = fmaf(r[6], dstep.z, fmaf(r[5], dstep.y, r[4] * dstep.x));
            float Bz = fmaf(r[10], dstep.z, fmaf(r[9], dstep.y, r[8] * dstep.x));
            for (int s = 0; s < nv; ++s) {
                float fs = (float)s;
                float psx = fmaf(fs, Bx, Ax), psy = fmaf(fs, By, Ay), psz = fmaf(fs, Bz, Az);
                float d2 = fmaf(psz, psz, fmaf(psy, psy, psx * psx));
                if (d2 <= 0.25f) {
                    float den, a; voxel_color(c, psx, psy, psz, d2, r[12], &den, &a);
                    dens[s] += den; ao[s] = fmaxf(ao[s], a);
                }
            }
        }
    }
}

/* shadow index of a column                                                        Fill.shader:211-221 */
static int column_shadow_index(const vpo_ctx* c, v3 v0, int X, int Y)
{
    /* _WorldToLight = lightCamera.worldToLocal; camera at gridCenter - fwd*200, light rotation  VPR.cs:365,535 */
    v3 camp = v3_make(c->gc[0] - c->fwd.x * c->fp.light_cam_distance, c->gc[1] - c->fwd.y * c->fp.light_cam_distance,
                      c->gc[2] - c->fwd.z * c->fp.light_cam_distance);
    v3 d = v3_sub(v0, camp);
    float z0 = (c->Rl[0 * 3 + 2] * d.x + c->Rl[1 * 3 + 2] * d.y) + c->Rl[2 * 3 + 2] * d.z;   /* lsVoxel0.z */
    float dm = c->depthmap ? c->depthmap[(size_t)Y * c->Nx * c->nv + X] : 1.0f;               /* tex2D at texel centre :217 */
    float a = 1.0f / (c->fp.light_far - c->fp.light_near), bq = -c->fp.light_near * a;
    float lsSceneDepth = (dm - bq) * (1.0f / a);                                              /* :218-219 */
    float q = (lsSceneDepth - z0) / c->one;                                                   /* :222 */
    if (!(q < 2.0e9f)) return 2000000000;
    if (q < -2.0e9f) return -2000000000;
    return (int)q;
}

/* propagate + store one column                                                    Fill.shader:224-269 */
static float column_propagate(const vpo_ctx* c, float T, int zz, const float* dens, const float* ao, int shadowIndex,
                              uint16_t* brick, int px, int py, float* T_out_unused)
{
    (void)T_out_unused;
    const int nv = c->nv;
    float transmitted = (zz == 0) ? c->fp.init_light_intensity : T;     /* :224 */
    float propagated = transmitted;
    const float diffuse = 0.4f;
    /* _MetavoxelBorderSize = Clamp(b, 0, nv-2)                           VPR.cs:528 */
    int bclamp = c->b < 0 ? 0 : (c->b > nv - 2 ? nv - 2 : c->b);
    int borderVoxelIndex = nv - bclamp;
    for (int s = 0; s < nv; ++s) {
        int inShadow = (s >= shadowIndex);
        if (inShadow) transmitted = 0.0f;
        else if (s < borderVoxelIndex) propagated = transmitted;        /* only the first loop updates it */
        float r = diffuse * transmitted + c->fp.ambient[0] * ao[s];
        float g = diffuse * transmitted + c->fp.ambient[1] * ao[s];
        float bl = diffuse * transmitted + c->fp.ambient[2] * ao[s];
        transmitted *= 1.0f / (1.0f + dens[s]);                          /* rcp(1 + density)  :244 */
        uint16_t* t = brick + (((size_t)s * nv + py) * nv + px) * 4;     /* volumeTex[int3(xy, slice)] */
        t[0] = f32_to_f16(r); t[1] = f32_to_f16(g); t[2] = f32_to_f16(bl); t[3] = f32_to_f16(dens[s]);
    }
    return propagated;                                                   /* lightPropogationTex[...] = propagatedLight :250 */
}

static int fill_impl(vpo_ctx* c, const float* light_in, int store_dens_only, float* tau_out)
{
    const int nv = c->nv;
    const size_t nv3 = (size_t)nv * nv * nv;
    const int LW = c->Nx * nv, LH = c->Ny * nv;
    /* GL.Clear(Color.red): light map := 1.0                              VPR.cs:498-499 */
    for (size_t i = 0; i < (size_t)LW * LH; ++i) c->lightmap[i] = light_in ? light_in[i] : 1.0f;
    if (store_dens_only) {
        free(c->dens_ao);
        c->dens_ao = (float*)malloc(((size_t)c->occupied * nv3 * 2 + 2) * sizeof(float));
        if (!c->dens_ao) return VP_ERR_OOM;
    }
    int ncols = c->Nx * c->Ny;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads(c))
    for (int col = 0; col < ncols; ++col) {
        int xx = col % c->Nx, yy = col / c->Nx;
        float dens[64], ao[64];
        for (int zz = c->z0; zz < c->z1; ++zz) {                         /* z order = draw order  VPR.cs:505 */
            size_t mi = mv_index(c, xx, yy, zz);
            int bi = c->brick_index[mi];
            if (bi < 0) continue;                                        /* empty MVs are skipped :511 */
            const int* ids = c->ids + c->offsets[mi];
            int n = c->offsets[mi + 1] - c->offsets[mi];
            uint16_t* brick = c->bricks + (size_t)bi * nv3 * 4;
            for (int py = 0; py < nv; ++py)
                for (int px = 0; px < nv; ++px) {
                    v3 v0;
                    int X = px + xx * nv, Y = py + yy * nv;
                    if (store_dens_only != 2) column_density(c, c->mvPos + 3 * mi, px, py, ids, n, dens, ao, &v0);
                    else {
                        /* second half of the split fill: reload density/ao, recompute v0 only */
                        const float* da = c->dens_ao + (size_t)bi * nv3 * 2;
                        for (int s = 0; s < nv; ++s) { dens[s] = da[(((size_t)s * nv + py) * nv + px) * 2]; ao[s] = da[(((size_t)s * nv + py) * nv + px) * 2 + 1]; }
                        float d0[64], a0[64]; (void)d0; (void)a0;
                        column_density(c, c->mvPos + 3 * mi, px, py, ids, 0, d0, a0, &v0);
                    }
                    if (store_dens_only == 1) {
                        float* da = c->dens_ao + (size_t)bi * nv3 * 2;
                        for (int s = 0; s < nv; ++s) { da[(((size_t)s * nv + py) * nv + px) * 2] = dens[s]; da[(((size_t)s * nv + py) * nv + px) * 2 + 1] = ao[s]; }
                    }
                    int si = column_shadow_index(c, v0, X, Y);
                    float* lm = c->lightmap + (size_t)Y * LW + X;
                    *lm = column_propagate(c, *lm, zz, dens, ao, si, brick, px, py, NULL);
                }
        }
    }
    if (tau_out) memcpy(tau_out, c->lightmap, (size_t)LW * LH * sizeof(float));
    return VP_OK;
}

VPO_API int vpo_fill(vpo_ctx* c, const vp_fill_params* p)
{
    if (!c) return VP_ERR_BAD_ARG;
    if (!c->have_frame || !c->ids) { snprintf(c->err, sizeof c->err, "vpo_fill before bin"); return VP_ERR_STATE; }
    if (c->nv > 64) return VP_ERR_UNSUPPORTED;
    int rc = retain_fill_params(c, p); if (rc) return rc;
    rc = assign_bricks(c); if (rc) return rc;
    rc = fill_impl(c, NULL, 0, NULL);
    c->filled = (rc == VP_OK);
    return rc;
}
/* split fill for slabs: see vp_fill_local / vp_fill_finish in vpfx.h.  tau_out/light_in are HOST arrays here. */
VPO_API int vpo_fill_local(vpo_ctx* c, const vp_fill_params* p, float* tau_out)
{
    if (!c || !tau_out) return VP_ERR_BAD_ARG;
    if (!c->have_frame || !c->ids) return VP_ERR_STATE;
    int rc = retain_fill_params(c, p); if (rc) return rc;
    rc = assign_bricks(c); if (rc) return rc;
    /* with T_in = 1 the slab's final light map IS its transmittance map; (zz==0) uses init intensity */
    return fill_impl(c, NULL, 1, tau_out);
}
VPO_API int vpo_fill_finish(vpo_ctx* c, const float* light_in)
{
    if (!c || !c->dens_ao) return VP_ERR_STATE;
    int rc = fill_impl(c, light_in, 2, NULL);
    c->filled = (rc == VP_OK);
    return rc;
}

VPO_API int vpo_read_brick(vpo_ctx* c, int xx, int yy, int zz, uint16_t* out)
{
    int bi = c->brick_index[mv_index(c, xx, yy, zz)];
    if (bi < 0 || !c->filled) return VP_ERR_STATE;
    size_t n = (size_t)c->nv * c->nv * c->nv * 4;
    memcpy(out, c->bricks + (size_t)bi * n, n * sizeof(uint16_t));
    return VP_OK;
}
VPO_API int vpo_read_lightmap(vpo_ctx* c, float* out)
{
    memcpy(out, c->lightmap, (size_t)c->Nx * c->nv * c->Ny * c->nv * sizeof(float));
    return VP_OK;
}

/* ------------------------------------------------------------------------------------------------ */
/* G1-G9: RenderMetavoxels / RenderMetavoxel / RM vert+frag / ROP blend   VPR.cs:613-794, RM:14-302   */
/* ------------------------------------------------------------------------------------------------ */
typedef struct { float key; int idx; } sort_item;

/* stable ascending merge sort on key (List.Sort is unstable in .NET; ties are DEFINED as stable here) */
static void stable_sort(sort_item* a, sort_item* tmp, int n)
{
    for (int w = 1; w < n; w *= 2) {
        for (int i = 0; i < n; i += 2 * w) {
            int m = i + w < n ? i + w : n, e = i + 2 * w < n ? i + 2 * w : n;
            int p = i, q = m, k = i;
            while (p < m && q < e) tmp[k++] = (a[q].key < a[p].key) ? a[q++] : a[p++];
            while (p < m) tmp[k++] = a[p++];
            while (q < e) tmp[k++] = a[q++];
        }
        memcpy(a, tmp, (size_t)n * sizeof *a);
    }
}

/* zBoundary                                                                        VPR.cs:642-648 */
static int z_boundary(const vpo_ctx* c, const vp_camera* cam)
{
    v3 lsCam = mul_point(c->Linv, v3_make(cam->cam_pos[0], cam->cam_pos[1], cam->cam_pos[2]));
    const float* p0 = c->mvPos;   /* mvGrid[0,0,0].mPos */
    float lsFirst = mul_point(c->Linv, v3_make(p0[0], p0[1], p0[2])).z;
    float over = (lsCam.z - lsFirst) / c->s;
    int zb = round_to_int(over);
    if (zb < -1) zb = -1;
    if (zb > c->Nz - 1) zb = c->Nz - 1;
    return zb;
}
VPO_API int vpo_z_boundary(vpo_ctx* c, const vp_camera* cam, int* zb) { *zb = z_boundary(c, cam); return VP_OK; }

/* draw order: returns malloc'd array of MV linear indices + kinds (0 OVER, 1 UNDER), restricted to the slab */
static int build_draw_order(const vpo_ctx* c, const vp_camera* cam, int** order_out, int** kind_out)
{
    int nxy = c->Nx * c->Ny;
    sort_item* it = (sort_item*)malloc((size_t)nxy * sizeof *it);
    sort_item* tmp = (sort_item*)malloc((size_t)nxy * sizeof *tmp);
    v3 cp = v3_make(cam->cam_pos[0], cam->cam_pos[1], cam->cam_pos[2]);
    for (int yy = 0; yy < c->Ny; ++yy)
        for (int xx = 0; xx < c->Nx; ++xx) {
            const float* p = c->mvPos + 3 * mv_index(c, xx, yy, 0);             /* mvGrid[0, yy, xx]  :622 */
            v3 d = v3_sub(v3_make(p[0], p[1], p[2]), cp);
            it[yy * c->Nx + xx].key = v3_dot(d, d);
            it[yy * c->Nx + xx].idx = yy * c->Nx + xx;
        }
    stable_sort(it, tmp, nxy);                                                  /* Sort() ascending   :628 */
    int zb = z_boundary(c, cam);
    int* order = (int*)malloc(((size_t)c->occupied + 1) * sizeof(int));
    int* kind = (int*)malloc(((size_t)c->occupied + 1) * sizeof(int));
    int n = 0;
    for (int zz = 0; zz <= zb; ++zz) {                                          /* phase A, far -> near :667-681 */
        if (zz < c->z0 || zz >= c->z1) continue;
        for (int i = nxy - 1; i >= 0; --i) {                                    /* Reverse()           :629 */
            size_t mi = (size_t)zz * nxy + it[i].idx;
            if (c->brick_index[mi] >= 0) { order[n] = (int)mi; kind[n++] = 0; }
        }
    }
    for (int zz = zb + 1; zz < c->Nz; ++zz) {                                   /* phase B, near -> far :695-711 */
        if (zz < c->z0 || zz >= c->z1) continue;
        for (int i = 0; i < nxy; ++i) {
            size_t mi = (size_t)zz * nxy + it[i].idx;
            if (c->brick_index[mi] >= 0) { order[n] = (int)mi; kind[n++] = 1; }
        }
    }
    free(it); free(tmp);
    *order_out = order; *kind_out = kind;
    return n;
}

/* tex3D(_VolumeTexture, tc): trilinear, wrap = repeat, no mips                       VPR.cs:769-770 */
static inline void sample_brick(const vpo_ctx* c, const uint16_t* brick, float tx, float ty, float tz, float out[4])
{
    const int nv = c->nv;
    float fx = tx * (float)nv - 0.5f, fy = ty * (float)nv - 0.5f, fz = tz * (float)nv - 0.5f;
    float x0 = floorf(fx), y0 = floorf(fy), z0 = floorf(fz);
    float wx = fx - x0, wy = fy - y0, wz = fz - z0;
    int ix0 = (int)x0, iy0 = (int)y0, iz0 = (int)z0;
    int ix1 = ix0 + 1, iy1 = iy0 + 1, iz1 = iz0 + 1;
    ix0 = ((ix0 % nv) + nv) % nv; ix1 = ((ix1 % nv) + nv) % nv;
    iy0 = ((iy0 % nv) + nv) % nv; iy1 = ((iy1 % nv) + nv) % nv;
    iz0 = ((iz0 % nv) + nv) % nv; iz1 = ((iz1 % nv) + nv) % nv;
    const float* lut = c->f16lut;
#define TX(z, y, x) (brick + ((((size_t)(z)) * nv + (y)) * nv + (x)) * 4)
    const uint16_t *t000 = TX(iz0, iy0, ix0), *t100 = TX(iz0, iy0, ix1), *t010 = TX(iz0, iy1, ix0), *t110 = TX(iz0, iy1, ix1);
    const uint16_t *t001 = TX(iz1, iy0, ix0), *t101 = TX(iz1, iy0, ix1), *t011 = TX(iz1, iy1, ix0), *t111 = TX(iz1, iy1, ix1);
#undef TX
    for (int ch = 0; ch < 4; ++ch) {
        float a00 = lut[t000[ch]], a10 = lut[t100[ch]], a01 = lut[t010[ch]], a11 = lut[t110[ch]];
        float b00 = lut[t001[ch]], b10 = lut[t101[ch]], b01 = lut[t011[ch]], b11 = lut[t111[ch]];
        float a0 = fmaf(wx, a10 - a00, a00), a1 = fmaf(wx, a11 - a01, a01);
        float b0 = fmaf(wx, b10 - b00, b00), b1 = fmaf(wx, b11 - b01, b01);
        float a = fmaf(wy, a1 - a0, a0), bq = fmaf(wy, b1 - b0, b0);
        out[ch] = fmaf(wz, bq - a, a);
    }
}

typedef struct {
    float W, H, aspect, negInvTan;    /* csRayDir.z = -rcp(tan(fov/2)) */
    v3 csVolOrigin;
    float maxDim, halfZ, zMin, rayLen, mvStep;
    float nearc, farc;
    int steps, soft, flags;
    float bo;
} rm_consts;

static void ray_setup(const rm_consts* k, int col, int row, v3* dir, v3* start)
{
    /* csRayDir.xy = 2*pos/res - 1; x *= aspect; z = -rcp(tan(fov/2)); normalize        RM.shader:188-194 */
    float px = (float)col + 0.5f, py = (float)row + 0.5f;
    v3 d;
    d.x = (2.0f * px / k->W) - 1.0f;
    d.y = (2.0f * py / k->H) - 1.0f;
    d.x *= k->aspect;
    d.z = k->negInvTan;
    float inv = 1.0f / sqrtf(v3_dot(d, d));
    d = v3_scale(d, inv);
    *dir = d;
    *start = v3_scale(d, k->zMin / d.z);                                               /* csAABBStart :212 */
}

/* fragment of one MV for one pixel.  Returns 1 if the rasteriser produces a fragment AND it blends
 * (src written), 0 otherwise.                                                          RM.shader:166-302 */
static int march_mv(const vpo_ctx* c, const rm_consts* k, const float* c2m, const uint16_t* brick, v3 dir, v3 start,
                    float sceneDepth, float src[4], int* nsamp)
{
    v3 o = mul_point(c2m, start);                                                      /* mvRay.o :216 */
    v3 d = mul_dir(c2m, dir);
    float inv = 1.0f / sqrtf(v3_dot(d, d));
    d = v3_scale(d, inv);                                                              /* mvRay.d :217 */
    /* IntersectBox                                                                     RM.shader:95-118 */
    float irx = 1.0f / d.x, iry = 1.0f / d.y, irz = 1.0f / d.z;
    float tbx = irx * (-0.5f - o.x), tby = iry * (-0.5f - o.y), tbz = irz * (-0.5f - o.z);
    float ttx = irx * (0.5f - o.x), tty = iry * (0.5f - o.y), ttz = irz * (0.5f - o.z);
    float tminx = fminf(ttx, tbx), tminy = fminf(tty, tby), tminz = fminf(ttz, tbz);
    float tmaxx = fmaxf(ttx, tbx), tmaxy = fmaxf(tty, tby), tmaxz = fmaxf(ttz, tbz);
    float t1 = fmaxf(fmaxf(tminx, tminy), fmaxf(tminx, tminz));
    float t2 = fminf(fminf(tmaxx, tmaxy), fminf(tmaxx, tmaxz));
    *nsamp = 0;
    if (t1 > t2) return 0;      /* no back-face fragment (and the shader's own test returns seethrough, :229-230) */
    /* rasteriser coverage (Cull Front, near/far clip, ZTest Less): the pixel-centre ray leaves the cube at t2;
     * that back-face point must lie inside the clip volume and in front of the opaque scene.   RM.shader:14 */
    float exitDepth = -(start.z + dir.z * (t2 * c->s));
    if (!(exitDepth > k->nearc) || !(exitDepth <= k->farc)) return 0;
    if (!(exitDepth < sceneDepth)) return 0;
    int tEntry = (int)ceilf(t1 / k->mvStep);                                           /* :236 */
    int tExit = (int)floorf(t2 / k->mvStep);                                           /* :237 */
    v3 camM = v3_make(M(c2m, 0, 3), M(c2m, 1, 3), M(c2m, 2, 3));                       /* mul(C2M,(0,0,0,1)) :238 */
    v3 co = v3_sub(camM, o);
    int tCamera = (int)(sqrtf(v3_dot(co, co)) / k->mvStep);                            /* :239 */
    if (tCamera > tEntry) tEntry = tCamera;                                            /* :240 */
    float res[3] = {0.f, 0.f, 0.f};
    float trans = 1.0f;
    v3 stepv = v3_scale(d, k->mvStep);                                                 /* mvRayStep :248 */
    v3 p = v3_add(o, v3_scale(stepv, (float)tExit));                                   /* :249 */
    float scale = 1.0f - 2.0f * k->bo;
    float invSoft = 1.0f / (float)k->soft;
    for (int si = tExit; si >= tEntry; --si) {                                         /* :254 */
        float sx = (p.x + 0.5f) * scale + k->bo, sy = (p.y + 0.5f) * scale + k->bo, sz = (p.z + 0.5f) * scale + k->bo;  /* :255-258 */
        float vox[4];
        sample_brick(c, brick, sx, sy, sz, vox);                                       /* :262 */
        float density = vox[3];
        if (si - tCamera < k->soft) density *= (float)(si - tCamera) * invSoft;        /* :267-270 */
        float bf = 1.0f / (1.0f + density);                                            /* :272 */
        for (int ch = 0; ch < 3; ++ch) res[ch] = vox[ch] + bf * (res[ch] - vox[ch]);   /* lerp(color, result, bf) :274 */
        trans *= bf;                                                                   /* :275 */
        p = v3_sub(p, stepv);                                                          /* :277 */
        (*nsamp)++;
    }
    src[0] = res[0]; src[1] = res[1]; src[2] = res[2]; src[3] = 1.0f - trans;          /* :301 */
    if (k->flags & VP_RM_SHOW_NUM_SAMPLES) {                                           /* debug view :283-299 */
        static const float tab[7][4] = {{0.f, 0.2f, 0.f, 0.5f}, {0.f, 0.5f, 0.f, 0.5f}, {0.5f, 0.5f, 0.f, 0.5f}, {0.6f, 0.4f, 0.f, 0.5f},
                                        {0.6f, 0.f, 0.f, 0.5f}, {0.8f, 0.f, 0.f, 0.5f}, {1.0f, 0.f, 0.f, 0.5f}};
        int n = *nsamp, i = n < 5 ? 0 : n < 10 ? 1 : n < 20 ? 2 : n < 30 ? 3 : n < 40 ? 4 : n < 50 ? 5 : 6;
        memcpy(src, tab[i], sizeof tab[i]);
    }
    return 1;
}

static void make_rm_consts(const vpo_ctx* c, const vp_camera* cam, const vp_raymarch_params* rp, rm_consts* k)
{
    k->W = (float)c->W; k->H = (float)c->H;
    k->aspect = k->W / k->H;                                         /* _ScreenRes.x / _ScreenRes.y  :190 */
    k->negInvTan = -(1.0f / (float)tan((double)cam->fov_y * 0.5));   /* uniform; evaluated once on the host */
    k->csVolOrigin = mul_point(cam->world_to_camera, v3_make(c->gc[0], c->gc[1], c->gc[2]));   /* :203 */
    int md = c->Nx > c->Ny ? c->Nx : c->Ny; if (c->Nz > md) md = c->Nz;
    k->maxDim = (float)md;
    k->halfZ = 1.73205f * 0.5f * k->maxDim * c->s;                   /* SQ_ROOT_3*0.5*maxGridDim*size :207 */
    k->zMin = k->csVolOrigin.z + k->halfZ;                           /* :208 */
    k->rayLen = 2.0f * k->halfZ;                                     /* :213 */
    float total = k->maxDim * (float)rp->steps_per_mv;               /* :220 */
    float invTotal = 1.0f / total;
    float mvRayLength = k->rayLen * (1.0f / c->s);                   /* :222 */
    k->mvStep = mvRayLength * invTotal;                              /* :223 */
    k->nearc = cam->near_clip; k->farc = cam->far_clip > 0.f ? cam->far_clip : 3.0e38f;
    k->steps = rp->steps_per_mv; k->soft = rp->soft_distance; k->flags = rp->flags;
    k->bo = (1.0f / (float)c->nv) * (float)c->b;                     /* rcp(_NumVoxels)*_MetavoxelBorderSize :245 */
}

/* conservative screen rectangle of an MV's (unbordered) cube */
static void mv_screen_rect(const vpo_ctx* c, const rm_consts* k, const vp_camera* cam, const float* mvp, int rect[4])
{
    float f = -k->negInvTan;
    float minx = 1e30f, miny = 1e30f, maxx = -1e30f, maxy = -1e30f;
    int full = 0;
    for (int i = 0; i < 8 && !full; ++i) {
        float lx = (i & 1) ? 0.5f : -0.5f, ly = (i & 2) ? 0.5f : -0.5f, lz = (i & 4) ? 0.5f : -0.5f;
        v3 w;
        w.x = mvp[0] + c->s * (c->Rl[0] * lx + c->Rl[1] * ly + c->Rl[2] * lz);
        w.y = mvp[1] + c->s * (c->Rl[3] * lx + c->Rl[4] * ly + c->Rl[5] * lz);
        w.z = mvp[2] + c->s * (c->Rl[6] * lx + c->Rl[7] * ly + c->Rl[8] * lz);
        v3 pc = mul_point(cam->world_to_camera, w);
        if (pc.z > -1e-3f) { full = 1; break; }
        float ndx = (pc.x / -pc.z) * f / k->aspect, ndy = (pc.y / -pc.z) * f;
        float sx = (ndx + 1.0f) * 0.5f * k->W - 0.5f, sy = (ndy + 1.0f) * 0.5f * k->H - 0.5f;
        if (sx < minx) minx = sx; if (sx > maxx) maxx = sx;
        if (sy < miny) miny = sy; if (sy > maxy) maxy = sy;
    }
    if (full) { rect[0] = 0; rect[1] = 0; rect[2] = c->W - 1; rect[3] = c->H - 1; return; }
    float x0 = floorf(minx) - 2.f, y0 = floorf(miny) - 2.f, x1 = ceilf(maxx) + 2.f, y1 = ceilf(maxy) + 2.f;
    if (x0 < 0.f) x0 = 0.f; if (y0 < 0.f) y0 = 0.f;
    if (x1 > (float)(c->W - 1)) x1 = (float)(c->W - 1); if (y1 > (float)(c->H - 1)) y1 = (float)(c->H - 1);
    rect[0] = (int)x0; rect[1] = (int)y0; rect[2] = (int)x1; rect[3] = (int)y1;   /* may be empty (x1 < x0) */
}

/* shared implementation.  If img_under == NULL: one image, OVER and UNDER applied to the same dst in draw order
 * (the reference).  Otherwise OVER-phase MVs composite into img_over and UNDER-phase MVs into img_under. */
static int raymarch_impl(vpo_ctx* c, const vp_camera* cam, const vp_raymarch_params* rp, float* img_over, float* img_under, int* mask)
{
    if (!c || !cam || !rp || !img_over) return VP_ERR_BAD_ARG;
    if (!c->filled) { snprintf(c->err, sizeof c->err, "raymarch before fill"); return VP_ERR_STATE; }
    if (rp->steps_per_mv < 1 || c->W < 1 || c->H < 1) return VP_ERR_BAD_ARG;
    rm_consts k; make_rm_consts(c, cam, rp, &k);
    float* occ_depth = NULL;
    const float* scene_depth = rp->scene_depth;
    if (!scene_depth && c->n_occluders > 0) {
        occ_depth = (float*)malloc((size_t)c->W * c->H * sizeof(float));
        vpo_render_scene_depth(c, cam, occ_depth);
        scene_depth = occ_depth;
    }
    int *order, *kind;
    int n = build_draw_order(c, cam, &order, &kind);
    const size_t nv3 = (size_t)c->nv * c->nv * c->nv;
    /* per draw: _CameraToMetavoxel = TRS(mvPos, lightRot, s).inverse * cameraToWorld   VPR.cs:774-778 */
    float* c2m = (float*)malloc(((size_t)n + 1) * 16 * sizeof(float));
    int* rects = (int*)malloc(((size_t)n + 1) * 4 * sizeof(int));
    int m = 0;
    for (int i = 0; i < n; ++i) {
        const float* mvp = c->mvPos + 3 * (size_t)order[i];
        float inv[16];
        trs_inverse(v3_make(mvp[0], mvp[1], mvp[2]), c->Rl, c->s, inv);
        mat_mul(inv, cam->camera_to_world, c2m + 16 * (size_t)i);
        mv_screen_rect(c, &k, cam, mvp, rects + 4 * (size_t)i);
        m |= 1 << kind[i];
    }
    if (mask) *mask = m;
    /* OnPreRender: particlesRT cleared to (0,0,0,0)                                    VPR.cs:171-172 */
    memset(img_over, 0, (size_t)c->W * c->H * 4 * sizeof(float));
    if (img_under) memset(img_under, 0, (size_t)c->W * c->H * 4 * sizeof(float));
    long total_samples = 0;
    const int band = 4;
    int nbands = (c->H + band - 1) / band;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : total_samples) num_threads(nthreads(c))
    for (int bi = 0; bi < nbands; ++bi) {
        int r0 = bi * band, r1 = r0 + band - 1; if (r1 > c->H - 1) r1 = c->H - 1;
        for (int i = 0; i < n; ++i) {                                  /* draws in submission order */
            const int* rc = rects + 4 * (size_t)i;
            int ya = rc[1] > r0 ? rc[1] : r0, yb = rc[3] < r1 ? rc[3] : r1;
            if (ya > yb || rc[0] > rc[2]) continue;
            const uint16_t* brick = c->bricks + (size_t)c->brick_index[order[i]] * nv3 * 4;
            const float* cm = c2m + 16 * (size_t)i;
            float* img = (img_under && kind[i] == 1) ? img_under : img_over;
            for (int row = ya; row <= yb; ++row)
                for (int col = rc[0]; col <= rc[2]; ++col) {
                    v3 dir, start; ray_setup(&k, col, row, &dir, &start);
                    float sd = scene_depth ? scene_depth[(size_t)row * c->W + col] : 3.0e38f;
                    float src[4]; int ns;
                    if (!march_mv(c, &k, cm, brick, dir, start, sd, src, &ns)) continue;
                    total_samples += ns;
                    if (k.flags & VP_RM_SHOW_BLEND_FUNC) {                     /* debug view          RM.shader:174-181 */
                        if (kind[i] == 0) { src[0] = 0.5f; src[1] = 0.5f; src[2] = 0.f; src[3] = 1.f; }
                        else { src[0] = 0.f; src[1] = 0.5f; src[2] = 0.5f; src[3] = 1.f; }
                    }
                    if (k.flags & VP_RM_SHOW_DRAW_ORDER) {                     /* DrawOrderColoring   RM.shader:123-138,170-173 */
                        int per = (int)ceilf((float)c->occupied / 3.0f);       /* _NumMetavoxelsCovered VPR.cs:755; _OrderIndex = i */
                        if (per < 1) per = 1;
                        int sel = i / per, idx = i % per;
                        float v = (float)(per - idx) / (float)per;
                        src[0] = sel >= 2 ? v : 0.f; src[1] = sel == 0 ? v : 0.f; src[2] = sel == 1 ? v : 0.f; src[3] = 1.f;
                    }
                    float* dst = img + ((size_t)row * c->W + col) * 4;
                    if (kind[i] == 0) {        /* Blend One OneMinusSrcAlpha (all channels)      VPR.cs:659-662 */
                        float ia = 1.0f - src[3];
                        for (int ch = 0; ch < 4; ++ch) dst[ch] = src[ch] + dst[ch] * ia;
                    } else {                   /* Blend OneMinusDstAlpha One                     VPR.cs:688-691 */
                        float ia = 1.0f - dst[3];
                        for (int ch = 0; ch < 4; ++ch) dst[ch] = src[ch] * ia + dst[ch];
                    }
                    if (k.flags & VP_RM_QUANTIZE_UNORM8)                       /* particlesRT is ARGB32 (Q19)   VPR.cs:228 */
                        for (int ch = 0; ch < 4; ++ch) dst[ch] = floorf(fminf(fmaxf(dst[ch], 0.f), 1.f) * 255.0f + 0.5f) / 255.0f;
                }
        }
    }
    c->samples = total_samples;
    free(order); free(kind); free(c2m); free(rects); free(occ_depth);
    return VP_OK;
}

VPO_API int vpo_raymarch(vpo_ctx* c, const vp_camera* cam, const vp_raymarch_params* rp, float* rgba)
{
    return raymarch_impl(c, cam, rp, rgba, NULL, NULL);
}
VPO_API int vpo_raymarch_partial(vpo_ctx* c, const vp_camera* cam, const vp_raymarch_params* rp, float* over, float* under, int* mask)
{
    if (!under) return VP_ERR_BAD_ARG;
    return raymarch_impl(c, cam, rp, over, under, mask);
}

/* ordered final blend of partial images (slab-granular restatement of VPR.cs:652-711) */
VPO_API int vpo_blend_partials(int W, int H, const float* const* partials, const int* kinds, int n, float* out)
{
    size_t np = (size_t)W * H;
    memset(out, 0, np * 4 * sizeof(float));
    for (int i = 0; i < n; ++i) {
        const float* s = partials[i];
        for (size_t p = 0; p < np; ++p) {
            float* d = out + 4 * p; const float* q = s + 4 * p;
            if (kinds[i] == 0) { float ia = 1.0f - q[3]; for (int ch = 0; ch < 4; ++ch) d[ch] = q[ch] + d[ch] * ia; }
            else { float ia = 1.0f - d[3]; for (int ch = 0; ch < 4; ++ch) d[ch] = q[ch] * ia + d[ch]; }
        }
    }
    return VP_OK;
}

/* CompositeParticles.shader: Blend One OneMinusSrcAlpha, One One                     Comp.shader:10 */
VPO_API int vpo_composite(int W, int H, const float* particles, float* scene)
{
    size_t np = (size_t)W * H;
    for (size_t p = 0; p < np; ++p) {
        const float* s = particles + 4 * p; float* d = scene + 4 * p;
        float ia = 1.0f - s[3];
        d[0] = s[0] + d[0] * ia; d[1] = s[1] + d[1] * ia; d[2] = s[2] + d[2] * ia;
        d[3] = s[3] + d[3];
    }
    return VP_OK;
}

VPO_API int vpo_get_stats(vpo_ctx* c, vp_stats* st)
{
    memset(st, 0, sizeof *st);
    st->particles = c->P;
    size_t n3 = (size_t)c->Nx * c->Ny * c->Nz;
    long occ = 0, pairs = 0, mx = 0;
    for (size_t i = 0; i < n3; ++i) {
        int zz = (int)(i / ((size_t)c->Nx * c->Ny));
        if (zz < c->z0 || zz >= c->z1) continue;
        int cnt = c->offsets[i + 1] - c->offsets[i];
        if (cnt) { occ++; pairs += cnt; if (cnt > mx) mx = cnt; }
    }
    st->occupied_mv = occ; st->pairs = pairs; st->max_pairs_per_mv = mx;
    st->voxels_filled = occ * (long)c->nv * c->nv * c->nv;
    st->samples = c->samples;
    st->brick_bytes = occ * (long)c->nv * c->nv * c->nv * 8;
    return VP_OK;
}

/* exposed for unit tests of the arithmetic spec */
VPO_API void vpo_sincos_deg(float deg, float* s, float* c) { sincos_deg(deg, s, c); }
VPO_API uint16_t vpo_f32_to_f16(float f) { return f32_to_f16(f); }
VPO_API float vpo_f16_to_f32(uint16_t h) { return f16_to_f32(h); }
VPO_API float vpo_sample_cubemap(const float* cube, int S, float dx, float dy, float dz) { return sample_cubemap(cube, S, dx, dy, dz); }
