// host_logic.cpp -- the C#-side ("L2") logic of the reference's hot path, restated as host C++:
// grid layout, per-pass uniforms, metavoxel ordering.  Counterpart of
//   VolumetricParticleRenderer.UpdateMetavoxelPositions      VPR.cs:370-394
//   SetFillPassConstants / SetRaymarchPassConstants          VPR.cs:523-554, 716-763
//   SortMetavoxelSlicesFarToNearFromEye + zBoundary          VPR.cs:613-648
// fp32 with the operation order of the arithmetic spec; no contraction.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <vector>

#include "vpfx_internal.h"

namespace {

struct V3 { float x, y, z; };

inline float cm(const float* m, int r, int c) { return m[c * 4 + r]; }   // Unity Matrix4x4 is column-major

// Matrix4x4.MultiplyPoint3x4 on a row-packed 3x4
inline V3 mul_point_rows(const float* rows, V3 v)
{
    V3 r;
    r.x = ((rows[0] * v.x + rows[1] * v.y) + rows[2] * v.z) + rows[3];
    r.y = ((rows[4] * v.x + rows[5] * v.y) + rows[6] * v.z) + rows[7];
    r.z = ((rows[8] * v.x + rows[9] * v.y) + rows[10] * v.z) + rows[11];
    return r;
}

}  // namespace

void hl_build_grid(vp_ctx* c)
{
    GridConsts& g = c->g;
    const vp_config& cfg = c->cfg;
    g.Nx = cfg.num_mv[0]; g.Ny = cfg.num_mv[1]; g.Nz = cfg.num_mv[2];
    g.nv = cfg.num_voxels; g.b = cfg.num_border;
    g.z0 = cfg.slab_z0; g.z1 = cfg.slab_z1;
    if (g.z0 == 0 && g.z1 == 0) g.z1 = g.Nz;
    g.s = cfg.mv_scale;
    g.sb = g.s * (float)g.nv / (float)(g.nv - 2 * g.b);        // mvScaleWithBorder              VPR.cs:139
    g.one = g.sb / (float)g.nv;                                 // oneVoxelSize                   Fill.shader:160
    g.inv_sb = 1.0f / g.sb;

    const float* L = c->L;
    for (int r = 0; r < 3; ++r)
        for (int k = 0; k < 3; ++k) g.Rl[r * 3 + k] = cm(L, r, k);
    // worldToLocal of the rigid light transform: rows = Rl^T, translation = -(row . t)
    const V3 t{cm(L, 0, 3), cm(L, 1, 3), cm(L, 2, 3)};
    for (int k = 0; k < 3; ++k) {
        float rx = g.Rl[0 * 3 + k] * 1.0f, ry = g.Rl[1 * 3 + k] * 1.0f, rz = g.Rl[2 * 3 + k] * 1.0f;
        g.Linv[k * 4 + 0] = rx; g.Linv[k * 4 + 1] = ry; g.Linv[k * 4 + 2] = rz;
        g.Linv[k * 4 + 3] = -((rx * t.x + ry * t.y) + rz * t.z);
    }
    for (int i = 0; i < 3; ++i) g.gc[i] = c->gc[i];
    const V3 lsO = mul_point_rows(g.Linv, V3{c->gc[0], c->gc[1], c->gc[2]});     // lsWorldOrigin    VPR.cs:380
    g.lsO[0] = lsO.x; g.lsO[1] = lsO.y; g.lsO[2] = lsO.z;
    for (int i = 0; i < 9; ++i) g.Rsb[i] = g.Rl[i] * g.sb;
    for (int k = 0; k < 3; ++k)
        for (int j = 0; j < 3; ++j) g.rowsb[k * 3 + j] = g.Rl[j * 3 + k] * g.inv_sb;
    // forward.normalized
    const V3 f{g.Rl[2], g.Rl[5], g.Rl[8]};
    const float mag = std::sqrt((f.x * f.x + f.y * f.y) + f.z * f.z);
    if (mag > 1e-5f) { g.fwd[0] = f.x / mag; g.fwd[1] = f.y / mag; g.fwd[2] = f.z / mag; }
    else g.fwd[0] = g.fwd[1] = g.fwd[2] = 0.f;

    // light localToWorld as rows
    float Lrows[12];
    for (int r = 0; r < 3; ++r)
        for (int k = 0; k < 4; ++k) Lrows[r * 4 + k] = cm(L, r, k);
    for (int zz = 0; zz < g.Nz; ++zz)
        for (int yy = 0; yy < g.Ny; ++yy)
            for (int xx = 0; xx < g.Nx; ++xx) {
                // integer N/2 (VPR.cs:388): zz = 0 is the slice nearest the light
                const V3 off{(float)(g.Nx / 2 - xx) * g.s, (float)(g.Ny / 2 - yy) * g.s, (float)(g.Nz / 2 - zz) * g.s};
                const V3 p = mul_point_rows(Lrows, V3{lsO.x - off.x, lsO.y - off.y, lsO.z - off.z});
                float* d = c->h_mvPos + 3 * (((size_t)zz * g.Ny + yy) * g.Nx + xx);
                d[0] = p.x; d[1] = p.y; d[2] = p.z;
            }
}

void hl_build_psys(vp_ctx* c, const float m[16])
{
    PsysConsts& p = c->psys;
    for (int r = 0; r < 3; ++r)
        for (int k = 0; k < 4; ++k) p.L2W[r * 4 + k] = cm(m, r, k);
    // particleSys.transform.forward; Quaternion.AngleAxis normalises its axis      VPR.cs:583
    const V3 a{cm(m, 0, 2), cm(m, 1, 2), cm(m, 2, 2)};
    const float am = std::sqrt((a.x * a.x + a.y * a.y) + a.z * a.z);
    if (am > 0.f) { p.axis[0] = a.x / am; p.axis[1] = a.y / am; p.axis[2] = a.z / am; }
    else { p.axis[0] = 0.f; p.axis[1] = 0.f; p.axis[2] = 1.f; }
    p.rot_in_radians = c->lay.rotation_in_radians;
}

void hl_build_fill_consts(vp_ctx* c, const vp_fill_params* p)
{
    FillConsts& f = c->fc;
    const GridConsts& g = c->g;
    f.opacity_factor = p->opacity_factor;
    f.D = p->displacement_scale;
    f.one_minus_D = 1.0f - p->displacement_scale;
    f.D_over_255 = p->displacement_scale / 255.0f;
    f.lds_pitch = cube_u8_pitch(c->cubeS);
    f.d_is_one = p->displacement_scale == 1.0f ? 1 : 0;
    // grey ambient (the reference's default, scene:9021): r = g = b in every voxel -> (luminance, density) bricks; needs border >= 1
    // (the border-less brick filters with wrap-around, which only the RGBA sampling path implements)
    f.grey = (p->ambient[0] == p->ambient[1] && p->ambient[1] == p->ambient[2] && g.b >= 1 && c->cfg.reserved[1] != VPFX_CFG_NO_GREY_BRICKS) ? 1 : 0;
    f.init_light = p->init_light_intensity;
    for (int i = 0; i < 3; ++i) f.amb[i] = p->ambient[i];
    f.fade = p->fade_out_particles;
    const float a = 1.0f / (p->light_far - p->light_near);       // Fill.shader:218
    f.bq = -p->light_near * a;
    f.inv_a = 1.0f / a;
    for (int i = 0; i < 3; ++i) {
        f.camp[i] = g.gc[i] - g.fwd[i] * p->light_cam_distance;  // UpdatePositionOfCameraAtLight  VPR.cs:365
        f.dstep[i] = g.fwd[i] * g.one;
    }
    f.cubeS = c->cubeS;
    f.half_s = 0.5f * (float)c->cubeS;
    f.half_s_m05 = f.half_s - 0.5f;
    const int bclamp = std::min(std::max(g.b, 0), g.nv - 2);     // Mathf.Clamp(b, 0, nv-2)        VPR.cs:528
    f.border_index = g.nv - bclamp;
}

int hl_z_boundary(const vp_ctx* c, const vp_camera* cam)
{
    const GridConsts& g = c->g;
    const V3 lsCam = mul_point_rows(g.Linv, V3{cam->cam_pos[0], cam->cam_pos[1], cam->cam_pos[2]});
    const float* p0 = c->h_mvPos;   // mvGrid[0,0,0].mPos
    const float lsFirst = mul_point_rows(g.Linv, V3{p0[0], p0[1], p0[2]}).z;
    const float over = (lsCam.z - lsFirst) / g.s;                // mvBlendOverIndex               VPR.cs:644
    int zb = (int)std::rint(over);                               // Mathf.RoundToInt: half-to-even
    return std::min(std::max(zb, -1), g.Nz - 1);
}

void hl_build_rm_consts(const vp_ctx* c, const vp_camera* cam, const vp_raymarch_params* rp, RmConsts* k)
{
    const GridConsts& g = c->g;
    std::memset(k, 0, sizeof *k);
    k->W = c->cfg.width; k->H = c->cfg.height;
    k->Nx = g.Nx; k->Ny = g.Ny; k->Nz = g.Nz; k->nv = g.nv; k->z0 = g.z0; k->z1 = g.z1;
    k->zB = hl_z_boundary(c, cam);
    k->steps = rp->steps_per_mv; k->soft = rp->soft_distance;
    k->flags = rp->flags;
    for (int i = 0; i < 3; ++i) k->cam_world[i] = cam->cam_pos[i];
    k->num_covered = c->h_meta.occupied;                                     // numMetavoxelsCovered  VPR.cs:515, 755
    k->aspect = (float)k->W / (float)k->H;                                   // RM.shader:190
    k->neg_inv_tan = -(1.0f / (float)std::tan((double)cam->fov_y * 0.5));    // RM.shader:193
    const float* w2c = cam->world_to_camera;
    const float csOz = ((cm(w2c, 2, 0) * g.gc[0] + cm(w2c, 2, 1) * g.gc[1]) + cm(w2c, 2, 2) * g.gc[2]) + cm(w2c, 2, 3);
    const float maxDim = (float)std::max(g.Nx, std::max(g.Ny, g.Nz));
    const float halfZ = 1.73205f * 0.5f * maxDim * g.s;                      // RM.shader:207
    k->zMin = csOz + halfZ;                                                  // RM.shader:208
    k->s = g.s;
    const float total = maxDim * (float)rp->steps_per_mv;                    // RM.shader:220
    k->mvStep = ((2.0f * halfZ) * (1.0f / g.s)) * (1.0f / total);            // RM.shader:221-223
    k->inv_mvStep = 1.0f / k->mvStep;
    k->nearc = cam->near_clip;
    k->farc = cam->far_clip > 0.f ? cam->far_clip : 3.0e38f;
    // _CameraToMetavoxel = TRS(mvPos, lightRot, s).inverse * cameraToWorld (VPR.cs:774-778).  Its linear part is
    // the same for every MV; the per-MV translation column is finished on the device (k_rm_prepare).
    {
        const float inv = 1.0f / g.s;
        for (int r = 0; r < 3; ++r)
            for (int j = 0; j < 3; ++j) k->inv_rows[r * 3 + j] = g.Rl[j * 3 + r] * inv;
        const float* B = cam->camera_to_world;
        for (int r = 0; r < 3; ++r)
            for (int col = 0; col < 3; ++col)
                k->c2m_lin[r * 3 + col] = (k->inv_rows[r * 3 + 0] * cm(B, 0, col) + k->inv_rows[r * 3 + 1] * cm(B, 1, col)) +
                                          k->inv_rows[r * 3 + 2] * cm(B, 2, col);
        for (int j = 0; j < 4; ++j) k->c2w_t[j] = cm(B, j, 3);
    }
    // camera space -> grid space: g = Rl^T (C2W p - mvPos[0,0,0]) / s + 0.5, evaluated in double (traversal only).
    const float* c2w = cam->camera_to_world;
    const float* p0 = c->h_mvPos;
    for (int r = 0; r < 3; ++r) {
        for (int col = 0; col < 3; ++col) {
            double acc = 0.0;
            for (int j = 0; j < 3; ++j) acc += (double)g.Rl[j * 3 + r] * (double)cm(c2w, j, col);
            k->c2g[r * 4 + col] = (float)(acc / (double)g.s);
        }
        double acc = 0.0;
        for (int j = 0; j < 3; ++j) acc += (double)g.Rl[j * 3 + r] * ((double)cm(c2w, j, 3) - (double)p0[j]);
        k->c2g[r * 4 + 3] = (float)(acc / (double)g.s + 0.5);
        k->camg[r] = k->c2g[r * 4 + 3];
    }
    // Brick rows run along grid x.  A lane quad (four consecutive lanes) costs one L1 cycle when its four footprint loads share a 128-byte
    // line, so consecutive lanes should step along the SCREEN axis on which grid x changes fastest: screen x normally, screen y when the
    // view is rolled (d grid-x / d camera-y dominates).  Scheduling only; the image does not depend on it.
    k->lane_transpose = std::fabs(k->c2g[1]) > std::fabs(k->c2g[0]) ? 1 : 0;
    // Pixel block of a ray-march wave (k_raymarch): 8 x 8 while a pixel step is about a texel or less (neighbouring rays share texels and
    // lines: the compact bundle wins, C3: 0.945 vs 0.972 ms), 16 x 4 once the lattice is sparser than the texels -- then nothing a wave
    // fetches is reused, the kernel runs at the memory side's pace and what counts is the lines per wave-sample, ~ rows x lines per row
    // (C5, 1.7 texels per pixel: 8.87 -> 7.57 ms, L2->fabric reads 58.7 -> 45.1 GB; 32 x 2 no better).  Texels per pixel at the grid centre:
    {
        const float dxc = g.gc[0] - cam->cam_pos[0], dyc = g.gc[1] - cam->cam_pos[1], dzc = g.gc[2] - cam->cam_pos[2];
        const float dist = std::sqrt(dxc * dxc + dyc * dyc + dzc * dzc);
        const float pixel = 2.0f * dist * (float)std::tan((double)cam->fov_y * 0.5) / (float)k->H;      // world size of a pixel there
        const float texel = g.s / (float)(g.nv - 2 * g.b);
        k->wave_lx = pixel > 1.25f * texel ? 4 : 3;
    }
#if VPFX_AB
    if (const char* e = std::getenv("VPFX_RM_WAVE_LX")) { const int v = std::atoi(e); if (v >= 3 && v <= 5) k->wave_lx = v; }
#endif
    k->texScale = (float)(g.nv - 2 * g.b);      // tc = (p + 0.5)(1 - 2b/nv) + b/nv; texel = tc*nv - 0.5   RM.shader:255-258
    k->texBias = (float)g.b - 0.5f;
    k->inv_soft = 1.0f / (float)rp->soft_distance;                           // rcp(_SoftDistance)       RM.shader:269
    k->alpha_cutoff = 0.0f;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Multi-GPU host logic: the slab cut and the compositing order of the slabs (no device work; exported through vp_plan_slabs /
// vp_blend_plan so that it is testable without a GPU).
// ---------------------------------------------------------------------------------------------------------------------------------
namespace {

// Optimal contiguous partition of w[0..nz) into `world` slabs of >= 1 slice each, minimising the heaviest slab (exact min-max DP;
// nz <= a few hundred, world <= 16).  Ties go to the cut whose last slab is closest to the uniform thickness, so a flat histogram gives
// uniform slabs.
void min_max_partition(int nz, int world, const std::vector<double>& w, int* cuts)
{
    std::vector<double> pre(nz + 1, 0.0);
    for (int z = 0; z < nz; ++z) pre[z + 1] = pre[z] + std::max(w[z], 0.0);
    if (!(pre[nz] > 0.0)) {                                  // no work anywhere: uniform
        for (int i = 0; i <= world; ++i) cuts[i] = (int)std::lround((double)i * nz / world);
        return;
    }
    const double INF = 1e300, uni = (double)nz / world;
    std::vector<std::vector<double>> dp(world + 1, std::vector<double>(nz + 1, INF));
    std::vector<std::vector<int>> arg(world + 1, std::vector<int>(nz + 1, 0));
    dp[0][0] = 0.0;
    for (int k = 1; k <= world; ++k)
        for (int z = k; z <= nz - (world - k); ++z) {
            double best = INF;
            int besty = k - 1;
            for (int y = k - 1; y < z; ++y) {                // last slab = [y, z)
                if (dp[k - 1][y] >= INF) continue;
                const double v = std::max(dp[k - 1][y], pre[z] - pre[y]);
                if (v < best || (v == best && std::fabs((z - y) - uni) < std::fabs((z - besty) - uni))) { best = v; besty = y; }
            }
            dp[k][z] = best; arg[k][z] = besty;
        }
    int z = nz;
    cuts[world] = nz;
    for (int k = world; k >= 1; --k) { z = arg[k][z]; cuts[k - 1] = z; }
}

// The finish pass of the split fill (stream (density, ao) back, store the bricks) as a share of the slab's local pass: 0.75 / 1.72, 0.38 / 0.94,
// 0.29 / 0.60 ms at 2 / 4 / 8 slabs of C3 (profiles/r03_scaling_model_C3_r8.txt).
#define VPFX_FINISH_SHARE 0.42
// Exact optimum of the one-group frame model over contiguous partitions into `world` slabs of >= 1 slice:
//     frame = max( F_0 + R_0 ,  max_r F_r  +  max_{r > 0} (C F_r + R_r) )
// F = per-slice cost of the fill's local pass, R = of the ray-march.  The local passes end in the all-gather of the transmittance maps, which
// completes when the slowest slab has filled (max_r F_r); after it every slab BUT THE FIRST runs its finish pass (C F) and its ray-march (R)
// back to back, and the image exchange waits for the slowest of those.  The first slab's fill is fused and needs nobody's light: since round 5
// the all-gather runs beside its compute stream (multi.cpp: exchange stream), so its march starts when its own fill ends -- its path is
// F_0 + R_0, whatever the others do.  For every candidate bound B on max_r F_r (all interval sums of F): a min-max DP over the partitions of
// the slices behind the first cut into world - 1 slabs that keep F <= B, then the best first cut; the best B wins.  nz^2 / 2 bounds x world x nz^2
// steps: ~4 M at nz = 32.
double two_maxima_partition(int nz, int world, const std::vector<double>& F, const std::vector<double>& R, double C, int* cuts)
{
    std::vector<double> pf(nz + 1, 0.0), pr(nz + 1, 0.0);
    for (int z = 0; z < nz; ++z) { pf[z + 1] = pf[z] + std::max(F[z], 0.0); pr[z + 1] = pr[z] + std::max(R[z], 0.0); }
    const double INF = 1e300;
    if (world == 1) { cuts[0] = 0; cuts[1] = nz; return pf[nz] + pr[nz]; }
    std::vector<double> bounds;
    for (int y = 0; y < nz; ++y)
        for (int z = y + 1; z <= nz; ++z) bounds.push_back(pf[z] - pf[y]);
    std::sort(bounds.begin(), bounds.end());
    bounds.erase(std::unique(bounds.begin(), bounds.end()), bounds.end());
    double best = INF;
    // g[k][y] = the smallest achievable max (C F + R) over the partitions of slices [y, nz) into k slabs with F <= B each; arg = the first cut
    std::vector<std::vector<double>> g(world, std::vector<double>(nz + 1));
    std::vector<std::vector<int>> arg(world, std::vector<int>(nz + 1));
    for (double B : bounds) {
        if (B >= best) break;                                // bounds ascend: no later one can win
        for (auto& row : g) std::fill(row.begin(), row.end(), INF);
        g[0][nz] = 0.0;
        for (int k = 1; k <= world - 1; ++k)
            for (int y = nz - k; y >= world - 1 - k + 1; --y)       // [y, nz) holds k slabs; world - 1 - k slabs + the first one still fit in front
                for (int z = y + 1; z <= nz - (k - 1); ++z) {
                    if (g[k - 1][z] >= INF || pf[z] - pf[y] > B) continue;
                    const double v = std::max(g[k - 1][z], (pr[z] - pr[y]) + C * (pf[z] - pf[y]));
                    if (v < g[k][y]) { g[k][y] = v; arg[k][y] = z; }
                }
        for (int z1 = 1; z1 <= nz - (world - 1); ++z1) {
            if (pf[z1] > B || g[world - 1][z1] >= INF) continue;
            // (the bound must be attained or undercut by the real maximum: evaluate the partition itself)
            double fmax = pf[z1];
            int y = z1;
            for (int k = world - 1; k >= 1; --k) { const int z = arg[k][y]; fmax = std::max(fmax, pf[z] - pf[y]); y = z; }
            const double t = std::max(pf[z1] + pr[z1], fmax + g[world - 1][z1]);
            if (t < best - 1e-12) {
                best = t;
                cuts[0] = 0; cuts[1] = z1;
                y = z1;
                for (int k = world - 1, i = 2; k >= 1; --k, ++i) { y = arg[k][y]; cuts[i] = y; }
            }
        }
    }
    return best;
}

}  // namespace

void hl_chain_groups(int world, int rm_groups, int* group_of_pos)
{
    const int G = std::min(std::max(rm_groups, 1), world);
    for (int p = 0; p < world; ++p) group_of_pos[p] = (int)(((long long)p * G) / world);
}

// The frame waits for the slowest slab in the fill's local pass (it ends in the all-gather of the transmittance maps), then every slab runs
// its finish pass (not the first one: fused fill) and its ray-march, and the image exchange waits for the slowest of those; with hand-off
// groups the ray-march waits for the slowest slab of every group in turn (groups run one after the other; within a group the slabs march
// concurrently).  The cut that balances fill + ray-march per slice need not minimise that.  Without a hand-off (one group) the modelled
// frame is a sum of two maxima, minimised exactly (two_maxima_partition).  With groups the candidates are that cut and the optimal min-max partitions of
// fill + alpha * raymarch for a few alpha (0 = fill only ... raymarch only), and the one with the smallest modelled frame wins (ties: the
// earliest candidate).  Groups are taken in rank order here (exact for a camera outside the grid along the light axis, the benchmark's
// case; a camera inside the grid reorders the chain around the straddling slab).
void hl_plan_slabs(int nz, int world, const double* fill_ms, const double* rm_ms, int rm_groups, int* cuts)
{
    if (world > nz) world = nz;
    std::vector<double> w(nz, 1.0);
    if (!fill_ms && !rm_ms) { min_max_partition(nz, world, w, cuts); return; }
    std::vector<int> gp(world);
    hl_chain_groups(world, rm_groups, gp.data());
    const double alphas[] = {-2.0, 0.0, 0.25, 0.5, 1.0, 2.0, 4.0, -1.0};
    double best_t = 1e300;
    std::vector<int> cand(world + 1);
    bool have = false;
    std::vector<double> Fv(nz), Rv(nz);
    for (int z = 0; z < nz; ++z) { Fv[z] = fill_ms ? fill_ms[z] : 0.0; Rv[z] = rm_ms ? rm_ms[z] : 0.0; }
    for (double a : alphas) {
        if (a == -2.0) {
            // the exact optimum for one group: nz^2 / 2 bounds x world x nz^2 steps -- 4 M at nz = 32, 34 M at 64, over half a billion at 128:
            // above 64 slices the frame path keeps to the alpha candidates
            if (nz > 64) continue;
            (void)two_maxima_partition(nz, world, Fv, Rv, VPFX_FINISH_SHARE, cand.data());
        } else {
            for (int z = 0; z < nz; ++z) w[z] = a < 0.0 ? Rv[z] : Fv[z] + a * Rv[z];
            min_max_partition(nz, world, w, cand.data());
        }
        double fmax = 0.0, finmax = 0.0, onemax = 0.0, first_path = 0.0;
        std::vector<double> gmax(world, 0.0);
        for (int r = 0; r < world; ++r) {
            double f = 0.0, m = 0.0;
            for (int z = cand[r]; z < cand[r + 1]; ++z) { f += fill_ms ? fill_ms[z] : 0.0; m += rm_ms ? rm_ms[z] : 0.0; }
            const double fin = r > 0 ? VPFX_FINISH_SHARE * f : 0.0;          // the fused first slab has no finish pass
            fmax = std::max(fmax, f);
            finmax = std::max(finmax, fin);
            if (r == 0) first_path = f + m; else onemax = std::max(onemax, fin + m);
            gmax[gp[r]] = std::max(gmax[gp[r]], m);
        }
        double t = fmax;
        if (gp[world - 1] == 0) t = std::max(first_path, fmax + onemax);   // one group: finish and ray-march of a slab run back to back; the fused first slab marches as soon as ITS fill is done
        else { t += finmax; for (double g : gmax) t += g; }           // chained groups (conservative: every finish before the first group)
        if (!have || t < best_t - 1e-12) { best_t = t; have = true; for (int i = 0; i <= world; ++i) cuts[i] = cand[i]; }
    }
}

// RenderMetavoxels at slab granularity (VPR.cs:652-711): phase A = slabs with slices zz <= zBoundary, zz ascending, blended OVER; phase B =
// slabs with slices zz > zBoundary, zz ascending, blended UNDER; the one slab that straddles zBoundary has an image in either phase.
// plan = the partial images in the reference's blend order.  chain = the same slabs FRONT TO BACK (reverse of phase A, then phase B):
// every slab earlier in the chain is composited in front of every later one -- for a phase-A slab behind the straddler only the
// straddler's phase-A image counts (its phase-B image is behind all of phase A).
int hl_blend_plan(int world, const int* cuts, int zb, int* chain, int* plan_rank, int* plan_which, int* plan_kind, int* straddler)
{
    int n = 0, strad = -1;
    for (int r = 0; r < world; ++r)
        if (cuts[r] <= zb) { plan_rank[n] = r; plan_which[n] = 0; plan_kind[n] = 0; ++n; }
    for (int r = 0; r < world; ++r) {
        const bool has_a = cuts[r] <= zb, has_b = cuts[r + 1] - 1 > zb;
        if (has_b) { plan_rank[n] = r; plan_which[n] = has_a ? 1 : 0; plan_kind[n] = 1; ++n; }
        if (has_a && has_b) strad = r;
    }
    int p = 0;
    if (strad >= 0) chain[p++] = strad;
    for (int r = world - 1; r >= 0; --r)
        if (cuts[r] <= zb && r != strad) chain[p++] = r;      // phase-A-only slabs, nearest the camera (highest zz) first
    for (int r = 0; r < world; ++r)
        if (cuts[r + 1] - 1 > zb && r != strad) chain[p++] = r;
    if (straddler) *straddler = strad;
    return n;
}

// The message schedule of the image exchange (vp_exchange_plan, include/vpfx.h): ONE definition, executed by multi.cpp on the GPUs and by the
// CPU tests over gloo.  tiles: all-to-all of screen pieces (+ the straddler's second image scattered) -> sharded blend -> gather on rank 0;
// all-gather (the north-star form): one all-gather of whole partial images (+ the straddler's second image to rank 0) -> blend on rank 0.
int hl_exchange_plan(int world, int rank, int strad, int all_gather, int phase, vp_xop* ops, int cap)
{
    int n = 0;
    auto put = [&](int kind, int peer, int buf, int index, int dbuf = -1, int dindex = -1) {
        if (n < cap) ops[n] = vp_xop{kind, peer, buf, index, dbuf, dindex};
        ++n;
    };
    if (!all_gather) {
        if (phase == 0) {
            for (int j = 0; j < world; ++j)
                if (j != rank) {
                    put(VP_XOP_RECV, j, VP_XBUF_PIECES, j);
                    put(VP_XOP_SEND, j, VP_XBUF_PRIMARY, j);
                }
            if (strad >= 0) {
                if (rank == strad) { for (int j = 0; j < world; ++j) if (j != rank) put(VP_XOP_SEND, j, VP_XBUF_SECOND, j); }
                else put(VP_XOP_RECV, strad, VP_XBUF_PIECES, world);
            }
            put(VP_XOP_COPY, -1, VP_XBUF_PRIMARY, rank, VP_XBUF_PIECES, rank);
            if (rank == strad) put(VP_XOP_COPY, -1, VP_XBUF_SECOND, rank, VP_XBUF_PIECES, world);
        } else {
            if (rank == 0) {
                for (int j = 1; j < world; ++j) put(VP_XOP_RECV, j, VP_XBUF_FINAL, j);
                put(VP_XOP_COPY, -1, VP_XBUF_PIECE_OUT, 0, VP_XBUF_FINAL, 0);
            } else {
                put(VP_XOP_SEND, 0, VP_XBUF_PIECE_OUT, 0);
            }
        }
    } else if (phase == 0) {
        put(VP_XOP_COPY, -1, VP_XBUF_PRIMARY, 0, VP_XBUF_PIECES, rank);
        put(VP_XOP_ALL_GATHER, -1, VP_XBUF_PIECES, rank);
        if (strad > 0) {
            if (rank == strad) put(VP_XOP_SEND, 0, VP_XBUF_SECOND, 0);
            else if (rank == 0) put(VP_XOP_RECV, strad, VP_XBUF_PIECES, world);
        }
        if (rank == 0 && strad == 0) put(VP_XOP_COPY, -1, VP_XBUF_SECOND, 0, VP_XBUF_PIECES, world);
    }
    return n;
}

