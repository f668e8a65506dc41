// bin.hip -- particle binning on the GPU: the reference's BinParticlesToMetavoxels (VPR.cs:397-457) and the
// per-particle part of FillMetavoxel's DisplacedParticle build (VPR.cs:575-588), restructured for CDNA4:
//
//   k_extract      caller records (AoS, arbitrary stride) -> SoA (world pos, size) + 64-byte fill record
//   k_bin<COUNT>   32 threads per particle, one candidate metavoxel each: candidate range + exact sphere/bordered-box test, atomic count
//   k_scan_*       two-launch tiled exclusive scan: CSR offsets, brick slots (z-major = draw order), totals
//   k_bin<SCATTER> same walk, atomic cursor -> unsorted CSR lists
//   k_sort_lists   per occupied MV: rank sort -> ascending particle index (the reference's list order, :452)
//   k_col_ordinal  per occupied MV: how many occupied MVs of its column lie in front of it (chained fill, fill.hip)
//
// The reference clears N^3 managed lists and rebuilds a 4x4 inverse per candidate on the CPU; here the bins
// are a CSR over an occupied-brick list and the per-MV inverse collapses to one shared 3x3 (rowsb) plus a
// per-MV translation.  All acceptance arithmetic follows DESIGN.md section 4 exactly (bit-identical lists).
#include "vpfx_internal.h"

#include <chrono>

namespace {

__device__ __forceinline__ float rd_f32(const uint8_t* p)
{
    // records may be only byte-aligned from the library's point of view
    uint32_t v = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
    return __uint_as_float(v);
}

// sin/cos of an angle in degrees: arithmetic spec 4.2 (exact quarter-turn reduction + fmaf polynomials).
__device__ __forceinline__ void sincos_deg(float deg, float& s, float& c)
{
    const float k = rintf(deg * (1.0f / 90.0f));
    const float r = fmaf(-90.0f, k, deg);
    const float x = r * 0.017453292519943295f;
    const float z = x * x;
    const float sp = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f), z * x, x);
    const float cp = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f),
                          z * z, fmaf(-0.5f, z, 1.0f));
    switch (((int)k) & 3) {
    case 0: s = sp; c = cp; break;
    case 1: s = cp; c = -sp; break;
    case 2: s = -sp; c = -cp; break;
    default: s = -cp; c = sp; break;
    }
}

__global__ void __launch_bounds__(256)
k_extract(const uint8_t* __restrict__ raw, int P, vp_particle_layout lay, PsysConsts ps,
          float4* __restrict__ ws4, float* __restrict__ rec)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const uint8_t* q = raw + (size_t)p * lay.stride;
    const float lx = rd_f32(q + lay.off_position), ly = rd_f32(q + lay.off_position + 4), lz = rd_f32(q + lay.off_position + 8);
    // wsParticlePos = particleSys.localToWorld.MultiplyPoint3x4(position)                     VPR.cs:418
    const float wx = ((ps.L2W[0] * lx + ps.L2W[1] * ly) + ps.L2W[2] * lz) + ps.L2W[3];
    const float wy = ((ps.L2W[4] * lx + ps.L2W[5] * ly) + ps.L2W[6] * lz) + ps.L2W[7];
    const float wz = ((ps.L2W[8] * lx + ps.L2W[9] * ly) + ps.L2W[10] * lz) + ps.L2W[11];
    const float size = rd_f32(q + lay.off_size);
    float rot = rd_f32(q + lay.off_rotation);
    if (ps.rot_in_radians) rot = rot * 57.29577951308232f;
    const float life = rd_f32(q + lay.off_lifetime), life0 = rd_f32(q + lay.off_start_lifetime);
    ws4[p] = make_float4(wx, wy, wz, size);

    // mWorldToLocal = TRS(wsPos, AngleAxis(rotation, psys.forward), size).inverse             VPR.cs:583
    float sh, ch;
    sincos_deg(rot * 0.5f, sh, ch);
    const float qx = ps.axis[0] * sh, qy = ps.axis[1] * sh, qz = ps.axis[2] * sh, qw = ch;
    const float x2 = qx + qx, y2 = qy + qy, z2 = qz + qz;
    const float xx = qx * x2, yy = qy * y2, zz = qz * z2, xy = qx * y2, xz = qx * z2, yz = qy * z2;
    const float wxx = qw * x2, wyy = qw * y2, wzz = qw * z2;
    float R[9];
    R[0] = 1.0f - (yy + zz); R[1] = xy - wzz;         R[2] = xz + wyy;
    R[3] = xy + wzz;         R[4] = 1.0f - (xx + zz); R[5] = yz - wxx;
    R[6] = xz - wyy;         R[7] = yz + wxx;         R[8] = 1.0f - (xx + yy);
    const float inv = 1.0f / size;
    float out[16];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float rx = R[0 * 3 + k] * inv, ry = R[1 * 3 + k] * inv, rz = R[2 * 3 + k] * inv;
        out[k * 4 + 0] = rx; out[k * 4 + 1] = ry; out[k * 4 + 2] = rz;
        out[k * 4 + 3] = -((rx * wx + ry * wy) + rz * wz);
    }
    out[12] = life / life0;            // mOpacity = lifetime / startLifetime                  VPR.cs:586
    out[13] = 0.f; out[14] = 0.f; out[15] = 0.f;   // per-frame slice step in particle space, filled by k_bin<0>
    float4* dst = reinterpret_cast<float4*>(rec + 16 * (size_t)p);
    dst[0] = make_float4(out[0], out[1], out[2], out[3]);
    dst[1] = make_float4(out[4], out[5], out[6], out[7]);
    dst[2] = make_float4(out[8], out[9], out[10], out[11]);
    dst[3] = make_float4(out[12], out[13], out[14], out[15]);
}

// BIN_LANES threads per particle, each taking every BIN_LANES-th candidate metavoxel of the particle's index range (27 candidates for a
// particle smaller than a metavoxel: one each).  One thread per particle walked them one after the other -- 27 dependent rounds of
// (position load, atomic) with six waves per CU: 42 us per pass at C3 for microseconds of work.  The order in which the lists are filled
// does not matter (k_sort_lists puts them into the reference's ascending-particle order).
// MODE 0: count, MODE 1: scatter, MODE 2: pairs per z-slice over the WHOLE grid (load-balancing histogram for the multi-GPU slab
// split).                                                                          VPR.cs:415-456
#define BIN_LANES 32
template <int MODE>
__global__ void __launch_bounds__(256)
k_bin(const float4* __restrict__ ws4, int P, GridConsts g, const float* __restrict__ mvPos,
      int* __restrict__ count_or_cursor, const int* __restrict__ offsets, int* __restrict__ ids, float* __restrict__ rec,
      const DevMeta* __restrict__ ahead_meta, int ahead_cap)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const int p = (int)(t / BIN_LANES), sub = (int)(t % BIN_LANES);
    if (p >= P) return;
    // scatter launched AHEAD of the host's look at the totals (launch_bin): it only runs if the lists fit the pool as allocated
    if (MODE == 1 && ahead_meta && (unsigned)ahead_meta->pairs > (unsigned)ahead_cap) return;
    const float4 w = ws4[p];
    if (MODE == 0 && sub == 0) {
        // B = W2P_linear * (_LightForward * oneVoxelSize): how far one slice step moves a voxel in this particle's
        // space (arithmetic spec 4.4).  Depends on the frame, so it is refreshed with every bin.
        float* r = rec + 16 * (size_t)p;
        const float dx = g.fwd[0] * g.one, dy = g.fwd[1] * g.one, dz = g.fwd[2] * g.one;
        r[13] = fmaf(r[2], dz, fmaf(r[1], dy, r[0] * dx));
        r[14] = fmaf(r[6], dz, fmaf(r[5], dy, r[4] * dx));
        r[15] = fmaf(r[10], dz, fmaf(r[9], dy, r[8] * dx));
    }
    // a particle with a non-finite position or size is skipped (undefined in the reference: C# (int)NaN, :434-438)
    if (!(fabsf(w.x) < INFINITY && fabsf(w.y) < INFINITY && fabsf(w.z) < INFINITY && fabsf(w.w) < INFINITY)) return;
    // lsParticlePos = light.worldToLocal * ws                                                  :419
    const float lx = ((g.Linv[0] * w.x + g.Linv[1] * w.y) + g.Linv[2] * w.z) + g.Linv[3];
    const float ly = ((g.Linv[4] * w.x + g.Linv[5] * w.y) + g.Linv[6] * w.z) + g.Linv[7];
    const float lz = ((g.Linv[8] * w.x + g.Linv[9] * w.y) + g.Linv[10] * w.z) + g.Linv[11];
    // pIndex = (ls - lsGridCenter)/s + N*0.5                                                   :422-423
    const float pix = (lx - g.lsO[0]) / g.s + (float)g.Nx * 0.5f;
    const float piy = (ly - g.lsO[1]) / g.s + (float)g.Ny * 0.5f;
    const float piz = (lz - g.lsO[2]) / g.s + (float)g.Nz * 0.5f;
    const float fe = (float)(int)rintf((w.w / 2.0f) / g.s);      // pExtents = RoundToInt(r/s)  :425
    // Vector3.Max(zero, ..) / Vector3.Min(limit, ..) then C# (int) truncation                  :431-438
    const int x0 = (int)fmaxf(0.f, pix - fe), x1 = (int)fminf((float)(g.Nx - 1), pix + fe);
    const int y0 = (int)fmaxf(0.f, piy - fe), y1 = (int)fminf((float)(g.Ny - 1), piy + fe);
    int z0 = (int)fmaxf(0.f, piz - fe), z1 = (int)fminf((float)(g.Nz - 1), piz + fe);
    // only the owned slab is binned (other slabs belong to other GPUs)
    if (MODE != 2) { z0 = max(z0, g.z0); z1 = min(z1, g.z1 - 1); }
    const float r = (w.w / 2.0f) / g.sb;                         // mvParticleRadius           :445
    const int nx = x1 - x0 + 1, ny = y1 - y0 + 1, nz = z1 - z0 + 1;
    if (nx <= 0 || ny <= 0 || nz <= 0) return;
    const int ncand = nx * ny * nz;
    for (int ci = sub; ci < ncand; ci += BIN_LANES) {
                const int xx = x0 + ci % nx, yy = y0 + (ci / nx) % ny, zz = z0 + ci / (nx * ny);
                const int mi = (zz * g.Ny + yy) * g.Nx + xx;
                const float mx = mvPos[3 * mi], my = mvPos[3 * mi + 1], mz = mvPos[3 * mi + 2];
                float r2 = r * r;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float a = g.rowsb[k * 3], bq = g.rowsb[k * 3 + 1], cq = g.rowsb[k * 3 + 2];
                    const float t = -((a * mx + bq * my) + cq * mz);
                    const float m = ((a * w.x + bq * w.y) + cq * w.z) + t;
                    // MathUtil.DoesBoxIntersectSphere                                     MathUtil.cs:15-22
                    if (m < -0.5f) { const float d = m - (-0.5f); r2 -= d * d; }
                    else if (m > 0.5f) { const float d = m - 0.5f; r2 -= d * d; }
                }
                if (r2 > 0.f) {
                    if (MODE == 2) atomicAdd(&count_or_cursor[zz], 1);
                    else if (MODE == 0) atomicAdd(&count_or_cursor[mi], 1);
                    else { const int slot = atomicAdd(&count_or_cursor[mi], 1); ids[offsets[mi] + slot] = p; }
                }
            }
}

// Exclusive scan of count[] -> offsets[], brick slots for occupied slab MVs in linear (= z-major draw) order, in
// fully coalesced launches over tiles of 1024 MVs: k_scan_tiles (per-tile totals) and k_scan_write (scans and writes its
// tile).  Up to SCAN_DIRECT_TILES tiles (4 M metavoxels; the benchmark grids have 32 / 256) every workgroup of k_scan_write
// re-reduces the tile totals in front of it itself -- cheaper than a third launch; beyond that (vp_create accepts 2^28
// metavoxels = 262 144 tiles, where that would be O(tiles^2)) k_scan_prefix turns the totals into exclusive prefixes first.
// Pair totals are accumulated in 64 bits: a CSR with more than INT_MAX pairs is reported (meta->pairs = -1), not wrapped.
#define SCAN_TILE 1024
#define SCAN_DIRECT_TILES 4096
struct TileTotals { long long pairs; int occ, mx; };

__device__ __forceinline__ void block_scan3(long long& a, int& b, int& m, long long* sh /* [3][16] */)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const long long ua = __shfl_up(a, d);
        const int ub = __shfl_up(b, d), um = __shfl_up(m, d);
        if (lane >= d) { a += ua; b += ub; m = max(m, um); }
    }
    if (lane == 63) { sh[wv] = a; sh[16 + wv] = b; sh[32 + wv] = m; }
    __syncthreads();
    long long ba = 0;
    int bb = 0, gm = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        if (w < wv) { ba += sh[w]; bb += (int)sh[16 + w]; }
        gm = max(gm, (int)sh[32 + w]);
    }
    a += ba; b += bb; m = gm;                        // inclusive over the workgroup; m = workgroup max
    __syncthreads();
}

__global__ void __launch_bounds__(SCAN_TILE)
k_scan_tiles(const int* __restrict__ count, int n3, int nxy, int z0, int z1, TileTotals* __restrict__ totals)
{
    __shared__ long long sh[48];
    const int i = blockIdx.x * SCAN_TILE + threadIdx.x;
    const int cnt = i < n3 ? count[i] : 0;
    const int zz = i / nxy;
    long long a = cnt;
    int b = (cnt != 0 && zz >= z0 && zz < z1) ? 1 : 0, m = cnt;
    block_scan3(a, b, m, sh);
    if (threadIdx.x == SCAN_TILE - 1) totals[blockIdx.x] = TileTotals{a, b, m};
}

// Large grids only (> SCAN_DIRECT_TILES tiles): exclusive prefix of the tile totals in place, one workgroup walking them in
// chunks of 1024; totals[ntiles] receives the grand totals (pairs, occupied, max).
__global__ void __launch_bounds__(SCAN_TILE)
k_scan_prefix(TileTotals* __restrict__ totals, int ntiles)
{
    __shared__ long long sh[48];
    __shared__ long long s_carry[3];
    if (threadIdx.x == 0) { s_carry[0] = 0; s_carry[1] = 0; s_carry[2] = 0; }
    __syncthreads();
    for (int base = 0; base < ntiles; base += SCAN_TILE) {
        const int t = base + threadIdx.x;
        const TileTotals tt = t < ntiles ? totals[t] : TileTotals{0, 0, 0};
        long long a = tt.pairs;
        int b = tt.occ, m = tt.mx;
        block_scan3(a, b, m, sh);
        const long long ca = s_carry[0];
        const int cb = (int)s_carry[1], cm = (int)s_carry[2];
        if (t < ntiles) totals[t] = TileTotals{ca + a - tt.pairs, cb + b - tt.occ, 0};
        __syncthreads();
        if (threadIdx.x == SCAN_TILE - 1) { s_carry[0] = ca + a; s_carry[1] = cb + b; s_carry[2] = max(cm, m); }
        __syncthreads();
    }
    if (threadIdx.x == 0) totals[ntiles] = TileTotals{s_carry[0], (int)s_carry[1], (int)s_carry[2]};
}

// The totals go straight into pinned, coherent host memory, followed by this frame's sequence number (system-scope release): the host
// polls that word (launch_bin) and reads the totals a microsecond after the scan has them -- while the kernels launched ahead of its wait
// (scatter, list sort) still run -- instead of sleeping until the whole stream has drained.
__device__ __forceinline__ void publish_totals(DevMeta* host_meta, const DevMeta& r, int seq)
{
    *host_meta = r;
    __hip_atomic_store(reinterpret_cast<int*>(host_meta + 1), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void __launch_bounds__(SCAN_TILE)
k_scan_write(const int* __restrict__ count, int n3, int nxy, int z0, int z1, const TileTotals* __restrict__ totals, int ntiles,
             int prefixed, int* __restrict__ offsets, int* __restrict__ brick_index, int* __restrict__ occ_list, int* __restrict__ cursor,
             DevMeta* __restrict__ meta, DevMeta* __restrict__ host_meta, int seq)
{
    __shared__ long long sh[48];
    __shared__ long long s_base[3];
    if (prefixed) {
        // k_scan_prefix has already turned the totals into exclusive prefixes (+ grand totals in slot ntiles)
        if (threadIdx.x == 0) { s_base[0] = totals[blockIdx.x].pairs; s_base[1] = totals[blockIdx.x].occ; s_base[2] = totals[ntiles].mx; }
    } else {
        // exclusive prefix of the tile totals in front of this tile (and the grand maximum)
        long long pa = 0;
        int pb = 0, pm = 0;
        for (int t = threadIdx.x; t < ntiles; t += SCAN_TILE) {
            const TileTotals tt = totals[t];
            if (t < (int)blockIdx.x) { pa += tt.pairs; pb += tt.occ; }
            pm = max(pm, tt.mx);
        }
        block_scan3(pa, pb, pm, sh);
        if (threadIdx.x == SCAN_TILE - 1) { s_base[0] = pa; s_base[1] = pb; s_base[2] = pm; }
    }
    __syncthreads();
    const long long base_pairs = s_base[0];
    const int base_occ = (int)s_base[1], gmax = (int)s_base[2];
    const int i = blockIdx.x * SCAN_TILE + threadIdx.x;
    const int cnt = i < n3 ? count[i] : 0;
    const int zz = i / nxy;
    const bool occ = cnt != 0 && zz >= z0 && zz < z1;                       // VPR.cs:511
    long long a = cnt;
    int b = occ ? 1 : 0, m = cnt;
    block_scan3(a, b, m, sh);
    if (i < n3) {
        offsets[i] = (int)(base_pairs + a - cnt);                           // only meaningful while the total fits (checked below)
        cursor[i] = 0;
        const int slot = base_occ + b - (occ ? 1 : 0);
        brick_index[i] = occ ? slot : -1;
        if (occ) occ_list[slot] = i;
    }
    if ((int)blockIdx.x == ntiles - 1 && threadIdx.x == SCAN_TILE - 1) {
        const long long total = base_pairs + a;
        offsets[n3] = (int)total;
        const DevMeta r{base_occ + b, total > 2147483647LL ? -1 : (int)total, gmax, 0};
        *meta = r;
        publish_totals(host_meta, r, seq);                                   // pinned host memory: no copy command for 16 bytes
    }
}

// Grids of at most SCAN_TILE metavoxels (the reference's own scene is 10^3): k_scan_tiles + k_scan_write + k_col_ordinal in ONE workgroup
// and one launch -- at these sizes the three launches were 15 us of a 48 us bin pass that does microseconds of work.
__global__ void __launch_bounds__(SCAN_TILE)
k_scan_small(const int* __restrict__ count, int n3, int nxy, int z0, int z1, int* __restrict__ offsets, int* __restrict__ brick_index,
             int* __restrict__ occ_list, int* __restrict__ cursor, int* __restrict__ ord, int* __restrict__ colcount,
             DevMeta* __restrict__ meta, DevMeta* __restrict__ host_meta, int seq)
{
    __shared__ long long sh[48];
    __shared__ unsigned char s_occ[SCAN_TILE];
    const int i = threadIdx.x;
    const int cnt = i < n3 ? count[i] : 0;
    const int zz = i / nxy;
    const bool occ = cnt != 0 && zz >= z0 && zz < z1;                       // VPR.cs:511
    s_occ[i] = occ ? 1 : 0;
    long long a = cnt;
    int b = occ ? 1 : 0, m = cnt;
    block_scan3(a, b, m, sh);                                               // (its barriers also publish s_occ)
    if (i < n3) {
        offsets[i] = (int)(a - cnt);
        cursor[i] = 0;
        const int slot = b - (occ ? 1 : 0);
        brick_index[i] = occ ? slot : -1;
        if (occ) {
            occ_list[slot] = i;
            const int col = i - zz * nxy;
            int n = 0;
            for (int z = z0; z < zz; ++z) n += s_occ[z * nxy + col];
            ord[i] = n;                                                     // position among the column's occupied metavoxels (k_col_ordinal)
        }
        if (i < nxy) {
            int n = 0;
            for (int z = z0; z < z1; ++z) n += s_occ[z * nxy + i];
            colcount[i] = n;
        }
    }
    if (i == SCAN_TILE - 1) {
        offsets[n3] = (int)a;
        const DevMeta r{b, a > 2147483647LL ? -1 : (int)a, m, 0};
        *meta = r;
        publish_totals(host_meta, r, seq);
    }
}

// Rank sort of one MV's list (ids are unique within a list).  One workgroup per occupied MV.
#define SORT_CAP 4096
__global__ void __launch_bounds__(256)
k_sort_lists(const int* __restrict__ occ_list, const int* __restrict__ offsets, const int* __restrict__ in,
             int* __restrict__ out, DevMeta* __restrict__ meta, int ahead_cap)
{
    __shared__ int s_ids[SORT_CAP];
    // launched ahead of the host's look at the totals (ahead_cap >= 0): one workgroup per metavoxel of the grid, those past the occupied
    // count leave; nothing runs if the lists do not fit the pool
    if (ahead_cap >= 0 && ((int)blockIdx.x >= meta->occupied || (unsigned)meta->pairs > (unsigned)ahead_cap)) return;
    const int mi = occ_list[blockIdx.x];
    const int off = offsets[mi], n = offsets[mi + 1] - off;
    if (n > SORT_CAP) {      // pathological list (> 4096 particles in one MV): the same rank sort straight from global memory
        for (int i = threadIdx.x; i < n; i += 256) {
            const int v = in[off + i];
            int rank = 0;
            for (int j = 0; j < n; ++j) rank += (in[off + j] < v) ? 1 : 0;
            out[off + rank] = v;
        }
        if (threadIdx.x == 0) atomicAdd(&meta->unsorted_lists, 1);      // counts the lists that took this slow path
        return;
    }
    for (int i = threadIdx.x; i < n; i += 256) s_ids[i] = in[off + i];
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 256) {
        const int v = s_ids[i];
        int rank = 0;
        for (int j = 0; j < n; ++j) rank += (s_ids[j] < v) ? 1 : 0;
        out[off + rank] = v;
    }
}


// Chained fill (fill.hip): position of every occupied MV of the owned slab among the occupied MVs of its (xx, yy) column, front to back,
// and the number of occupied MVs per column.
__global__ void __launch_bounds__(256)
k_col_ordinal(const int* __restrict__ brick_index, int nxy, int z0, int z1, int* __restrict__ ord, int* __restrict__ colcount)
{
    // one wave per column, one lane per slice (64 at a time): the ordinal is the number of occupied slices below the lane in the ballot.
    // (One thread per column walked its slices one dependent load after the other: 11 us at C3 for 1 024 threads.)
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= nxy) return;
    int n = 0;
    for (int zb = z0; zb < z1; zb += 64) {
        const int zz = zb + lane;
        const bool occ = zz < z1 && brick_index[zz * nxy + i] >= 0;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(occ);
        if (occ) ord[zz * nxy + i] = n + __builtin_popcountll(m & ((1ull << lane) - 1ull));
        n += __builtin_popcountll(m);
    }
    if (lane == 0) colcount[i] = n;
}

}  // namespace

int launch_extract(vp_ctx* c)
{
    if (c->P == 0) return VP_OK;
    hipLaunchKernelGGL(k_extract, dim3((c->P + 255) / 256), dim3(256), 0, c->stream,
                       c->d_raw, c->P, c->lay, c->psys, c->d_ws, c->d_rec);
    VP_HIP(hipGetLastError());
    return VP_OK;
}

int launch_z_histogram(vp_ctx* c, int* d_hist)
{
    const GridConsts& g = c->g;
    VP_HIP(hipMemsetAsync(d_hist, 0, (size_t)g.Nz * sizeof(int), c->stream));
    if (c->P > 0)
        hipLaunchKernelGGL(k_bin<2>, dim3((unsigned)(((size_t)c->P * BIN_LANES + 255) / 256)), dim3(256), 0, c->stream, c->d_ws, c->P, g, c->d_mvPos, d_hist,
                           (const int*)nullptr, (int*)nullptr, c->d_rec, (const DevMeta*)nullptr, 0);
    VP_HIP(hipGetLastError());
    return VP_OK;
}

int launch_bin(vp_ctx* c)
{
    const GridConsts& g = c->g;
    const int n3 = (int)c->n3, nxy = g.Nx * g.Ny;
    const unsigned nb = (unsigned)(((size_t)c->P * BIN_LANES + 255) / 256);      // BIN_LANES threads per particle
    VP_HIP(hipMemsetAsync(c->d_count, 0, c->n3 * sizeof(int), c->stream));
    VP_HIP(hipEventRecord(c->ev[0][0], c->stream));
    if (c->P > 0) {
        hipLaunchKernelGGL(k_bin<0>, dim3(nb), dim3(256), 0, c->stream, c->d_ws, c->P, g, c->d_mvPos, c->d_count,
                           (const int*)nullptr, (int*)nullptr, c->d_rec, (const DevMeta*)nullptr, 0);
    }
    const bool small = n3 <= SCAN_TILE;
    const int seq = ++c->bin_seq;                           // this frame's tag of the totals in pinned host memory (never 0 = the initial value)
    if (c->bin_seq == 0x7fffffff) c->bin_seq = 0;
    if (small) {
        hipLaunchKernelGGL(k_scan_small, dim3(1), dim3(SCAN_TILE), 0, c->stream, c->d_count, n3, nxy, g.z0, g.z1, c->d_offsets, c->d_brick_index,
                           c->d_occ_list, c->d_cursor, c->d_ord, c->d_colcount, c->d_meta, c->d_meta_host, seq);
    } else {
        const int ntiles = (n3 + SCAN_TILE - 1) / SCAN_TILE;
        hipLaunchKernelGGL(k_scan_tiles, dim3(ntiles), dim3(SCAN_TILE), 0, c->stream, c->d_count, n3, nxy, g.z0, g.z1,
                           (TileTotals*)c->d_scan_totals);
        const int prefixed = ntiles > SCAN_DIRECT_TILES ? 1 : 0;
        if (prefixed) hipLaunchKernelGGL(k_scan_prefix, dim3(1), dim3(SCAN_TILE), 0, c->stream, (TileTotals*)c->d_scan_totals, ntiles);
        hipLaunchKernelGGL(k_scan_write, dim3(ntiles), dim3(SCAN_TILE), 0, c->stream, c->d_count, n3, nxy, g.z0, g.z1,
                           (const TileTotals*)c->d_scan_totals, ntiles, prefixed, c->d_offsets, c->d_brick_index, c->d_occ_list, c->d_cursor, c->d_meta,
                           c->d_meta_host, seq);
        hipLaunchKernelGGL(k_col_ordinal, dim3((nxy + 3) / 4), dim3(256), 0, c->stream, c->d_brick_index, nxy, g.z0, g.z1, c->d_ord, c->d_colcount);
    }
    // The totals are needed on the host to size the pair and brick pools.  While the host waits for them the GPU already fills the lists
    // into the pool as it stands: both kernels check the totals against its capacity themselves and do nothing if it is too small (a
    // frame whose cloud outgrew the pool: re-launched below after the reallocation).  Large grids sort after the wait -- one workgroup
    // per metavoxel of the GRID is only free while that is a thousand.
    const int ahead_cap = (int)(c->pairs_cap < 2147483647u ? c->pairs_cap : 2147483647u);
    const bool ahead = c->P > 0 && c->pairs_cap > 0, ahead_sort = ahead && small;
    if (ahead) {
        hipLaunchKernelGGL(k_bin<1>, dim3(nb), dim3(256), 0, c->stream, c->d_ws, c->P, g, c->d_mvPos, c->d_cursor,
                           (const int*)c->d_offsets, c->d_ids_tmp, c->d_rec, (const DevMeta*)c->d_meta, ahead_cap);
        if (ahead_sort)
            hipLaunchKernelGGL(k_sort_lists, dim3(n3), dim3(256), 0, c->stream, c->d_occ_list, c->d_offsets,
                               (const int*)c->d_ids_tmp, c->d_ids, c->d_meta, ahead_cap);
    }
    VP_HIP(hipGetLastError());
    // wait for the TOTALS, not for the stream: poll the sequence word the scan writes behind them.  The stream is asked now and then (a failed
    // launch would never publish); after a while the host stops spinning and sleeps on the stream as before (the GPU may be a frame behind).
    {
        const volatile int* seq_word = reinterpret_cast<const volatile int*>(c->h_meta_host + 1);
        const auto t_begin = std::chrono::steady_clock::now();
        bool seen = false;
        for (unsigned spin = 0; !seen; ++spin) {
            if (__atomic_load_n(const_cast<const int*>(seq_word), __ATOMIC_ACQUIRE) == seq) { seen = true; break; }
            if ((spin & 1023u) == 1023u) {
                const hipError_t q = hipStreamQuery(c->stream);
                if (q == hipSuccess) break;                               // everything has run: the totals are there
                if (q != hipErrorNotReady) return vp_fail(c, VP_ERR_HIP, "vp_bin: %s", hipGetErrorString(q));
                if (std::chrono::steady_clock::now() - t_begin > std::chrono::microseconds(400)) break;
            }
        }
        if (!seen) VP_HIP(hipStreamSynchronize(c->stream));
    }
    c->h_meta = *c->h_meta_host;
    if (c->h_meta.pairs < 0)
        return vp_fail(c, VP_ERR_UNSUPPORTED, "vp_bin: more than INT_MAX (particle, metavoxel) pairs; the CSR offsets are 32-bit");
    const size_t pairs = (size_t)c->h_meta.pairs;
    bool scattered = ahead, sorted = ahead_sort;
    if (pairs > c->pairs_cap) {
        if (c->d_ids) VP_HIP(hipFree(c->d_ids));
        if (c->d_ids_tmp) VP_HIP(hipFree(c->d_ids_tmp));
        c->d_ids = c->d_ids_tmp = nullptr;
        c->pairs_cap = pairs + pairs / 4 + 1024;
        VP_HIP(hipMalloc(&c->d_ids, c->pairs_cap * sizeof(int)));
        VP_HIP(hipMalloc(&c->d_ids_tmp, c->pairs_cap * sizeof(int)));
        scattered = sorted = false;                        // the kernels launched ahead saw that the pool was too small and left
    }
    if (c->P > 0 && pairs > 0) {
        if (!scattered)
            hipLaunchKernelGGL(k_bin<1>, dim3(nb), dim3(256), 0, c->stream, c->d_ws, c->P, g, c->d_mvPos, c->d_cursor,
                               (const int*)c->d_offsets, c->d_ids_tmp, c->d_rec, (const DevMeta*)nullptr, 0);
        if (!sorted)
            hipLaunchKernelGGL(k_sort_lists, dim3(c->h_meta.occupied), dim3(256), 0, c->stream, c->d_occ_list, c->d_offsets,
                               (const int*)c->d_ids_tmp, c->d_ids, c->d_meta, -1);
        VP_HIP(hipGetLastError());
    }
    VP_HIP(hipEventRecord(c->ev[0][1], c->stream));
    c->ev_valid[0] = true;
    return VP_OK;
}
