"""Independent float64 numpy restatement of the hot path, used ONLY to pin the C oracle.

TEST INFRASTRUCTURE (parity unpinned, see vp_oracle.c).  Written from the reference's shader / C# text with a
different structure than vp_oracle.c (vectorised per metavoxel, float64, literal `voxelWorldPos += fwd*one`
stepping, per-draw 4x4 matrices built with numpy.linalg.inv) so that a transcription error in either one shows
up as a disagreement.  Small scenes only.

Reference files: VPR.cs = Assets/Main Scene/VolumetricParticleRenderer.cs, Fill/RM = the two metavoxel shaders.
"""
from __future__ import annotations

import math

import numpy as np


def _colmajor_to_mat(m16):
    return np.asarray(m16, dtype=np.float64).reshape(4, 4).T


def _trs(pos, R, s):
    m = np.eye(4)
    m[:3, :3] = R * s
    m[:3, 3] = pos
    return m


def _angle_axis(deg, axis):
    a = math.radians(deg)
    x, y, z = axis / np.linalg.norm(axis)
    c, s = math.cos(a), math.sin(a)
    C = 1 - c
    return np.array([[c + x * x * C, x * y * C - z * s, x * z * C + y * s],
                     [y * x * C + z * s, c + y * y * C, y * z * C - x * s],
                     [z * x * C - y * s, z * y * C + x * s, c + z * z * C]])


def _f16(x):
    return np.asarray(x, dtype=np.float64).astype(np.float32).astype(np.float16)


class Twin:
    def __init__(self, sc):
        self.sc = sc
        self.Nx, self.Ny, self.Nz = sc.N
        self.nv, self.b, self.s = sc.nv, sc.border, float(sc.mv_scale)
        self.sb = self.s * self.nv / (self.nv - 2 * self.b)                     # VPR.cs:139
        self.one = self.sb / self.nv                                            # Fill.shader:160
        self.L = _colmajor_to_mat(sc.light_to_world)
        self.Linv = np.linalg.inv(self.L)
        self.Rl = self.L[:3, :3].copy()
        self.fwd = self.Rl[:, 2] / np.linalg.norm(self.Rl[:, 2])
        self.gc = np.asarray(sc.grid_center, dtype=np.float64)
        self.psys = _colmajor_to_mat(sc.psys_local_to_world)

    # ---- A.1 grid -------------------------------------------------------------------------------------
    def grid(self):
        lsO = (self.Linv @ np.append(self.gc, 1.0))[:3]
        pos = np.zeros((self.Nz, self.Ny, self.Nx, 3))
        for zz in range(self.Nz):
            for yy in range(self.Ny):
                for xx in range(self.Nx):
                    off = np.array([self.Nx // 2 - xx, self.Ny // 2 - yy, self.Nz // 2 - zz]) * self.s   # :388
                    pos[zz, yy, xx] = (self.L @ np.append(lsO - off, 1.0))[:3]
        self.mvPos, self.lsO = pos, lsO
        return pos

    # ---- A.2 binning ----------------------------------------------------------------------------------
    def bin(self):
        sc = self.sc
        P = len(sc.particles)
        self.ws = (self.psys[:3, :3] @ sc.particles["position"].astype(np.float64).T).T + self.psys[:3, 3]
        self.size = sc.particles["size"].astype(np.float64)
        rot = sc.particles["rotation"].astype(np.float64)
        self.rot_deg = np.degrees(rot) if sc.layout.rotation_in_radians else rot
        self.opacity = sc.particles["lifetime"].astype(np.float64) / sc.particles["startLifetime"].astype(np.float64)
        lists = {}
        N = np.array([self.Nx, self.Ny, self.Nz], dtype=np.float64)
        for p in range(P):
            ls = (self.Linv @ np.append(self.ws[p], 1.0))[:3]
            pidx = (ls - self.lsO) / self.s + N * 0.5                                            # :422-423
            e = float(np.rint((self.size[p] / 2.0) / self.s))                                   # RoundToInt, half-to-even
            lo = np.maximum(0.0, pidx - e)
            hi = np.minimum(N - 1, pidx + e)
            lo_i, hi_i = np.trunc(lo).astype(int), np.trunc(hi).astype(int)                     # C# (int)
            for zz in range(lo_i[2], hi_i[2] + 1):
                for yy in range(lo_i[1], hi_i[1] + 1):
                    for xx in range(lo_i[0], hi_i[0] + 1):
                        w2m = np.linalg.inv(_trs(self.mvPos[zz, yy, xx], self.Rl, self.sb))     # :440-442
                        m = (w2m @ np.append(self.ws[p], 1.0))[:3]
                        r = (self.size[p] / 2.0) / self.sb
                        r2 = r * r
                        for k in range(3):                                                       # MathUtil.cs:15-22
                            if m[k] < -0.5:
                                r2 -= (m[k] + 0.5) ** 2
                            elif m[k] > 0.5:
                                r2 -= (m[k] - 0.5) ** 2
                        if r2 > 0:
                            lists.setdefault((zz, yy, xx), []).append(p)
        self.lists = lists
        return lists

    # ---- B.5 cubemap ----------------------------------------------------------------------------------
    def _cube(self, d):
        cube = self.sc.cubemap.astype(np.float64)
        S = cube.shape[1]
        x, y, z = d[..., 0], d[..., 1], d[..., 2]
        ax, ay, az = np.abs(x), np.abs(y), np.abs(z)
        isz = (az >= ax) & (az >= ay)              # ties: z before y before x (GCN cube instructions; D3D leaves it open)
        isy = ~isz & (ay >= ax)
        isx = ~isz & ~isy
        ma = np.where(isx, ax, np.where(isy, ay, az))
        face = np.where(isx, np.where(x >= 0, 0, 1), np.where(isy, np.where(y >= 0, 2, 3), np.where(z >= 0, 4, 5)))
        sc_ = np.where(isx, np.where(x >= 0, -z, z), np.where(isy, x, np.where(z >= 0, x, -x)))
        tc_ = np.where(isx, -y, np.where(isy, np.where(y >= 0, z, -z), -y))
        ma = np.where(ma > 0, ma, 1.0)
        u, v = sc_ / ma, tc_ / ma
        fx, fy = (u + 1) / 2 * S - 0.5, (v + 1) / 2 * S - 0.5
        x0, y0 = np.floor(fx), np.floor(fy)
        tx, ty = fx - x0, fy - y0
        c = lambda i: np.clip(i, 0, S - 1).astype(int)
        t00, t10 = cube[face, c(y0), c(x0)], cube[face, c(y0), c(x0 + 1)]
        t01, t11 = cube[face, c(y0 + 1), c(x0)], cube[face, c(y0 + 1), c(x0 + 1)]
        a, b = t00 + tx * (t10 - t00), t01 + tx * (t11 - t01)
        return a + ty * (b - a)

    # ---- A.3 / A.4 fill -------------------------------------------------------------------------------
    def fill(self):
        sc, nv = self.sc, self.nv
        LW = self.Nx * nv
        self.light = np.ones((self.Ny * nv, LW))                                                # GL.Clear(Color.red)
        self.bricks = {}
        amb = np.array(sc.ambient, dtype=np.float64)
        psys_fwd = self.psys[:3, 2] / np.linalg.norm(self.psys[:3, 2])
        camp = self.gc - self.fwd * 200.0                                                       # VPR.cs:365
        w2l = np.linalg.inv(_trs(camp, self.Rl, 1.0))
        px, py = np.meshgrid(np.arange(nv) + 0.5, np.arange(nv) + 0.5, indexing="xy")           # px varies along axis 1
        bidx = nv - min(max(self.b, 0), nv - 2)
        for zz in range(self.Nz):                                                               # z-major draw order :505
            for yy in range(self.Ny):
                for xx in range(self.Nx):
                    ids = self.lists.get((zz, yy, xx))
                    if not ids:
                        continue
                    m2w = _trs(self.mvPos[zz, yy, xx], self.Rl, self.sb)
                    norm0 = np.stack([(px - nv / 2) / nv, (py - nv / 2) / nv, np.full_like(px, (0 - nv / 2) / nv),
                                      np.ones_like(px)], axis=-1)                                # Fill.shader:103
                    v0 = norm0 @ m2w.T                                                          # [py][px][4]
                    dens = np.zeros((nv, nv, nv))
                    ao = np.zeros((nv, nv, nv))
                    for j, p in enumerate(ids):
                        w2p = np.linalg.inv(_trs(self.ws[p], _angle_axis(self.rot_deg[p], psys_fwd), self.size[p]))  # :583
                        v = v0[..., :3].copy()
                        for sl in range(nv):
                            ps = v @ w2p[:3, :3].T + w2p[:3, 3]
                            d2 = (ps * ps).sum(-1)
                            hit = d2 <= 0.25
                            raw = self._cube(2 * ps)
                            net = sc.displacement_scale * raw + (1.0 - sc.displacement_scale)
                            dq = (4 * d2 - net) / (0.7 * net - net)
                            t = np.clip(dq, 0, 1)
                            den = t * t * (3 - 2 * t) * sc.opacity_factor
                            if sc.fade == 1:
                                den = den * self.opacity[p]
                            dens[sl] += np.where(hit, den, 0.0)
                            ao[sl] = np.where(hit, np.maximum(ao[sl], net), ao[sl])
                            v = v + self.fwd * self.one                                          # :183,207
                    # shadow index + propagate                                                   Fill.shader:211-269
                    z0 = (v0 @ w2l.T)[..., 2]
                    Y, X = np.arange(nv) + yy * nv, np.arange(nv) + xx * nv
                    dm = sc.light_depth_map[np.ix_(Y, X)].astype(np.float64) if sc.light_depth_map is not None else np.ones((nv, nv))
                    scene_z = dm * (1000.0 - 0.3) + 0.3
                    shadow = np.trunc((scene_z - z0) / self.one)
                    T = np.ones((nv, nv)) if zz == 0 else self.light[np.ix_(Y, X)].copy()
                    prop = T.copy()
                    brick = np.zeros((nv, nv, nv, 4), dtype=np.float16)
                    for sl in range(nv):
                        insh = sl >= shadow
                        T = np.where(insh, 0.0, T)
                        if sl < bidx:
                            prop = np.where(insh, prop, T)
                        rgb = 0.4 * T[..., None] + amb[None, None, :] * ao[sl][..., None]
                        brick[sl, :, :, :3] = _f16(rgb)
                        brick[sl, :, :, 3] = _f16(dens[sl])
                        T = T / (1.0 + dens[sl])
                    self.light[np.ix_(Y, X)] = prop
                    self.bricks[(zz, yy, xx)] = brick

    # ---- A.5 / A.6 ray-march --------------------------------------------------------------------------
    def raymarch(self):
        sc, nv = self.sc, self.nv
        W, H = sc.width, sc.height
        w2c, c2w = np.asarray(sc.world_to_cam, dtype=np.float64), np.asarray(sc.cam_to_world, dtype=np.float64)
        cam_pos = np.asarray(sc.cam_pos, dtype=np.float64)
        col, row = np.meshgrid(np.arange(W) + 0.5, np.arange(H) + 0.5, indexing="xy")
        d = np.stack([(2 * col / W - 1) * (W / H), 2 * row / H - 1, np.full_like(col, -1 / math.tan(math.radians(sc.fov_y_deg) / 2))], -1)
        d /= np.linalg.norm(d, axis=-1, keepdims=True)
        csO = (w2c @ np.append(self.gc, 1.0))[:3]
        maxdim = max(self.Nx, self.Ny, self.Nz)
        halfz = 1.73205 * 0.5 * maxdim * self.s
        zmin = csO[2] + halfz
        start = d * (zmin / d[..., 2:3])
        step = ((2 * halfz) / self.s) / (maxdim * sc.steps)
        bo = self.b / nv
        # order                                                                                 VPR.cs:613-711
        keys = [((self.mvPos[0, yy, xx] - cam_pos) ** 2).sum() for yy in range(self.Ny) for xx in range(self.Nx)]
        asc = sorted(range(len(keys)), key=lambda i: keys[i])                                   # stable
        lsCam = (self.Linv @ np.append(cam_pos, 1.0))[:3]
        lsFirst = (self.Linv @ np.append(self.mvPos[0, 0, 0], 1.0))[2]
        zb = int(np.clip(np.rint((lsCam[2] - lsFirst) / self.s), -1, self.Nz - 1))
        draws = [(zz, i, 0) for zz in range(0, zb + 1) for i in reversed(asc)] + \
                [(zz, i, 1) for zz in range(zb + 1, self.Nz) for i in asc]
        dst = np.zeros((H, W, 4))
        samples = 0
        for zz, i, kind in draws:
            yy, xx = divmod(i, self.Nx)
            brick = self.bricks.get((zz, yy, xx))
            if brick is None:
                continue
            c2m = np.linalg.inv(_trs(self.mvPos[zz, yy, xx], self.Rl, self.s)) @ c2w           # :774-778
            o = start @ c2m[:3, :3].T + c2m[:3, 3]
            dm = d @ c2m[:3, :3].T
            dm /= np.linalg.norm(dm, axis=-1, keepdims=True)
            with np.errstate(divide="ignore", invalid="ignore"):
                inv = 1.0 / dm
                tb, tt = inv * (-0.5 - o), inv * (0.5 - o)
            t1 = np.fmax.reduce(np.fmin(tt, tb), axis=-1)
            t2 = np.fmin.reduce(np.fmax(tt, tb), axis=-1)
            exit_depth = -(start[..., 2] + d[..., 2] * t2 * self.s)
            cover = (t1 <= t2) & (exit_depth > sc.near) & (exit_depth <= sc.far)
            if sc.scene_depth is not None:
                cover &= exit_depth < sc.scene_depth
            if not cover.any():
                continue
            t_entry = np.ceil(t1 / step)
            t_exit = np.floor(t2 / step)
            camm = c2m[:3, 3]
            t_cam = np.trunc(np.linalg.norm(camm - o, axis=-1) / step)
            t_entry = np.maximum(t_entry, t_cam)
            res = np.zeros((H, W, 3))
            trans = np.ones((H, W))
            kmax = int(np.nanmax(np.where(cover, t_exit, -1e9)))
            kmin = int(np.nanmin(np.where(cover, t_entry, 1e9)))
            f32 = brick.astype(np.float64)
            for k in range(kmax, kmin - 1, -1):                                                # back to front
                act = cover & (k <= t_exit) & (k >= t_entry)
                if not act.any():
                    continue
                p = o + k * step * dm
                tc = (p + 0.5) * (1 - 2 * bo) + bo
                f = tc * nv - 0.5
                f0 = np.floor(f)
                w = f - f0
                i0 = np.mod(f0.astype(int), nv)
                i1 = np.mod(i0 + 1, nv)
                def tx(iz, iy, ix):
                    return f32[iz, iy, ix]
                wx, wy, wz = w[..., 0:1], w[..., 1:2], w[..., 2:3]
                c00 = tx(i0[..., 2], i0[..., 1], i0[..., 0]) * (1 - wx) + tx(i0[..., 2], i0[..., 1], i1[..., 0]) * wx
                c10 = tx(i0[..., 2], i1[..., 1], i0[..., 0]) * (1 - wx) + tx(i0[..., 2], i1[..., 1], i1[..., 0]) * wx
                c01 = tx(i1[..., 2], i0[..., 1], i0[..., 0]) * (1 - wx) + tx(i1[..., 2], i0[..., 1], i1[..., 0]) * wx
                c11 = tx(i1[..., 2], i1[..., 1], i0[..., 0]) * (1 - wx) + tx(i1[..., 2], i1[..., 1], i1[..., 0]) * wx
                c = (c00 * (1 - wy) + c10 * wy) * (1 - wz) + (c01 * (1 - wy) + c11 * wy) * wz
                rho = c[..., 3]
                soft = (k - t_cam) < sc.soft_distance
                rho = np.where(soft, rho * (k - t_cam) / sc.soft_distance, rho)
                bf = 1.0 / (1.0 + rho)
                res = np.where(act[..., None], c[..., :3] + bf[..., None] * (res - c[..., :3]), res)
                trans = np.where(act, trans * bf, trans)
                samples += int(act.sum())
            src = np.concatenate([res, (1 - trans)[..., None]], -1)
            if kind == 0:
                new = src + dst * (1 - src[..., 3:4])
            else:
                new = src * (1 - dst[..., 3:4]) + dst
            dst = np.where(cover[..., None], new, dst)
        self.samples = samples
        self.z_boundary = zb
        return dst
