"""GPU parity on the edge cases and option branches of the path (all through the C ABI, all against the oracle)."""
import ctypes as C

import numpy as np
import pytest
import torch

from vpfx_amd import abi, engine as E, scene as S
from vpfx_amd.manager import MetavoxelManager
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def both(sc, **kw):
    o, g = O.Oracle(sc.config()), E.Engine(sc.config(), **kw)
    for x in (o, g):
        x.set_frame(sc.light_to_world, sc.grid_center)
        x.bin(sc.particles, sc.layout, sc.psys_local_to_world)
        x.fill(sc.fill_params())
    return o, g


def check(sc, exact=True, rgba_tol=1e-3, near_zero_abs=0.0, **kw):
    o, g = both(sc, exact=exact, early_out=False, **kw)
    np.testing.assert_array_equal(o.bin_counts(), g.bin_counts())
    co = o.bin_counts()
    for zz, yy, xx in zip(*np.nonzero(co)):
        a, b = o.read_brick(xx, yy, zz).view(np.uint16), g.read_brick(xx, yy, zz).view(np.uint16)
        if exact:
            assert np.array_equal(a, b), (xx, yy, zz)
        else:
            du = np.abs(a.astype(np.int32) - b.astype(np.int32))
            if du.max() > 1:
                # default math: <= 1 fp16 ulp, except in near-zero values, where the smoothstep's t = 10/3 - (40/3) d2 / net cancels and the
                # reciprocal's last bit shows as a few ulp of an almost-zero density: bounded ABSOLUTELY there (the rule of the randomised
                # sweep, scripts/fuzz_parity.py); the caller passes the bound it accepts (0 = none)
                fa, fb = o.read_brick(xx, yy, zz).astype(np.float32), g.read_brick(xx, yy, zz).astype(np.float32)
                worst_abs = float(np.abs(fa - fb)[du > 1].max())
                assert worst_abs <= near_zero_abs, (xx, yy, zz, int(du.max()), worst_abs)
    np.testing.assert_allclose(g.read_lightmap(), o.read_lightmap(), rtol=1e-5, atol=1e-9)
    io, ig = o.raymarch(sc.camera(), sc.raymarch_params()), g.raymarch(sc.camera(), sc.raymarch_params())
    assert np.abs(io - ig).max() <= rgba_tol
    assert o.stats()["samples"] == g.stats()["samples"]        # every lattice sample of the oracle, none more
    return o, g, io, ig


def test_empty_particle_array():
    sc = S.make_scene("e", dims=(4, 16, 0, 64, 48))
    o, g, io, ig = check(sc)
    assert g.stats()["occupied_mv"] == 0
    np.testing.assert_array_equal(g.read_lightmap(), 1.0)
    np.testing.assert_array_equal(ig, 0.0)


def test_non_finite_particles_are_skipped():
    """NaN / inf positions or sizes (undefined in the reference: (int)NaN indexes its grid) are skipped; everything else is
    unchanged, ids included.  Zero-size particles cover nothing."""
    sc = S.make_scene("T0")
    ref_o, ref_g = both(sc)
    img_ref = ref_g.raymarch(sc.camera(), sc.raymarch_params())
    P = len(sc.particles)
    extra = np.repeat(sc.particles[:6], 1, axis=0).copy()
    raw = extra.view(np.float32).reshape(6, -1)
    lay = sc.layout
    pos, size = lay.off_position // 4, lay.off_size // 4
    raw[0, pos] = np.nan
    raw[1, pos + 1] = np.inf
    raw[2, pos + 2] = -np.inf
    raw[3, size] = np.nan
    raw[4, size] = np.inf
    raw[5, size] = 0.0
    sc.particles = np.concatenate([sc.particles, extra])
    o, g = both(sc)
    assert g.stats()["particles"] == P + 6
    np.testing.assert_array_equal(g.bin_counts(), ref_g.bin_counts())
    np.testing.assert_array_equal(o.bin_counts(), ref_o.bin_counts())
    img = g.raymarch(sc.camera(), sc.raymarch_params())
    assert np.isfinite(img).all() and np.array_equal(img, img_ref)
    assert np.array_equal(g.read_lightmap(), ref_g.read_lightmap())


def test_fade_and_radians_and_moved_particle_system():
    sc = S.make_scene("x", dims=(6, 16, 300, 96, 64), rotation_in_radians=True, fade=1)
    rot = S.quat_to_matrix((0.1, 0.3, -0.2, 0.927))
    sc.psys_local_to_world = S.to_colmajor16(S.trs((0.7, -0.4, 1.1), rot))
    sc.grid_center = np.array([0.5, -0.25, 0.75], dtype=np.float32)
    check(sc)


@pytest.mark.parametrize("width,height,rolled", [(96, 64, False), (61, 47, False), (96, 64, True), (640, 400, False)])
def test_both_wave_pixel_block_shapes_render_the_oracle_frame(width, height, rolled):
    """k_raymarch's wave covers 8 x 8 pixels while a pixel is about a texel or less and 16 x 4 (4 x 16 when the lanes run down the
    columns: rolled views) once the ray lattice is sparser than the texels (hl_build_rm_consts: pixel > 1.25 texels at the grid centre).
    Scheduling only -- both must give the oracle's frame with the oracle's samples, at image sizes that are not multiples of either block."""
    sc = S.make_scene("C1", cubemap="r8")
    sc.width, sc.height = width, height
    if rolled:
        pos = np.asarray(sc.cam_pos, dtype=np.float64)
        f = -pos / np.linalg.norm(pos)
        right = np.cross((0.0, 1.0, 0.0), f); right /= np.linalg.norm(right)
        c2w = np.eye(4)
        c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = np.cross(f, right) * -1.0, right, -f, pos      # camera rolled by 90 degrees about its axis
        sc.cam_to_world, sc.world_to_cam = c2w, np.linalg.inv(c2w)
    dist = float(np.linalg.norm(np.asarray(sc.cam_pos) - np.asarray(sc.grid_center)))
    texels_per_pixel = (2.0 * dist * np.tan(np.radians(60.0) / 2) / height) / (sc.mv_scale / (sc.nv - 2 * sc.border))
    assert (texels_per_pixel > 1.25) == (height < 100)          # the first three cases take the 16 x 4 block, the last the 8 x 8 one
    check(sc, exact=False)


@pytest.mark.parametrize("dims", [(5, 16, 200, 80, 60), (3, 32, 60, 64, 64)])
def test_odd_grids(dims):
    check(S.make_scene("odd", dims=dims))


@pytest.mark.parametrize("steps,soft", [(1, 1), (2, 20), (3, 1), (7, 1000), (128, 5), (200, 20), (1000, 3)])
def test_extreme_step_counts_and_soft_distances(steps, soft):
    """_NumRaymarchStepsPerMV and _SoftDistance are inspector ints (VPR.cs:92-93): one lattice point per metavoxel diagonal up to a thousand (every
    visit then outruns the sample loops' unrolling: the 4 / 2 / 1 remainder paths), a fade that covers one lattice index or every sample of the ray."""
    sc = S.make_scene("steps", dims=(3, 16, 150, 72, 56))
    sc.steps, sc.soft_distance = steps, soft
    for cam in ((1.1, 0.7, -1.6), (0.2, 0.1, 0.3)):                  # outside, and inside the grid (tCamera clamps tEntry)
        D = 0.8 * 3 * sc.mv_scale
        sc.set_camera(tuple(D * x for x in cam))
        check(sc, exact=True)
        check(sc, exact=False)


@pytest.mark.parametrize("case", ["fov_10", "fov_150", "image_1x1", "image_7x3", "opacity_0", "opacity_5", "displacement_0", "ambient_0", "far_grid",
                                  "near_far_tight", "tiny_mv", "huge_mv"])
def test_inspector_extremes(case):
    """The other inspector fields at the ends of their useful ranges (VPR.cs:84-97 and the camera): field of view, image sizes below one wave's
    pixel block, opacityFactor 0 (an empty cloud that still occupies its metavoxels) and 5 (saturates within a voxel), no displacement, black
    ambient, a grid 5 km from the origin (float cancellation in every world-space difference), clip planes that cut the cloud, metavoxels of
    3 cm and of 300 m."""
    sc = S.make_scene(case, dims=(3, 16, 120, 64, 48))
    cam_scale = 1.0
    if case == "fov_10": sc.fov_y_deg = 10.0
    elif case == "fov_150": sc.fov_y_deg = 150.0
    elif case == "image_1x1": sc.width, sc.height = 1, 1
    elif case == "image_7x3": sc.width, sc.height = 7, 3
    elif case == "opacity_0": sc.opacity_factor = 0.0
    elif case == "opacity_5": sc.opacity_factor = 5.0
    elif case == "displacement_0": sc.displacement_scale = 0.0
    elif case == "ambient_0": sc.ambient = (0.0, 0.0, 0.0)
    elif case == "far_grid":
        off = np.array([5000.0, -3000.0, 4000.0], dtype=np.float32)
        sc.grid_center = (np.asarray(sc.grid_center, dtype=np.float32) + off).astype(np.float32)
        m = np.array(sc.psys_local_to_world, dtype=np.float32).copy()
        m[12:15] += off                                                    # column-major: the translation column
        sc.psys_local_to_world = m
    elif case in ("tiny_mv", "huge_mv"):
        k = 0.01 if case == "tiny_mv" else 100.0
        cam_scale = k
        sc.mv_scale = float(sc.mv_scale * k)
        for f in ("position",):
            sc.particles[f] *= np.float32(k)
        sc.particles["size"] *= np.float32(k)
        sc.near, sc.far = sc.near * k, sc.far * k
    D = 0.8 * 3 * sc.mv_scale
    gc = np.asarray(sc.grid_center, dtype=np.float64)
    sc.set_camera(tuple(gc + D * np.array([1.1, 0.6, -1.5])), target=tuple(gc))
    if case == "near_far_tight":
        sc.near, sc.far = 1.2 * D, 2.2 * D                                 # both planes inside the cloud's depth range
    check(sc, exact=True, rgba_tol=1e-3)
    # (opacityFactor 5 scales the near-zero densities at the smoothstep's far edge by 125: the few-ulp deviation of default math there, see check)
    check(sc, exact=False, rgba_tol=1e-3, near_zero_abs=1e-4 if case == "opacity_5" else 0.0)


def test_non_cubic_grid():
    sc = S.make_scene("nc", dims=(6, 16, 300, 96, 64))
    sc.N = (4, 6, 5)
    check(sc)


@pytest.mark.parametrize("border", [0, 2, 3])
def test_border_sizes(border):
    sc = S.make_scene("b", dims=(4, 16, 150, 96, 64), border=border)
    check(sc)


@pytest.mark.parametrize("nv,border", [(16, 7), (9, 4), (32, 15), (64, 31), (5, 2), (3, 1), (2, 0)])
def test_widest_borders_and_smallest_bricks(nv, border):
    """numBorderVoxels up to the largest value vp_create accepts (2 b < nv: an interior of one or two voxels, VPR.cs:139 divides by nv - 2 b) and
    the smallest bricks: the texel lattice of the ray-march then spans one or two voxels, the border index of the light propagation sits next to
    the brick's last slice, and at nv = 2 / 3 every 8 x 8 fill tile is mostly overhang."""
    sc = S.make_scene("wb", dims=(3, nv, 120, 72, 56), border=border)
    check(sc, exact=True)
    check(sc, exact=False)


@pytest.mark.parametrize("axis", ["+z", "-z", "+x", "-x", "+y", "-y", "on_cell_plane", "in_corner"])
def test_axis_aligned_views_of_an_axis_aligned_grid(axis):
    """Light axes = world axes, grid centred on the origin, the camera EXACTLY on a grid axis and an odd image size, so that the centre pixel's ray has
    two direction components that are exactly zero (1 / 0 in the slab tests: IntersectBox relies on min / max dropping the NaN of inf x 0,
    RM.shader:95-118; the cell walk never steps along such an axis), whole pixel columns run inside one cell plane, and ties between cell faces are
    exact.  Also a camera sitting exactly ON a plane between two metavoxel layers, and one exactly in a grid corner."""
    sc = S.make_scene("axis", dims=(4, 16, 300, 65, 49))
    sc.light_to_world = S.to_colmajor16(np.eye(4))
    sc.grid_center = np.zeros(3, dtype=np.float32)
    D = 0.8 * 4 * sc.mv_scale
    up = (0.0, 1.0, 0.0)
    pos = {"+z": (0, 0, D), "-z": (0, 0, -D), "+x": (D, 0, 0), "-x": (-D, 0, 0), "+y": (0, D, 0), "-y": (0, -D, 0),
           "on_cell_plane": (sc.mv_scale * 1.0, 0.0, -D), "in_corner": (2 * sc.mv_scale, 2 * sc.mv_scale, -2 * sc.mv_scale)}[axis]
    if axis in ("+y", "-y"):
        up = (0.0, 0.0, 1.0)
    sc.cam_to_world, sc.world_to_cam = S.look_at_camera(pos, (0.0, 0.0, 0.0), up)
    sc.cam_pos = np.asarray(pos, dtype=np.float32)
    check(sc, exact=True)
    check(sc, exact=False)


@pytest.mark.parametrize("z", [-4.5, -1.5, 1.5, 4.5, -7.5, 7.5])
def test_camera_exactly_between_two_light_axis_layers(z):
    """zBoundary = clamp(RoundToInt((lsCam.z - lsFirstSlice.z) / s), -1, Nz - 1) (VPR.cs:640-649) with the quotient EXACTLY on .5: Mathf.RoundToInt
    rounds half to even, so 0.5 -> 0, 1.5 -> 2, 2.5 -> 2, 3.5 -> 4 (clamped to 3), -0.5 -> -0 = 0, 4.5 -> 4 (clamped); the phase split of the
    slabs -- which metavoxels blend OVER, which UNDER -- follows it.  Single context and a fan-out whose slab cuts meet the boundary."""
    sc = S.make_scene("ztie", dims=(4, 16, 300, 64, 48))
    sc.light_to_world = S.to_colmajor16(np.eye(4))
    sc.grid_center = np.zeros(3, dtype=np.float32)
    pos = (0.4, 0.3, z)
    sc.cam_to_world, sc.world_to_cam = S.look_at_camera(pos, (5.0, 0.6, z + 0.7))
    sc.cam_pos = np.asarray(pos, dtype=np.float32)
    o, g, io, ig = check(sc, exact=True)
    q = (z - (-6.0)) / 3.0
    expect = int(min(max(np.round(q), -1), 3))                           # numpy rounds half to even like Mathf.RoundToInt
    assert g.z_boundary(sc.camera()) == expect == o.z_boundary(sc.camera())
    m = E.Engine(sc.config(devices=[0] * 4, multi_flags=abi.VP_MULTI_PEER_COPY | abi.VP_MULTI_UNIFORM_SLABS))
    m.set_frame(sc.light_to_world, sc.grid_center)
    m.bin(sc.particles, sc.layout, sc.psys_local_to_world)
    m.fill(sc.fill_params())
    single = E.Engine(sc.config())
    single.set_frame(sc.light_to_world, sc.grid_center)
    single.bin(sc.particles, sc.layout, sc.psys_local_to_world)
    single.fill(sc.fill_params())
    assert np.abs(m.raymarch(sc.camera(), sc.raymarch_params()) - single.raymarch(sc.camera(), sc.raymarch_params())).max() <= 2e-5
    m.close(); single.close()


@pytest.mark.parametrize("N", [4, 5])
def test_binning_at_its_ties(N):
    """A.2's corner cases, made exact: particles ON metavoxel boundaries and centres (pIndex an integer or a half-integer: the (int) truncation and
    the even-N half-metavoxel skew of Q1 / Q2), radii of exactly 0.5 s, 1.5 s, 2.5 s (RoundToInt rounds half to even: candidate range +-0, +-2, +-2),
    radius 0.5 s - 1 ulp / + 1 ulp, particles just outside the grid (within and beyond their radius; a negative hi of -0.5 truncates to index 0) and
    far outside.  Lists, not only counts, must be the oracle's."""
    s_mv = 3.0
    sc = S.make_scene("ties", dims=(N, 16, 8, 64, 48))
    sc.light_to_world = S.to_colmajor16(np.eye(4))
    sc.grid_center = np.zeros(3, dtype=np.float32)
    pts, sizes = [], []
    half = np.float32(0.5 * s_mv)
    radii = [half, np.float32(1.5 * s_mv), np.float32(2.5 * s_mv), np.nextafter(half, np.float32(0)), np.nextafter(half, np.float32(10)), np.float32(0.3 * s_mv)]
    coords = [k * 0.5 * s_mv for k in range(-N - 2, N + 3)]                # every half metavoxel from beyond one end of the grid to beyond the other
    rng = np.random.default_rng(N)
    for x in coords:
        for r in radii:
            y, z = (float(rng.choice(coords)) for _ in range(2))
            pts.append((x, y, z)); sizes.append(2 * r)
            pts.append((y, z, x)); sizes.append(2 * r)
    pts += [(50.0, 0.0, 0.0), (-50.0, 50.0, 50.0), (0.0, 0.0, -(N / 2 + 0.5) * s_mv - 0.4 * s_mv)]
    sizes += [2.0, 2.0, 1.0 * s_mv]
    P = len(pts)
    parts = np.zeros(P, dtype=S.PARTICLE_DTYPE)
    parts["position"] = np.asarray(pts, dtype=np.float32)
    parts["size"] = np.asarray(sizes, dtype=np.float32)
    parts["rotation"] = rng.uniform(0, 360, P).astype(np.float32)
    parts["lifetime"], parts["startLifetime"] = 3.0, 6.0
    sc.particles = parts
    o, g, io, ig = check(sc, exact=True)
    cnt = o.bin_counts()
    assert cnt.sum() > P                                               # the big ones cover many metavoxels
    for zz in range(N):
        for yy in range(N):
            for xx in range(N):
                assert np.array_equal(g.bin_list(xx, yy, zz), o.bin_list(xx, yy, zz)), (xx, yy, zz)


@pytest.mark.parametrize("cubemap", ["f32", "r8", "r8_zero_texels_D1", "f32_zero_texels_D1"])
def test_cube_map_face_ties(cubemap):
    """Directions with |x| == |y| EXACTLY (the cube map's edges; App. B.5 / DESIGN 4.5: v_cubeid's tie rule z before y before x is the spec's): light
    axes = world axes, unrotated particles at the grid centre and at points symmetric in x and y, so that every voxel on a diagonal of such a
    particle looks along a face edge -- whole planes of ties instead of the measure-zero set a random scene offers."""
    sc = S.make_scene("ties", dims=(2, 16, 6, 64, 48), cubemap=cubemap[:3].rstrip("_"))
    if cubemap.endswith("_D1"):
        # displacement scale exactly 1 over a map with many exact zeros: net displacement 0 makes the reference's smoothstep jump (x / +0), so on
        # the tie planes -- in-face coordinates that are exact integers or half-integers -- a bilinear weight of exactly 0 decides a voxel's density
        rng = np.random.default_rng(7)
        z = rng.random(sc.cubemap.shape) < 0.4
        sc.cubemap = np.where(z, 0, sc.cubemap).astype(sc.cubemap.dtype)
        sc.displacement_scale = 1.0
    sc.light_to_world = S.to_colmajor16(np.eye(4))
    sc.grid_center = np.zeros(3, dtype=np.float32)
    sc.psys_local_to_world = S.to_colmajor16(np.eye(4))
    pts = [(0.0, 0.0, 0.0), (1.5, 1.5, 0.3), (-1.5, -1.5, -0.6), (1.5, -1.5, 1.0), (0.75, 0.75, -1.5), (0.0, 0.0, 1.5)]
    parts = np.zeros(len(pts), dtype=S.PARTICLE_DTYPE)
    parts["position"] = np.asarray(pts, dtype=np.float32)
    parts["size"] = np.asarray([4.0, 2.0, 2.0, 2.0, 1.0, 2.0], dtype=np.float32)      # powers of two: the divisions by size are exact
    parts["rotation"] = 0.0
    parts["lifetime"], parts["startLifetime"] = 3.0, 6.0
    sc.particles = parts
    check(sc, exact=True)
    # default math at D = 1: net displacement near 0 amplifies the reciprocal's last bit in almost-zero densities (a few fp16 ulp, < 1e-6 absolute:
    # the sweep's rule); what must NOT happen is a voxel on the other side of the jump -- that would be a whole opacityFactor, 4e-2
    check(sc, exact=False, near_zero_abs=1e-5 if cubemap.endswith("_D1") else 0.0)


def test_nv64_extension():
    """64^3-voxel bricks (beyond the reference's NUM_VOXELS 32 cap, Q21): two 32-slice register chunks."""
    sc = S.make_scene("n64", dims=(3, 64, 40, 96, 64))
    check(sc)


def test_many_particles_in_one_metavoxel():
    sc = S.make_scene("dense", dims=(4, 16, 700, 64, 48))
    sc.particles["position"] *= 0.15                       # pile them up: several hundred per MV
    o, g, _, _ = check(sc)
    assert g.stats()["max_pairs_per_mv"] > 256


def test_more_particles_in_one_metavoxel_than_the_lds_sort_holds():
    """> 4096 particles in one MV: the list still comes out in ascending particle index (the reference's summation order), so
    the bricks stay bit-identical to the oracle."""
    sc = S.make_scene("pile", dims=(1, 16, 6000, 48, 32))
    sc.particles["position"] *= 0.02
    o, g, _, _ = check(sc)
    assert g.stats()["max_pairs_per_mv"] > 4096


def test_light_depth_map_and_scene_depth():
    sc = S.make_scene("T0")
    nv, N = sc.nv, sc.N[0]
    rng = np.random.default_rng(3)
    d = np.full((N * nv, N * nv), 1.0, dtype=np.float32)
    d[: N * nv // 2] = (200.0 - 0.3 + rng.uniform(-3, 3, (N * nv // 2, N * nv))).astype(np.float32) / (1000.0 - 0.3)
    sc.light_depth_map = d
    sc.scene_depth = np.full((sc.height, sc.width), 1e9, dtype=np.float32)
    sc.scene_depth[:, : sc.width // 2] = 10.0
    o, g, io, ig = check(sc)
    assert (ig[:, : sc.width // 3, 3].mean()) < ig[:, sc.width // 2 + 8:, 3].mean()


def test_default_fast_mode_on_option_branches():
    sc = S.make_scene("x", dims=(5, 32, 200, 96, 64), fade=1)
    check(sc, exact=False)


def test_call_order_and_argument_errors():
    sc = S.make_scene("T0")
    g = E.Engine(sc.config())
    with pytest.raises(E.VpfxError) as ei:
        g.bin(sc.particles, sc.layout, sc.psys_local_to_world)          # before set_frame
    assert ei.value.code == abi.VP_ERR_STATE
    g.set_frame(sc.light_to_world, sc.grid_center)
    with pytest.raises(E.VpfxError) as ei:
        g.fill(sc.fill_params())                                        # before bin
    assert ei.value.code == abi.VP_ERR_STATE
    g.bin(sc.particles, sc.layout, sc.psys_local_to_world)
    with pytest.raises(E.VpfxError) as ei:
        g.raymarch(sc.camera(), sc.raymarch_params())                   # before fill
    assert ei.value.code == abi.VP_ERR_STATE
    fp = sc.fill_params()
    fp.cubemap = None
    with pytest.raises(E.VpfxError) as ei:
        g.fill(fp)                                                      # no cubemap resident yet
    assert ei.value.code == abi.VP_ERR_BAD_ARG
    lay = S.particle_layout()
    lay.off_size = 200                                                  # outside the 84-byte record
    with pytest.raises(E.VpfxError) as ei:
        g.bin(sc.particles, lay, sc.psys_local_to_world)
    assert ei.value.code == abi.VP_ERR_BAD_ARG
    with pytest.raises(E.VpfxError):
        g.read_brick(0, 0, 0)                                           # not filled


def test_refill_and_rebin_are_idempotent():
    sc = S.make_scene("T0")
    g = E.Engine(sc.config())
    g.set_frame(sc.light_to_world, sc.grid_center)
    imgs = []
    for _ in range(3):
        g.bin(sc.particles, sc.layout, sc.psys_local_to_world)
        g.fill(sc.fill_params())
        imgs.append(g.raymarch(sc.camera(), sc.raymarch_params()))
    np.testing.assert_array_equal(imgs[0], imgs[1])
    np.testing.assert_array_equal(imgs[0], imgs[2])


@pytest.mark.parametrize("N", [4, 12])
def test_a_cloud_that_outgrows_the_pair_pool_between_frames(N):
    """vp_bin fills the lists into the pool as allocated while the host still waits for the totals (launch_bin); a frame whose cloud
    does not fit must be re-launched after the reallocation, and a smaller cloud afterwards must not see the larger one's lists.  N = 4:
    the one-workgroup scan of small grids (lists sorted ahead of the wait as well); N = 12: the tiled scan."""
    scenes = [S.make_scene("g", dims=(N, 16, P, 48, 40), seed=70 + i) for i, P in enumerate((12, 2500, 40, 9000, 40))]
    g = E.Engine(scenes[0].config(), exact=True, early_out=False)
    for sc in scenes:
        o = O.Oracle(sc.config())
        for x in (o, g):
            x.set_frame(sc.light_to_world, sc.grid_center)
            x.bin(sc.particles, sc.layout, sc.psys_local_to_world)
            x.fill(sc.fill_params())
        co = o.bin_counts()
        np.testing.assert_array_equal(co, g.bin_counts())
        assert o.stats()["pairs"] == g.stats()["pairs"] and o.stats()["max_pairs_per_mv"] == g.stats()["max_pairs_per_mv"]
        for zz, yy, xx in list(zip(*np.nonzero(co)))[::3]:
            assert np.array_equal(o.read_brick(xx, yy, zz).view(np.uint16), g.read_brick(xx, yy, zz).view(np.uint16)), (xx, yy, zz)
        np.testing.assert_allclose(g.read_lightmap(), o.read_lightmap(), rtol=1e-5, atol=1e-9)
        assert np.abs(o.raymarch(sc.camera(), sc.raymarch_params()) - g.raymarch(sc.camera(), sc.raymarch_params())).max() <= 1e-3
        o.close()


def test_composite_kernel():
    sc = S.make_scene("T0")
    g = E.Engine(sc.config())
    rng = np.random.default_rng(9)
    a = rng.random((sc.height, sc.width, 1)).astype(np.float32)
    p = np.concatenate([rng.random((sc.height, sc.width, 3)).astype(np.float32) * a, a], -1)
    scene_rgba = rng.random((sc.height, sc.width, 4)).astype(np.float32)
    dp, ds = torch.from_numpy(p).cuda(), torch.from_numpy(scene_rgba).cuda()
    g.composite_device(dp.data_ptr(), ds.data_ptr())
    torch.cuda.synchronize()
    np.testing.assert_allclose(ds.cpu().numpy(), O.composite(p, scene_rgba), atol=1e-6)


def test_manager_mirrors_reference_frame_driver():
    sc = S.make_scene("T0")
    m = MetavoxelManager(sc.N[0], sc.N[1], sc.N[2], sc.mv_scale, sc.nv, sc.border, sc.width, sc.height)
    m.Start()
    m.SetLight(sc.light_to_world)
    m.SetGridCenter(sc.grid_center)
    m.SetDisplacementTexture(sc.cubemap)
    m.updateInterval = 2
    cam = sc.camera()
    img0 = m.OnPostRender(0, sc.particles, sc.layout, cam).copy()      # frame 0: bin + fill + march
    covered0 = m.numMetavoxelsCovered
    moved = sc.particles.copy()
    moved["position"] += 0.5
    img1 = m.OnPostRender(1, moved, sc.layout, cam)                    # frame 1: march only (updateInterval gating, VPR.cs:186)
    np.testing.assert_array_equal(img0, img1)
    img2 = m.OnPostRender(2, moved, sc.layout, cam)                    # frame 2: refilled with the moved particles
    assert np.abs(img2 - img0).max() > 1e-3
    o = O.Oracle(sc.config())
    o.set_frame(sc.light_to_world, sc.grid_center)
    o.bin(sc.particles, sc.layout, sc.psys_local_to_world)
    o.fill(sc.fill_params())
    assert np.abs(o.raymarch(cam, sc.raymarch_params()) - img0).max() <= 1e-3
    assert covered0 == o.stats()["occupied_mv"]
    scene_rgba = np.full((sc.height, sc.width, 4), 0.25, dtype=np.float32)
    m.OnPostRender(3, moved, sc.layout, cam, mainSceneRT=scene_rgba)
    np.testing.assert_allclose(scene_rgba, O.composite(m.particlesRT, np.full_like(scene_rgba, 0.25)), atol=1e-6)


@pytest.mark.parametrize("pos", [(0.9, 19.0, 0.6), (1.0, -19.0, 0.4), (19.2, 1.9, 1.0), (-19.2, -1.0, 2.0), (2.0, 1.0, 19.2), (0.4, 0.2, -2.0)])
def test_camera_positions_all_around_the_grid(pos):
    """Cameras on the light side, below, on both flanks (rays running ALONG the slabs: many cells per slab, both blend phases),
    behind, and inside the cloud: the front-to-back compositing and the streamed cell walk must give the oracle's image, and
    with the early-out off the oracle's sample count."""
    sc = S.make_scene("C1")
    sc.set_camera(pos)
    o, g = both(sc)
    cam, rp = sc.camera(), sc.raymarch_params()
    io, ig = o.raymarch(cam, rp), g.raymarch(cam, rp)
    assert o.stats()["samples"] > 1e6 and np.abs(io - ig).max() <= 1e-3
    assert g.stats()["samples"] <= o.stats()["samples"]
    g2 = E.Engine(sc.config(), early_out=False)
    g2.set_frame(sc.light_to_world, sc.grid_center)
    g2.bin(sc.particles, sc.layout, sc.psys_local_to_world)
    g2.fill(sc.fill_params())
    assert np.abs(g2.raymarch(cam, rp) - io).max() <= 1e-3
    assert o.stats()["samples"] == g2.stats()["samples"]


def test_config2_full_parity():
    """BASELINE config 2: 16^3 x 32^3, 10k particles, 1280x720 -- full per-pixel comparison."""
    sc = S.make_scene("C2")
    o, g = both(sc)
    io, ig = o.raymarch(sc.camera(), sc.raymarch_params()), g.raymarch(sc.camera(), sc.raymarch_params())
    assert np.abs(io - ig).max() <= 1e-3
    np.testing.assert_array_equal(o.bin_counts(), g.bin_counts())
    np.testing.assert_allclose(g.read_lightmap(), o.read_lightmap(), rtol=1e-5, atol=1e-9)
    assert g.stats()["samples"] <= o.stats()["samples"]               # saturation early-out skips exact no-ops only
    st = g.stats()
    assert st["occupied_mv"] == 1785 and st["pairs"] == 48728 and st["max_pairs_per_mv"] == 70


def test_config3_full_parity_and_properties():
    """BASELINE config 3 (the benchmark workload) against the oracle on the GPU box's host cores, plus the
    size-independent properties used for larger configs."""
    sc = S.make_scene("C3")
    g = E.Engine(sc.config(), early_out=False)
    g.set_frame(sc.light_to_world, sc.grid_center)
    g.bin(sc.particles, sc.layout, sc.psys_local_to_world)
    g.fill(sc.fill_params())
    ig = g.raymarch(sc.camera(), sc.raymarch_params())
    st = g.stats()
    assert st["occupied_mv"] == 11325 and st["pairs"] == 480441 and st["max_pairs_per_mv"] == 83
    lm = g.read_lightmap()
    assert lm.min() >= 0.0 and lm.max() <= 1.0 and np.isfinite(ig).all()
    assert ig[..., 3].min() >= 0.0 and ig[..., 3].max() <= 1.0 + 1e-6
    # determinism: a second fill + march of the same inputs is bit-identical (sorted lists => fixed summation order)
    g.fill(sc.fill_params())
    np.testing.assert_array_equal(g.read_lightmap(), lm)
    np.testing.assert_array_equal(g.raymarch(sc.camera(), sc.raymarch_params()), ig)
    o = O.Oracle(sc.config())
    o.set_frame(sc.light_to_world, sc.grid_center)
    o.bin(sc.particles, sc.layout, sc.psys_local_to_world)
    o.fill(sc.fill_params())
    io = o.raymarch(sc.camera(), sc.raymarch_params())
    np.testing.assert_array_equal(o.bin_counts(), g.bin_counts())
    np.testing.assert_allclose(lm, o.read_lightmap(), rtol=1e-5, atol=1e-9)
    assert np.abs(io - ig).max() <= 1e-3
    assert o.stats()["samples"] == st["samples"]               # all 714 M lattice samples of the oracle, none more


def test_occluder_boxes_produce_both_depth_inputs_on_the_gpu():
    sc = S.make_scene("T0")
    L = np.asarray(sc.light_to_world, dtype=np.float64).reshape(4, 4).T
    R = L[:3, :3]
    boxes = [S.make_box(R[:, 2] * 2.0 + R[:, 0] * 3.0, (3.0, 4.0, 0.25), R.T),
             S.make_box((-2.0, 1.0, -1.0), (1.0, 2.0, 1.5), S.quat_to_matrix((0.2, -0.1, 0.3, 0.927)).T),
             S.make_box((0.0, -7.0, 0.0), (40.0, 0.5, 40.0))]
    o, g = O.Oracle(sc.config()), E.Engine(sc.config(), exact=True, early_out=False)
    for x in (o, g):
        x.set_frame(sc.light_to_world, sc.grid_center)
        x.set_occluders(boxes)
    do, dg = o.render_light_depth(), g.render_light_depth()
    assert (do < 1).mean() > 0.2
    np.testing.assert_allclose(dg, do, rtol=0, atol=2e-7)
    so, sg = o.render_scene_depth(sc.camera()), g.render_scene_depth(sc.camera())
    assert ((so < 1e30) == (sg < 1e30)).mean() > 0.9995
    both_hit = (so < 1e30) & (sg < 1e30)
    np.testing.assert_allclose(sg[both_hit], so[both_hit], rtol=1e-5)
    for x in (o, g):
        x.bin(sc.particles, sc.layout, sc.psys_local_to_world)
        x.fill(sc.fill_params())
    np.testing.assert_allclose(g.read_lightmap(), o.read_lightmap(), rtol=1e-5, atol=1e-9)
    io, ig = o.raymarch(sc.camera(), sc.raymarch_params()), g.raymarch(sc.camera(), sc.raymarch_params())
    # a pixel whose box-edge depth differs in the last ulp may flip a whole-MV rejection; allow a handful of such pixels
    bad = (np.abs(io - ig).max(axis=-1) > 1e-3).sum()
    assert bad <= 3, bad
    assert ig[..., 3].mean() < 0.9 * E_free_alpha(sc)


def test_occluder_cylinders_and_ellipsoids_produce_both_depth_inputs_on_the_gpu():
    """ABI 6 (VERDICT r5 missing #1): the reference scene's cylinders (scene:1755,5462,8382,8623) -- and spheres -- as analytic solids beside
    boxes, through vp_set_occluders2: both depth inputs vs the oracle (same fp32 operation order: equal up to silhouette flips), then a whole
    frame with them."""
    sc = S.make_scene("T0")
    L = np.asarray(sc.light_to_world, dtype=np.float64).reshape(4, 4).T
    R = L[:3, :3]
    solids = [S.make_solid(abi.VP_OCC_CYLINDER, R[:, 2] * 1.0 + R[:, 0] * 2.5, (1.2, 2.5, 0.8), S.quat_to_matrix((0.3, -0.2, 0.1, 0.927)).T),
              S.make_solid(abi.VP_OCC_CYLINDER, (-2.0, 0.0, -1.0), (0.5, 1.0, 0.5)),                       # Unity's unit cylinder, axis = world y
              S.make_solid(abi.VP_OCC_CYLINDER, (1.0, 2.0, 0.0), (0.7, 3.0, 0.7), R.T),                    # axis = light up: rays PARALLEL to nothing special
              S.make_solid(abi.VP_OCC_CYLINDER, (0.0, 0.0, 3.0), (0.9, 1.5, 0.9), np.roll(R.T, 1, axis=0)),  # axis = light forward: light rays parallel to the axis (qq == 0 branch)
              S.make_solid(abi.VP_OCC_ELLIPSOID, (2.0, -1.0, 1.0), (1.5, 0.6, 1.0), S.quat_to_matrix((0.1, 0.5, -0.2, 0.837)).T),
              S.make_box((0.0, -7.0, 0.0), (40.0, 0.5, 40.0))]
    o, g = O.Oracle(sc.config()), E.Engine(sc.config(), exact=True, early_out=False)
    for x in (o, g):
        x.set_frame(sc.light_to_world, sc.grid_center)
        x.set_occluders(solids)
    do, dg = o.render_light_depth(), g.render_light_depth()
    assert (do < 1).mean() > 0.2
    flips = ((do < 1) != (dg < 1)).sum()
    assert flips <= 2, flips
    both_hit = (do < 1) & (dg < 1)
    np.testing.assert_allclose(dg[both_hit], do[both_hit], rtol=0, atol=2e-7)
    so, sg = o.render_scene_depth(sc.camera()), g.render_scene_depth(sc.camera())
    assert ((so < 1e30) == (sg < 1e30)).mean() > 0.9995
    both_hit = (so < 1e30) & (sg < 1e30)
    np.testing.assert_allclose(sg[both_hit], so[both_hit], rtol=1e-5)
    for x in (o, g):
        x.bin(sc.particles, sc.layout, sc.psys_local_to_world)
        x.fill(sc.fill_params())
    lo, lg = o.read_lightmap(), g.read_lightmap()
    close = np.isclose(lg, lo, rtol=1e-5, atol=1e-9)
    assert (~close).sum() <= 2, (~close).sum()                # a silhouette texel of the depth map may flip a column's shadow index
    io, ig = o.raymarch(sc.camera(), sc.raymarch_params()), g.raymarch(sc.camera(), sc.raymarch_params())
    bad = (np.abs(io - ig).max(axis=-1) > 1e-3).sum()
    assert bad <= 3, bad
    # typed boxes == vp_obb boxes, bit for bit; bad solids are refused
    bx = S.make_box((-2.0, 1.0, -1.0), (1.0, 2.0, 1.5), S.quat_to_matrix((0.2, -0.1, 0.3, 0.927)).T)
    g.set_occluders([bx])
    d_box = g.render_light_depth()
    g.set_occluders([S.make_solid(abi.VP_OCC_BOX, (-2.0, 1.0, -1.0), (1.0, 2.0, 1.5), S.quat_to_matrix((0.2, -0.1, 0.3, 0.927)).T)])
    np.testing.assert_array_equal(g.render_light_depth(), d_box)
    with pytest.raises(Exception):
        g.set_occluders([S.make_solid(9, (0, 0, 0), (1, 1, 1))])
    with pytest.raises(Exception):
        g.set_occluders([S.make_solid(abi.VP_OCC_ELLIPSOID, (0, 0, 0), (1, -1, 1))])
    np.testing.assert_array_equal(g.render_light_depth(), d_box)          # a refused call leaves the previous set in place


def test_depth_maps_rendered_from_the_solids_are_reused_only_while_they_are_valid():
    """Round 6: the eye depth / light depth map rendered from the solids stay in the context while camera, frame and solids are what they were rendered
    for.  Every step of a sequence that changes one of those (or overwrites a buffer with a caller's map) must give, bit for bit, the frame a FRESH
    context gives."""
    sc = S.make_scene("T1")
    L = np.asarray(sc.light_to_world, dtype=np.float64).reshape(4, 4).T
    A = [S.make_solid(abi.VP_OCC_CYLINDER, (0.5, 0.0, -2.0), (1.0, 4.0, 1.0)), S.make_box((0.0, -6.0, 0.0), (40.0, 0.5, 40.0))]
    B = [S.make_solid(abi.VP_OCC_ELLIPSOID, (-1.0, 1.0, -3.0), (2.0, 1.0, 1.5)), S.make_box((3.0, 0.0, 0.0), (0.5, 5.0, 5.0))]
    cams = [(-1.5, 0.9, -14.0), (6.0, 3.0, -12.0)]

    def fresh(solids, cam_pos, scene_depth=None, light_pos=None):
        s2 = S.make_scene("T1")
        s2.set_camera(cam_pos)
        if light_pos is not None:
            s2.light_to_world = light_pos
        e = E.Engine(s2.config())
        e.set_frame(s2.light_to_world, s2.grid_center)
        e.set_occluders(solids)
        e.bin(s2.particles, s2.layout, s2.psys_local_to_world)
        e.fill(s2.fill_params())
        rp = s2.raymarch_params()
        if scene_depth is not None:
            rp.scene_depth = scene_depth.ctypes.data_as(abi.c_float_p)
        img, lm = e.raymarch(s2.camera(), rp), e.read_lightmap()
        e.close()
        return img, lm

    g = E.Engine(sc.config())
    g.set_frame(sc.light_to_world, sc.grid_center)
    g.set_occluders(A)
    g.bin(sc.particles, sc.layout, sc.psys_local_to_world)

    def frame(cam_pos, scene_depth=None):
        sc.set_camera(cam_pos)
        g.fill(sc.fill_params())
        rp = sc.raymarch_params()
        if scene_depth is not None:
            rp.scene_depth = scene_depth.ctypes.data_as(abi.c_float_p)
        return g.raymarch(sc.camera(), rp), g.read_lightmap()

    def same(a, b):
        np.testing.assert_array_equal(a[0], b[0]); np.testing.assert_array_equal(a[1], b[1])
    ref = fresh(A, cams[0])
    assert ref[0][..., 3].max() > 0.05 and (ref[1] < 1).any()
    same(frame(cams[0]), ref)
    same(frame(cams[0]), ref)                                   # both maps reused
    same(frame(cams[1]), fresh(A, cams[1]))                      # the camera moved: eye depth rendered again
    g.set_occluders(B)
    same(frame(cams[1]), fresh(B, cams[1]))                      # the solids changed: both maps rendered again
    wall = np.full((sc.height, sc.width), 3.0e38, dtype=np.float32); wall[:, : sc.width // 2] = 9.0
    same(frame(cams[1], wall), fresh(B, cams[1], wall))          # a caller's depth buffer overwrites the rendered one ...
    same(frame(cams[1]), fresh(B, cams[1]))                      # ... and is not mistaken for it afterwards
    # the light turns: the light depth map belongs to the frame it was rendered for
    L2 = S.to_colmajor16(S.trs((0.0, 0.0, -44.34), S.quat_to_matrix((0.30, 0.10, 0.0, 0.9487))))
    sc.light_to_world = L2
    g.set_frame(sc.light_to_world, sc.grid_center)
    g.bin(sc.particles, sc.layout, sc.psys_local_to_world)
    same(frame(cams[1]), fresh(B, cams[1], light_pos=L2))
    g.set_occluders([])
    same(frame(cams[1]), fresh([], cams[1], light_pos=L2))       # no solids: no occlusion (the stale buffers are not consulted)
    g.close()


def E_free_alpha(sc):
    g = E.Engine(sc.config())
    g.set_frame(sc.light_to_world, sc.grid_center)
    g.bin(sc.particles, sc.layout, sc.psys_local_to_world)
    g.fill(sc.fill_params())
    return g.raymarch(sc.camera(), sc.raymarch_params())[..., 3].mean()


@pytest.mark.parametrize("flag,tol", [(abi.VP_RM_QUANTIZE_UNORM8, 1.01 / 255), (abi.VP_RM_SHOW_NUM_SAMPLES, 1e-6), (abi.VP_RM_SHOW_BLEND_FUNC, 1e-6),
                                      (abi.VP_RM_QUANTIZE_UNORM8 | abi.VP_RM_SHOW_NUM_SAMPLES, 1.01 / 255)])
def test_render_target_emulation_and_debug_views(flag, tol):
    """8-bit particlesRT emulation (quirk Q19) and the two shader debug views, GPU vs oracle."""
    sc = S.make_scene("T0")
    sc.set_camera((1.5, 14.0, 1.0))                      # OVER and UNDER phases both present
    o, g = both(sc, early_out=False)
    rp = sc.raymarch_params()
    rp.flags = flag
    io, ig = o.raymarch(sc.camera(), rp), g.raymarch(sc.camera(), rp)
    d = np.abs(io - ig)
    if flag & abi.VP_RM_QUANTIZE_UNORM8:
        # both are on the 1/255 lattice; a blend landing within float noise of a rounding boundary may differ by one step
        np.testing.assert_allclose(io * 255, np.rint(io * 255), atol=1e-3)
        np.testing.assert_allclose(ig * 255, np.rint(ig * 255), atol=1e-3)
        assert d.max() <= tol and (d > 1e-6).mean() < 2e-3
    else:
        assert d.max() <= tol
    if flag == abi.VP_RM_SHOW_BLEND_FUNC:
        cols = {tuple(np.round(c, 3)) for c in ig.reshape(-1, 4)[::7]}
        assert (0.0, 0.0, 0.0, 0.0) in cols and len(cols) >= 2


def test_demo_scene_sequence_with_emitter_and_occluders():
    """The reference's own scene (10^3 x 32^3 grid, emitter parameters, camera, light, and its eight Default-layer meshes: ground / back /
    two cubes / four cylinders) driven frame by frame through MetavoxelManager.OnPostRender; the frame after each refill is checked against the oracle."""
    sc, em, boxes = S.make_demo_scene(width=256, height=192)
    m = MetavoxelManager(10, 10, 10, 3.0, 32, 1, sc.width, sc.height)
    m.Start()
    m.SetLight(sc.light_to_world)
    m.SetGridCenter(sc.grid_center)
    m.psysLocalToWorld = sc.psys_local_to_world
    m.SetDisplacementTexture(sc.cubemap)
    m.SetOccluders(boxes)
    cam = sc.camera()
    checked = 0
    for frame in range(6):
        em.step(1.0 / 30.0)
        parts = em.particles()
        assert 50 <= len(parts) <= 60
        img = m.OnPostRender(frame, parts, sc.layout, cam)
        if frame % m.updateInterval == 0 and frame in (0, 4):
            o = O.Oracle(sc.config())
            o.set_frame(sc.light_to_world, sc.grid_center)
            o.set_occluders(boxes)
            o.bin(parts, sc.layout, sc.psys_local_to_world)
            o.fill(sc.fill_params())
            ref = o.raymarch(cam, sc.raymarch_params())
            assert m.numMetavoxelsCovered == o.stats()["occupied_mv"] > 0
            bad = (np.abs(ref - img).max(axis=-1) > 1e-3).sum()
            assert bad <= 3, bad
            assert img[..., 3].max() > 0.05
            checked += 1
    assert checked == 2
