"""Parity of the HIP path (through the C ABI) with the CPU oracle.  Run on the GPU box: -m gpu.

The bar: the primary face-id plane must agree with the oracle on every pixel the oracle did not flag as
NaN-dependent (the acceptance bar is >= 99.9 %; these scenes give 100 %), the ray counts (primary,
shadow, bounce) must be identical -- they are sums over every hit of the whole ray tree, so they check
the secondary rays too -- and colours must agree within PIXEL_TOL = 1e-3 per channel (the acceptance
tolerance; observed differences are a few 1e-8, from powf)."""
import ctypes as C
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN, scene_path
from parity_util import FACE_ID_MIN_MATCH, PIXEL_TOL, compare, make_pair, oracle_camera

import raytracer_amd as R
from raytracer_amd import _lib as L
from raytracer_amd import tiles, workloads

pytestmark = pytest.mark.gpu

ALL = ["reflect_test", "reflect_test_2", "shadowmodel", "NthTest", "glass", "toy", "toy_on_plane", "dodgeColorTest", "bunny",
       "textured_quad"]


def assert_parity(r, exact_faces=True):
    assert r["rays_gpu"] == r["rays_oracle"], (r["rays_gpu"], r["rays_oracle"])
    if exact_faces:
        assert r["face_mismatch_pixels_nanfree"] == 0 and r["face_mismatch_pixels"] == 0
    assert r["face_match_nanfree"] >= FACE_ID_MIN_MATCH and r["face_match"] >= FACE_ID_MIN_MATCH
    assert r["bad_pixels_nanfree"] == 0
    assert r["max_err_nanfree_matching"] <= PIXEL_TOL and r["max_err_matching"] <= PIXEL_TOL


@pytest.mark.parametrize("name", ALL)
def test_default_settings(name):
    """Reference defaults: one light at (-1,1,1), 1 hard + 12 soft samples, 5 reflections, 64x64 .. 160x160."""
    size = 160 if name in ("toy", "bunny") else 96
    fly, orc = make_pair(name, width=size, height=size)
    assert_parity(compare(fly, orc, size, size))


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "*_256.npz"))), ids=lambda p: os.path.basename(p)[:-4])
def test_against_committed_golden_vectors(path):
    g = np.load(path)
    name = os.path.basename(path)[:-8]
    fly = R.Flyscene(model=scene_path(name), output="").initialize(256, 256)
    got = fly.render()
    clean = g["nan_mask"] == 0
    assert np.array_equal(got["face_id"][clean], g["face_id"][clean])
    assert np.array_equal(got["face_id"], g["face_id_nanfree"])
    assert np.abs(got["rgb"] - g["rgb_nanfree"]).max() <= PIXEL_TOL
    assert np.abs(got["rgb"] - g["rgb"])[clean].max() <= PIXEL_TOL
    st = got["stats"]
    assert [st.rays_primary, st.rays_shadow, st.rays_bounce] == g["rays_nanfree"].tolist()


def test_texture_branch_from_arrays_matches_obj_path_and_oracle():
    """shadeOffFace's texture branch (flyscene.cpp:516-542): texel picked by area-weighted, un-normalised
    texture coordinates; faces without `vt` carry (-1,-1) and land on the clamped first texel."""
    from raytracer_amd.flyscene import texture_arrays

    fly, orc = make_pair("textured_quad", width=128, height=128)
    r = compare(fly, orc, 128, 128)
    assert_parity(r)
    # texels of both images must show up in the frame: many distinct colours on the two textured faces
    quad = np.isin(r["got"]["face_id"], (0, 1))
    assert len(np.unique(r["got"]["rgb"][quad].round(4), axis=0)) > 50
    # the same scene handed over as arrays + textures renders the same image bit for bit
    a = fly.arrays()
    t = texture_arrays(fly)
    mats = [dict(ka=m[0:3], kd=m[3:6], ks=m[6:9], shininess=m[9]) for m in a["materials"].reshape(-1, 10)]
    again = R.Flyscene(arrays=(a["raw_vertices"].reshape(-1, 3), a["raw_normals"].reshape(-1, 3), a["face_vertex_ids"].reshape(-1, 3),
                               a["face_material_ids"], mats), textures=t, output="").initialize(128, 128)
    assert np.array_equal(again.render()["rgb"].view(np.uint32), r["got"]["rgb"].view(np.uint32))
    # without the textures the material colour is used and the frame differs
    plain = R.Flyscene(arrays=(a["raw_vertices"].reshape(-1, 3), a["raw_normals"].reshape(-1, 3), a["face_vertex_ids"].reshape(-1, 3),
                               a["face_material_ids"], mats), output="").initialize(128, 128)
    assert np.abs(plain.render()["rgb"] - r["got"]["rgb"])[quad].max() > 0.05


def test_single_stream_flag_and_many_live_scenes():
    """Every frame runs on its caller's stream alone (the library owns no streams), so several reflective scenes alive
    at once add none, and RTAMD_FLAG_SINGLE_STREAM, kept for callers that set it, gives the same image."""
    flies = [make_pair("reflect_test", width=64, height=64)[0] for _ in range(4)]
    frames = [f.render()["rgb"] for f in flies]
    for fr in frames[1:]:
        assert np.array_equal(fr.view(np.uint32), frames[0].view(np.uint32))
    flies[0].set_params(flags=L.FLAG_SINGLE_STREAM)
    assert np.array_equal(flies[0].render()["rgb"].view(np.uint32), frames[0].view(np.uint32))
    assert flies[0].render()["stats"].rays_bounce > 0


@pytest.mark.parametrize("name", ["NthTest", "toy_on_plane", "bunny", "textured_quad"])
def test_shadow_grids_and_tree_walk_agree(name):
    """Shadow rays take per-sample candidate grids by default (all rays of a sample are parallel); forcing them
    through the BVH must give the same frame bit for bit: both run the same exact face test."""
    kw = dict(model=scene_path(name), output="", lights=[(-1, 1, 1), (0.0, 1.0, 1.5), (1.5, 1.0, 1e-16)])
    a = R.Flyscene(**kw).initialize(96, 96)
    b = R.Flyscene(shadow_rays=L.SHADOW_RAYS_TREE, **kw).initialize(96, 96)
    assert a.info().shadow_grid_entries > 0 and b.info().shadow_grid_entries == 0
    ra, rb = a.render(), b.render()
    assert np.array_equal(ra["rgb"].view(np.uint32), rb["rgb"].view(np.uint32))
    assert np.array_equal(ra["face_id"], rb["face_id"])
    sa, sb = ra["stats"], rb["stats"]
    assert (sa.rays_shadow, sa.rays_bounce, sa.hits) == (sb.rays_shadow, sb.rays_bounce, sb.hits)
    assert sa.rays_shadow > 0


@pytest.mark.parametrize("name,view", [("NthTest", "default"), ("toy_on_plane", "default"), ("bunny", "default"), ("toy", "inside"),
                                       ("reflect_test", "close"), ("textured_quad", "default")])
def test_screen_bins_and_tree_walk_agree_for_primary_rays(name, view):
    """Primary rays are resolved by binning the projected faces per pass; forcing them through the BVH must give
    the same planes bit for bit -- also with faces beside and behind the eye, SSAA sub-samples, a row partition
    and a camera that has been turned.  The automatic choice (bins unless their lists come out long, learnt from
    the passes before) must give the same frame whichever way it goes, frame after frame."""
    kw = dict(model=scene_path(name), output="")
    a = R.Flyscene(primary_rays=L.PRIMARY_RAYS_BINS, **kw).initialize(120, 88)
    b = R.Flyscene(primary_rays=L.PRIMARY_RAYS_TREE, **kw).initialize(120, 88)
    c = R.Flyscene(**kw).initialize(120, 88)
    for f in (a, b, c):
        if view == "inside":
            f.camera_translate(0.0, 0.0, 38.0)  # steps of 0.05: the eye ends up inside the model's bounding box
            f.camera_rotate(0.3, -0.7)
        elif view == "close":
            f.camera_translate(2.0, -1.0, 24.0)
            f.camera_rotate(-0.2, 0.4)
    for params in (dict(), dict(ssaa_level=2), dict(row_tile=8, num_parts=3, part=1)):
        for f in (a, b, c):
            f.set_params(**params)
        ra, rb = a.render(), b.render()
        assert ra["stats"].primary_path == 1 and rb["stats"].primary_path == 0
        if name == "bunny":  # (scenes of a few large faces keep them all in the list every pixel scans)
            assert ra["stats"].bin_entries > 0 and ra["stats"].bins_nonempty > 0
        assert np.array_equal(ra["face_id"], rb["face_id"]), (name, view, params)
        assert np.array_equal(ra["rgb"].view(np.uint32), rb["rgb"].view(np.uint32)), (name, view, params)
        sa, sb = ra["stats"], rb["stats"]
        assert (sa.rays_primary, sa.rays_shadow, sa.rays_bounce, sa.hits) == (sb.rays_primary, sb.rays_shadow, sb.rays_bounce, sb.hits)
        for again in range(3):  # the second frame already knows what the first frame's binning pass found
            rc = c.render()
            assert np.array_equal(rc["face_id"], rb["face_id"]), (name, view, params, again)
            assert np.array_equal(rc["rgb"].view(np.uint32), rb["rgb"].view(np.uint32)), (name, view, params, again)
        for f in (a, b, c):
            f.set_params(ssaa_level=1, row_tile=16, num_parts=1, part=0)
        if not params:
            assert (ra["face_id"] >= 0).any()


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_projection_margins_under_attack(seed):
    """Faces chosen to strain the screen-space candidate structures, whose boxes are widened by a sixteenth of a pixel only
    (the bound on a projected corner's error is 0.008 pixel at 8K): corners almost in the eye plane (depth 1e-6 .. 1e-3 of
    their distance from the eye), corners that project near the 16384-pixel limit of the edge functions, faces seen
    edge-on, in an image 7680 pixels wide and from an eye close to the geometry.  Face raster and screen bins against the
    tree walk, bit for bit."""
    rng = np.random.default_rng(500 + seed)
    W, H = 7680, 240
    aspect, tan_half = W / H, np.tan(np.radians(30.0))
    eye = np.array([0.0, 0.0, 2.0])

    def on_screen(n, depth_lo=0.5, depth_hi=3.0):
        depth = rng.uniform(depth_lo, depth_hi, n)
        nx, ny = rng.uniform(-1.0, 1.0, n) * aspect * tan_half, rng.uniform(-1.0, 1.0, n) * tan_half
        return np.stack([nx * depth, ny * depth, 2.0 - depth], 1)

    tris = []
    n = 1500
    # (a) one corner almost in the eye plane, far to the side; the other two ordinary
    a0 = on_screen(n)
    a1 = a0 + rng.uniform(-0.3, 0.3, (n, 3))
    reach = rng.uniform(0.2, 3.0, n)
    delta = 10.0 ** rng.uniform(-6.5, -3.0, n)
    ang = rng.uniform(0, 2 * np.pi, n)
    a2 = np.stack([reach * np.cos(ang), reach * np.sin(ang), 2.0 - delta * reach], 1)
    tris.append(np.stack([a0, a1, a2], 1))
    # (b) one corner projecting near +-16384 pixels (the limit up to which the projected edges are used)
    b0, b1 = on_screen(n), on_screen(n)
    depth = rng.uniform(0.5, 2.0, n)
    px = rng.choice([-1.0, 1.0], n) * (16384.0 + rng.uniform(-40.0, 40.0, n))
    nxb = (2.0 * px / W - 1.0) * aspect * tan_half
    b2 = np.stack([nxb * depth, rng.uniform(-0.3, 0.3, n) * depth, 2.0 - depth], 1)
    tris.append(np.stack([b0, b1, b2], 1))
    # (c) edge-on: two corners on (almost) one line of sight
    c0 = on_screen(n)
    c1 = eye + (c0 - eye) * rng.uniform(1.2, 2.5, (n, 1)) + rng.normal(0, 1e-5, (n, 3))
    c2 = c0 + rng.uniform(-0.2, 0.2, (n, 3))
    tris.append(np.stack([c0, c1, c2], 1))
    # (d) filler: faces of a pixel or so, several per pixel column
    d0 = on_screen(6 * n, 1.0, 3.0)
    tris.append(np.stack([d0, d0 + rng.normal(0, 0.002, (6 * n, 3)), d0 + rng.normal(0, 0.002, (6 * n, 3))], 1))
    tri = np.concatenate(tris).astype(np.float32)
    verts = tri.reshape(-1, 3)
    faces = np.arange(len(verts), dtype=np.int32).reshape(-1, 3)
    mats = [dict(ka=[0.1] * 3, kd=[0.3 + 0.2 * m, 0.8 - 0.2 * m, 0.4], ks=[0.0] * 3, shininess=8.0) for m in range(3)]
    arrays = (verts, None, faces, rng.integers(0, 3, len(faces)).astype(np.int32), mats)
    kw = dict(arrays=arrays, lights=[(-1.0, 1.0, 1.0)], soft_points=0, normalize_model=False, output="")
    auto = R.Flyscene(**kw).initialize(W, H)
    bins = R.Flyscene(primary_rays=L.PRIMARY_RAYS_BINS, **kw).initialize(W, H)
    tree = R.Flyscene(primary_rays=L.PRIMARY_RAYS_TREE, shadow_rays=L.SHADOW_RAYS_TREE, **kw).initialize(W, H)
    for params in (dict(), dict(row_tile=16, num_parts=8, part=5)):
        for f in (auto, bins, tree):
            f.set_params(max_reflections=0, **params)
        ra, rb, rt = auto.render(), bins.render(), tree.render()
        assert rt["stats"].primary_path == 0 and rb["stats"].primary_path == 1
        assert (rt["face_id"] >= 0).sum() > 2000
        for r in (ra, rb):
            assert np.array_equal(r["face_id"], rt["face_id"]), params
            assert np.array_equal(r["rgb"].view(np.uint32), rt["rgb"].view(np.uint32)), params
            assert (r["stats"].rays_shadow, r["stats"].hits) == (rt["stats"].rays_shadow, rt["stats"].hits)


def _edge_attack_scene(size, ssaa):
    """Faces for test_screen_space_edge_test_under_attack (arguments of Flyscene)."""

    W, H = size
    rng = np.random.default_rng(900 + W + ssaa)
    aspect, tan_half = W / H, np.tan(np.radians(30.0))
    eye = np.array([0.0, 0.0, 2.0])

    def through(px, py, depth):  # world point on the ray of screen position (px, py) (Camera::screenToWorld, default pose)
        nx = (2.0 * px / W - 1.0) * aspect * tan_half
        ny = (1.0 - 2.0 * py / H) * tan_half
        return np.stack([nx * depth, ny * depth, 2.0 - depth], -1)

    tris = []
    n = 1200
    sub = rng.integers(0, ssaa, (n, 1)) / ssaa  # a sample of one of the passes
    # (a) corners on sample rays, edges through the samples between them (lattice steps with a common divisor)
    x0, y0 = rng.integers(8, W - 40, (n, 1)), rng.integers(8, H - 40, (n, 1))
    step = rng.integers(1, 5, (n, 2)) * rng.integers(2, 5, (n, 1))
    other = rng.integers(-6, 7, (n, 2))
    px = np.concatenate([x0, x0 + step[:, :1], x0 + other[:, :1]], 1) + sub
    py = np.concatenate([y0, y0 + step[:, 1:], y0 + other[:, 1:]], 1) + sub
    depth = rng.uniform(0.6, 3.0, (n, 1)) + rng.uniform(-0.2, 0.2, (n, 3))
    tris.append(through(px, py, depth))
    # (b) slivers: two corners far apart, the third almost on the line between them
    ax, ay = rng.uniform(0, W, (n, 1)), rng.uniform(0, H, (n, 1))
    bx, by = ax + rng.uniform(-120, 120, (n, 1)), ay + rng.uniform(-120, 120, (n, 1))  # (a box beyond 16384 pixels is not this pass's)
    t = rng.uniform(0.1, 0.9, (n, 1))
    off = 10.0 ** rng.uniform(-4.0, -1.0, (n, 1)) * rng.choice([-1.0, 1.0], (n, 1))
    length = np.hypot(bx - ax, by - ay) + 1e-9
    cx, cy = ax + t * (bx - ax) - off * (by - ay) / length, ay + t * (by - ay) + off * (bx - ax) / length
    d = rng.uniform(0.8, 2.5, (n, 1))
    tris.append(through(np.concatenate([ax, bx, cx], 1), np.concatenate([ay, by, cy], 1), d + rng.uniform(-0.05, 0.05, (n, 3))))
    # (c) faces far smaller than a pixel, around sample points
    sx, sy = rng.integers(0, W, (n, 1)) + sub, rng.integers(0, H, (n, 1)) + sub
    r = 10.0 ** rng.uniform(-3.0, -0.5, (n, 1))
    ang = rng.uniform(0, 2 * np.pi, (n, 1)) + np.array([[0.0, 2.1, 4.2]])
    tris.append(through(sx + r * np.cos(ang), sy + r * np.sin(ang), rng.uniform(0.8, 2.5, (n, 1)) + np.zeros((n, 3))))
    # (d) one corner thousands of pixels off the screen, the other two near that side's border (a box of a few thousand pixels:
    # larger ones are not this pass's, they are tested pixel by pixel in the compaction)
    side = rng.choice([-1.0, 1.0], (n, 1))
    gx = np.where(side < 0, rng.uniform(0, 40, (n, 2)), rng.uniform(W - 40, W, (n, 2)))
    gy = rng.uniform(40, H - 40, (n, 1)) + rng.uniform(-20, 20, (n, 2))
    fx = np.where(side < 0, -rng.uniform(2000.0, 15000.0, (n, 1)), W + rng.uniform(2000.0, 15000.0 - W, (n, 1)))
    fy = gy[:, :1] + rng.uniform(-300.0, 300.0, (n, 1))
    tris.append(through(np.concatenate([gx, fx], 1), np.concatenate([gy, fy], 1), rng.uniform(0.8, 2.5, (n, 1)) + rng.uniform(-0.1, 0.1, (n, 3))))
    tri = np.concatenate(tris).astype(np.float32)
    assert np.all(tri[..., 2] < eye[2] - 0.3)
    verts = tri.reshape(-1, 3)
    faces = np.arange(len(verts), dtype=np.int32).reshape(-1, 3)
    mats = [dict(ka=[0.1] * 3, kd=[0.3 + 0.2 * m, 0.8 - 0.2 * m, 0.4], ks=[0.0] * 3, shininess=8.0) for m in range(3)]
    arrays = (verts, None, faces, rng.integers(0, 3, len(faces)).astype(np.int32), mats)
    kw = dict(arrays=arrays, lights=[(-1.0, 1.0, 1.0)], soft_points=0, normalize_model=False, output="")
    return kw


@pytest.mark.parametrize("size,ssaa", [((1920, 1080), 1), ((333, 777), 2), ((640, 480), 3)])
def test_screen_space_edge_test_under_attack(size, ssaa):
    """The face pass drops a (face, pixel) pair when the pixel's sample lies outside the PROJECTED triangle by more than a
    margin, before the exact test sees it.  Faces chosen to sit on that decision: corners exactly on the rays of pixel
    samples with edges through further samples (the exact test decides those pixels by its rounding), slivers a hundred
    pixels long and a hundredth of a pixel wide (orientation of the projected triangle uncertain), faces of a ten-thousandth
    of a pixel, corners thousands of pixels off the screen.  Against the tree walk, bit for bit, whole and as one rank's
    share; every sub-pixel offset of the SSAA passes moves the samples."""
    W, H = size
    kw = _edge_attack_scene(size, ssaa)
    auto = R.Flyscene(**kw).initialize(W, H)
    tree = R.Flyscene(primary_rays=L.PRIMARY_RAYS_TREE, shadow_rays=L.SHADOW_RAYS_TREE, **kw).initialize(W, H)
    for params in (dict(), dict(row_tile=16, num_parts=4, part=1)):
        for f in (auto, tree):
            f.set_params(max_reflections=0, ssaa_level=ssaa, **params)
        ra, rt = auto.render(), tree.render()
        assert rt["stats"].primary_path == 0 and ra["stats"].primary_path == 1
        assert (rt["face_id"] >= 0).sum() > 2000
        assert np.array_equal(ra["face_id"], rt["face_id"]), params
        assert np.array_equal(ra["rgb"].view(np.uint32), rt["rgb"].view(np.uint32)), params
        assert (ra["stats"].rays_shadow, ra["stats"].hits) == (rt["stats"].rays_shadow, rt["stats"].hits)
    if W % 2 == 1 and ssaa == 2:
        # The centre column of the last pass: its rays have a direction whose x is exactly zero and run IN the plane of a
        # side of every box that a face with a corner at x = 0 ends at (the corners of group (a) lie on such rays).  Ray by
        # ray against the oracle (the face ids of a frame are those of its last pass; plane distances that are not a number
        # rejected, as the HIP path defines it: see parity_util.compare).
        from oracle.oracle_py import OracleScene, screen_ray
        orc = OracleScene(arrays=kw["arrays"], lights=[(-1.0, 1.0, 1.0, 1.0, 1.0, 1.0)], soft_points=0, normalize_model=False)
        cam = orc.default_camera(W, H)
        auto.set_params(max_reflections=0, ssaa_level=ssaa, row_tile=16, num_parts=1, part=0)
        ra = auto.render()
        hits = 0
        for y in range(H):
            d = screen_ray(cam, W // 2 + 0.5, y + 0.5)
            assert d[0] == 0.0
            hit, fid, _, _, _ = orc.intersects(list(cam.origin), d, reject_nan=1)
            assert ra["face_id"][y, W // 2] == (fid if hit else -2), y
            hits += int(hit)
        assert hits > 20


@pytest.mark.parametrize("size", [(101, 75), (64, 49)])
def test_rays_with_zero_direction_components(size):
    """Axis-aligned mirrors in front of the default camera and lights on the coordinate axes: the centre row and column of
    an odd-sized image, their reflections and the shadow rays of a light on an axis have directions with components that
    are exactly zero, and the faces end exactly on the planes x = 0 and y = 0 those rays run in (the sides of the tree's
    boxes, the borders of the shadow grids' cells).  Every path against the oracle, and against each other bit for bit."""
    W, H = size
    rng = np.random.default_rng(7 + W)
    tris = []
    # a mirror wall behind (z = -0.6) and a mirror floor (y = -0.5), both split along x = 0
    for x0, x1 in ((-1.2, 0.0), (0.0, 1.2)):
        tris += [[(x0, -0.5, -0.6), (x1, -0.5, -0.6), (x1, 0.9, -0.6)], [(x0, -0.5, -0.6), (x1, 0.9, -0.6), (x0, 0.9, -0.6)]]
        tris += [[(x0, -0.5, -0.6), (x1, -0.5, 1.2), (x1, -0.5, -0.6)], [(x0, -0.5, -0.6), (x0, -0.5, 1.2), (x1, -0.5, 1.2)]]
    n_mirror = len(tris)
    # small dull faces that end exactly at x = 0 or y = 0, in front of the wall and above the floor (seen directly, in the
    # mirrors, and as occluders of the lights on the axes)
    for _ in range(60):
        c = rng.uniform(-0.4, 0.4, 3) + np.array([0.0, 0.1, 0.3])
        t = c + rng.uniform(-0.15, 0.15, (3, 3))
        axis = int(rng.integers(0, 2))
        side = rng.choice([-1.0, 1.0])
        t[:, axis] = side * np.abs(t[:, axis])
        t[int(rng.integers(0, 3)), axis] = 0.0
        if rng.random() < 0.3:
            t[int(rng.integers(0, 3)), axis] = 0.0  # an edge in the plane
        tris.append(t.tolist())
    tri = np.array(tris, np.float32)
    verts = tri.reshape(-1, 3)
    faces = np.arange(len(verts), dtype=np.int32).reshape(-1, 3)
    mats = [dict(ka=[0.05] * 3, kd=[0.2, 0.2, 0.25], ks=[0.9] * 3, shininess=32.0),
            dict(ka=[0.1] * 3, kd=[0.8, 0.5, 0.3], ks=[0.0] * 3, shininess=8.0),
            dict(ka=[0.1] * 3, kd=[0.3, 0.6, 0.8], ks=[0.0] * 3, shininess=8.0)]
    fmat = np.concatenate([np.zeros(n_mirror, np.int32), rng.integers(1, 3, len(faces) - n_mirror).astype(np.int32)])
    arrays = (verts, None, faces, fmat, mats)
    lights = [(0.0, 1.5, 0.0), (0.0, 0.0, 1.5), (1.0, 0.0, 0.0)]
    kw = dict(arrays=arrays, lights=lights, soft_points=0, normalize_model=False, width=W, height=H)
    fly, orc = make_pair(**kw)
    r = compare(fly, orc, W, H, max_reflections=3)
    assert_parity(r)
    assert r["rays_gpu"][2] > 1000  # reflections happen
    base = r["got"]
    # the centre column / row really are such rays
    cam = oracle_camera(orc, fly.getCamera())
    from oracle.oracle_py import screen_ray
    if W % 2 == 1:
        assert screen_ray(cam, W // 2 + 0.5, 3.5)[0] == 0.0
    if H % 2 == 1:
        assert screen_ray(cam, 3.5, H // 2 + 0.5)[1] == 0.0
    for build in (dict(primary_rays=L.PRIMARY_RAYS_TREE, shadow_rays=L.SHADOW_RAYS_TREE), dict(primary_rays=L.PRIMARY_RAYS_BINS),
                  dict(builder=L.BUILDER_DEVICE_LBVH), dict(builder=L.BUILDER_DEVICE_LBVH, primary_rays=L.PRIMARY_RAYS_TREE, shadow_rays=L.SHADOW_RAYS_TREE)):
        other = R.Flyscene(arrays=arrays, lights=lights, soft_points=0, normalize_model=False, output="", **build).initialize(W, H)
        other.set_params(max_reflections=3)
        got = other.render()
        assert np.array_equal(got["face_id"], base["face_id"]), build
        assert np.array_equal(got["rgb"].view(np.uint32), base["rgb"].view(np.uint32)), build
        assert (got["stats"].rays_shadow, got["stats"].rays_bounce, got["stats"].hits) == (base["stats"].rays_shadow, base["stats"].rays_bounce, base["stats"].hits)


def _far_scene(translate, seed, n=3000, size=0.08, ks=0.0):
    """A soup of small faces two units in front of a camera that has been moved by `translate` (Flycamera steps), with its
    light: coordinates as large as the camera's, depths of about two."""
    rng = np.random.default_rng(seed)
    one = (np.zeros((3, 3), np.float32), None, np.array([[0, 1, 2]], np.int32), np.zeros(1, np.int32),
           [dict(ka=[0.1] * 3, kd=[0.5] * 3, ks=[0.0] * 3, shininess=8.0)])
    probe = R.Flyscene(arrays=one, lights=[(0, 0, 0)], soft_points=0, normalize_model=False, output="").initialize(64, 64, upload=False)
    probe.camera_translate(*translate)
    eye = np.array(list(probe.getCamera().origin), np.float64)
    centre = eye + np.array([0.0, 0.0, -2.0])
    c = centre + rng.uniform(-0.8, 0.8, (n, 1, 3)) * np.array([1.6, 0.9, 0.5])
    tri = (c + rng.uniform(-size, size, (n, 3, 3))).astype(np.float32)
    verts = tri.reshape(-1, 3)
    faces = np.arange(3 * n, dtype=np.int32).reshape(n, 3)
    mats = [dict(ka=[0.1] * 3, kd=[0.3 + 0.2 * m, 0.8 - 0.2 * m, 0.4], ks=[ks] * 3, shininess=8.0) for m in range(3)]
    arrays = (verts, None, faces, rng.integers(0, 3, n).astype(np.int32), mats)
    light = tuple((centre + np.array([-1.0, 1.0, 1.0])).tolist())
    return arrays, [light], eye


@pytest.mark.parametrize("translate", [(1200.0, -900.0, 1300.0), (12000.0, -9000.0, 13000.0), (150000.0, 90000.0, -130000.0)])
def test_scene_far_from_the_origin_seen_from_nearby(translate):
    """The exact test computes with the coordinates themselves: with an eye and faces a thousand units from the origin and two
    units apart, its decisions are good to a few units in the last place of 1000, which is a tenth of a pixel and more at 4K.
    The margins of the screen-space structures and the slack of the tree's boxes grow with the largest coordinate
    (FrameConst::coord_slack, DeviceScene::box_slack); with the fixed sixteenth of a pixel the face pass lost 33 pixels of
    this frame.  Face pass and screen bins against the tree walk at 4K, bit for bit, and everything against the oracle on a
    small image with reflections."""
    arrays, lights, eye = _far_scene(translate, 5)
    assert np.abs(eye).max() > 50.0
    kw = dict(arrays=arrays, lights=lights, soft_points=0, normalize_model=False, output="")
    W, H = 3840, 2160
    out = []
    for build in (dict(primary_rays=L.PRIMARY_RAYS_TREE, shadow_rays=L.SHADOW_RAYS_TREE), dict(), dict(primary_rays=L.PRIMARY_RAYS_BINS)):
        f = R.Flyscene(**kw, **build).initialize(W, H)
        f.camera_translate(*translate)
        f.set_params(max_reflections=0)
        out.append(f.render())
        f.close()
    assert out[0]["stats"].primary_path == 0 and out[2]["stats"].primary_path == 1
    # (ten thousand units out the margins are tens of pixels, the faces' boxes pass 16384 pixels and the face pass gives way)
    assert out[1]["stats"].primary_path == (1 if np.abs(eye).max() < 2000.0 else 0)
    assert (out[0]["face_id"] >= 0).sum() > 1000000
    for r in out[1:]:
        assert np.array_equal(r["face_id"], out[0]["face_id"])
        assert np.array_equal(r["rgb"].view(np.uint32), out[0]["rgb"].view(np.uint32))
        assert (r["stats"].rays_shadow, r["stats"].hits) == (out[0]["stats"].rays_shadow, out[0]["stats"].hits)
    arrays, lights, _ = _far_scene(translate, 6, n=400, size=0.25, ks=0.6)
    fly, orc = make_pair(arrays=arrays, lights=lights, soft_points=2, normalize_model=False, width=96, height=54)
    fly.camera_translate(*translate)
    r = compare(fly, orc, 96, 54, max_reflections=2)
    assert_parity(r)
    assert r["rays_gpu"][2] > 500


@pytest.mark.parametrize("size,step,ssaa", [((128, 96), 4, 1), ((97, 131), 3, 1), ((64, 64), 2, 2)])
def test_watertight_mesh_with_corners_and_edges_on_the_sample_rays(size, step, ssaa):
    """A height field whose corners lie on the rays of every step-th pixel sample: those rays pass through a corner shared by
    six faces, the rays of the samples between them run along shared edges and diagonals.  Which face such a ray hits, or
    whether it slips between two, is decided by the last bit of the reference's arithmetic and by its visiting order;
    every path must decide like the oracle, for the primary rays, the shadow rays that start on those corners and edges, and
    the reflections."""
    W, H = size
    rng = np.random.default_rng(31 + W + step)
    aspect, tan_half = W / H, np.tan(np.radians(30.0))
    nx_, ny_ = W // step + 3, H // step + 3
    ix, iy = np.meshgrid(np.arange(nx_) - 1, np.arange(ny_) - 1, indexing="xy")
    px, py = ix * step + (ssaa - 1) / ssaa, iy * step + (ssaa - 1) / ssaa  # samples of the LAST pass (whose face ids a frame reports)
    depth = 2.4 + 0.35 * np.sin(ix * 0.37) * np.cos(iy * 0.29) + rng.uniform(-0.02, 0.02, ix.shape)
    nx = (2.0 * px / W - 1.0) * aspect * tan_half
    ny = (1.0 - 2.0 * py / H) * tan_half
    verts = np.stack([nx * depth, ny * depth, 2.0 - depth], -1).reshape(-1, 3).astype(np.float32)
    idx = (iy + 1) * nx_ + (ix + 1)
    a, b, c, d = idx[:-1, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, 1:].ravel(), idx[1:, :-1].ravel()
    flip = rng.random(a.shape) < 0.5  # the quads' diagonals alternate at random
    t0 = np.where(flip[:, None], np.stack([a, b, d], 1), np.stack([a, b, c], 1))
    t1 = np.where(flip[:, None], np.stack([b, c, d], 1), np.stack([a, c, d], 1))
    faces = np.concatenate([t0, t1]).astype(np.int32)
    mats = [dict(ka=[0.1] * 3, kd=[0.7, 0.4, 0.3], ks=[0.0] * 3, shininess=8.0), dict(ka=[0.05] * 3, kd=[0.3, 0.5, 0.7], ks=[0.5] * 3, shininess=32.0)]
    arrays = (verts, None, faces, rng.integers(0, 2, len(faces)).astype(np.int32), mats)
    lights = [(-1.0, 1.0, 1.0), (0.6, 0.3, 1.8)]
    kw = dict(arrays=arrays, lights=lights, soft_points=2, normalize_model=False, width=W, height=H)
    fly, orc = make_pair(**kw)
    r = compare(fly, orc, W, H, all_faces=True, max_reflections=2, ssaa_level=ssaa)
    # The HIP path accepts a face by the face test alone: it must equal the oracle run in which no box turns a ray away, on
    # every pixel; and the faithful run wherever the reference's own boxes do not decide (a ray through a corner of a leaf's
    # box may be turned away by the reference's rounded box test from a face its face test accepts: a handful of these rays).
    assert r["face_mismatch_pixels_all_faces"] == 0 and r["bad_pixels_all_faces"] == 0
    assert r["rays_gpu"] == r["rays_oracle_all_faces"]
    assert r["face_mismatch_pixels_where_boxes_agree"] == 0 and r["bad_pixels_where_boxes_agree"] == 0
    assert r["box_dependent_pixels"] <= 2e-3 * W * H
    # (the mesh covers the image, but a tenth of these rays slips between two faces: the reference's edge functions turn a point
    # on a shared edge away from both sides when the last bit falls that way, and so does every path here)
    assert 0.8 < (r["got"]["face_id"] >= 0).mean() < 1.0 and r["rays_gpu"][2] > 100
    base = r["got"]
    for build in (dict(primary_rays=L.PRIMARY_RAYS_TREE, shadow_rays=L.SHADOW_RAYS_TREE), dict(primary_rays=L.PRIMARY_RAYS_BINS),
                  dict(builder=L.BUILDER_DEVICE_LBVH), dict(builder=L.BUILDER_DEVICE_LBVH, primary_rays=L.PRIMARY_RAYS_TREE, shadow_rays=L.SHADOW_RAYS_TREE)):
        other = R.Flyscene(arrays=arrays, lights=lights, soft_points=2, normalize_model=False, output="", **build).initialize(W, H)
        other.set_params(max_reflections=2, ssaa_level=ssaa)
        got = other.render()
        assert np.array_equal(got["face_id"], base["face_id"]), build
        assert np.array_equal(got["rgb"].view(np.uint32), base["rgb"].view(np.uint32)), build
        assert (got["stats"].rays_shadow, got["stats"].rays_bounce, got["stats"].hits) == (base["stats"].rays_shadow, base["stats"].rays_bounce, base["stats"].hits)


def test_automatic_choice_between_screen_space_and_tree():
    """Primary rays are resolved face by face in screen space unless the faces are many against the pixels (a projection per
    face then costs more than a tree walk per pixel) or too many faces reach round the eye for the per-pixel list; forced
    screen bins keep their lists only while these stay short.  Whatever is chosen, the frame is the same."""
    frames = {}
    for size, path in (((96, 72), 0), ((320, 240), 1)):  # 10 faces per pixel, then 0.9
        auto = R.Flyscene(model=scene_path("bunny"), output="").initialize(*size)
        tree = R.Flyscene(model=scene_path("bunny"), output="", primary_rays=L.PRIMARY_RAYS_TREE).initialize(*size)
        first, second = auto.render(), auto.render()
        assert first["stats"].primary_path == path and second["stats"].primary_path == path
        ref = tree.render()
        assert ref["stats"].primary_path == 0
        for r in (first, second):
            assert np.array_equal(r["face_id"], ref["face_id"]) and np.array_equal(r["rgb"].view(np.uint32), ref["rgb"].view(np.uint32))
        frames[size] = first
    # the eye inside the mesh: hundreds of faces have a corner beside or behind the eye; the pass gives way to the tree walk,
    # and the next frames of the view do not try again
    inside = R.Flyscene(model=scene_path("toy"), output="").initialize(160, 120)
    walk = R.Flyscene(model=scene_path("toy"), output="", primary_rays=L.PRIMARY_RAYS_TREE).initialize(160, 120)
    for f in (inside, walk):
        f.camera_translate(0.0, 0.0, 38.0)
        f.camera_rotate(0.3, -0.7)
    got = [inside.render() for _ in range(3)]
    ref = walk.render()
    for r in got:
        assert np.array_equal(r["face_id"], ref["face_id"]) and np.array_equal(r["rgb"].view(np.uint32), ref["rgb"].view(np.uint32))
    # forced bins: a dense mesh in a small image gives long lists, which AUTO-with-bins would turn down; forced, they are kept
    small = R.Flyscene(model=scene_path("bunny"), output="", primary_rays=L.PRIMARY_RAYS_BINS).initialize(320, 240)
    st = small.render()["stats"]
    assert st.primary_path == 1 and st.bin_entries > 38 * st.bins_nonempty > 0
    large = R.Flyscene(model=scene_path("bunny"), output="", primary_rays=L.PRIMARY_RAYS_BINS).initialize(1600, 1200)
    st = [large.render(face_ids=False)["stats"] for _ in range(2)][-1]
    assert st.primary_path == 1 and 0 < st.bin_entries <= 38 * st.bins_nonempty


def test_view_change_with_frames_in_flight_and_bands_with_different_verdicts():
    """Whether the screen-space pass of the primary rays stands or gives way to the tree walk is decided on the device, per
    pass and per band of rows; what earlier frames reported is only a hint for launch sizes.  (1) Frames of a view the pass
    accepts are still in flight when frames of a view it turns down (eye inside the mesh) are enqueued behind them: the
    stale report must not cost the second view its walk.  (2) The same view in bands of rows, of which some accept and
    some give way, frame after frame."""
    import torch

    W, H = 160, 120
    auto = R.Flyscene(model=scene_path("toy"), output="").initialize(W, H)
    walk = R.Flyscene(model=scene_path("toy"), output="", primary_rays=L.PRIMARY_RAYS_TREE).initialize(W, H)
    outside = walk.render()
    out_cam = auto.getCamera()
    for f in (auto, walk):
        f.camera_translate(0.0, 0.0, 38.0)
        f.camera_rotate(0.3, -0.7)
    inside = walk.render()
    in_cam = auto.getCamera()
    assert not np.array_equal(outside["face_id"], inside["face_id"])
    stream = torch.cuda.Stream()
    for trial in range(3):
        bufs = []
        for k in range(10):  # five frames of the outside view, then five of the inside view, no wait in between
            rgb = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
            fid = torch.full((H, W), -7, dtype=torch.int32, device="cuda")
            auto.render_device(rgb, fid, camera=out_cam if k < 5 else in_cam, stream=stream)
            bufs.append((rgb, fid))
        torch.cuda.synchronize()
        for k, (rgb, fid) in enumerate(bufs):
            ref = outside if k < 5 else inside
            assert np.array_equal(fid.cpu().numpy(), ref["face_id"]), (trial, k)
            assert np.array_equal(rgb.cpu().numpy().view(np.uint32), ref["rgb"].view(np.uint32)), (trial, k)
    # bands: the faces that reach round the eye are counted per band
    banded = R.Flyscene(model=scene_path("toy"), output="").initialize(W, H)
    banded.camera_translate(0.0, 0.0, 38.0)
    banded.camera_rotate(0.3, -0.7)
    banded.render()
    banded.set_workspace_limit(banded.workspace_bytes() // 6)
    for again in range(4):
        got = banded.render()
        assert np.array_equal(got["face_id"], inside["face_id"]), again
        assert np.array_equal(got["rgb"].view(np.uint32), inside["rgb"].view(np.uint32)), again


def test_add_light_after_initialize_is_shaded_but_casts_no_shadow_rays():
    """Flyscene::addLight (flyscene.hpp:76) only appends to `lights`; the shadow sample points were made once, in
    initialize (flyscene.cpp:117).  The new light takes part in shadeOffFace's loop (:505) and not in getLightIntensity's
    (:880): L + 1 shading lights, L sampled ones."""
    from oracle.oracle_py import OracleScene

    for name, depth in (("toy_on_plane", 0), ("reflect_test", 3)):
        fly, orc = make_pair(name, width=96, height=96)
        before = fly.render()
        fly.camera_translate(6.0, -4.0, 9.0)
        fly.addLight()
        centre = tuple(fly.getCamera().origin)
        orc.add_light(centre)
        r = compare(fly, orc, 96, 96, max_reflections=depth)
        assert_parity(r)
        st = r["got"]["stats"]
        assert st.rays_shadow == st.hits * 13  # still the thirteen sample points of the first light
        assert fly.arrays()["light_samples"].shape == (13, 3)
        fly.camera_reset()
        after = fly.render()
        assert np.abs(after["rgb"] - before["rgb"]).max() > 0.05  # the light is there ...
        # ... and a scene whose second light WAS sampled looks different: that one casts shadow rays
        both = R.Flyscene(model=scene_path(name), output="", lights=[(-1.0, 1.0, 1.0), centre]).initialize(96, 96)
        both.set_params(max_reflections=depth)
        sb = both.render()["stats"]
        assert sb.rays_shadow == sb.hits * 26 and sb.hits > 0
    # no light at initialize: getLightIntensity sums over nothing, the light of addLight shades with intensity zero
    fly = R.Flyscene(model=scene_path("NthTest"), output="", lights=[]).initialize(64, 64)
    orc = OracleScene(scene_path("NthTest"), lights=[])
    fly.addLight()
    orc.add_light(tuple(fly.getCamera().origin))
    r = compare(fly, orc, 64, 64)
    assert_parity(r)
    assert r["got"]["stats"].rays_shadow == 0 and (r["got"]["face_id"] >= 0).any()


@pytest.mark.parametrize("name", ["NthTest", "toy_on_plane", "bunny", "reflect_test", "textured_quad", "glass"])
def test_device_built_hierarchy_gives_the_same_frame(name):
    """The hierarchy built on the GPU (Morton-order radix tree) replaces the host SAH tree without changing a bit
    of the frame: every ray kind is sent through it (no bins, no grids) and compared with the default path."""
    kw = dict(model=scene_path(name), output="")
    a = R.Flyscene(**kw).initialize(120, 88)
    b = R.Flyscene(builder=L.BUILDER_DEVICE_LBVH, primary_rays=L.PRIMARY_RAYS_TREE, shadow_rays=L.SHADOW_RAYS_TREE, **kw).initialize(120, 88)
    c = R.Flyscene(builder=L.BUILDER_DEVICE_LBVH, max_leaf_faces=1, **kw).initialize(120, 88)
    assert a.info().builder == L.BUILDER_HOST_SAH
    assert b.info().builder == L.BUILDER_DEVICE_LBVH and c.info().builder == L.BUILDER_DEVICE_LBVH
    assert b.info().num_device_faces == a.info().num_device_faces
    # with the device builder the shadow grids are built on the GPU as well: the same (face, cell) pairs as on the host
    assert c.info().shadow_grid_entries == a.info().shadow_grid_entries > 0 and b.info().shadow_grid_entries == 0
    for f in (a, b, c):
        f.camera_translate(1.0, -0.5, 6.0)
        f.camera_rotate(-0.1, 0.2)
    ra = a.render()
    assert (ra["face_id"] >= 0).any() and ra["stats"].rays_shadow > 0
    for other in (b, c):
        ro = other.render()
        assert np.array_equal(ra["face_id"], ro["face_id"])
        assert np.array_equal(ra["rgb"].view(np.uint32), ro["rgb"].view(np.uint32))
        sa, so = ra["stats"], ro["stats"]
        assert (sa.rays_primary, sa.rays_shadow, sa.rays_bounce, sa.hits) == (so.rays_primary, so.rays_shadow, so.rays_bounce, so.hits)


def test_device_builder_on_a_large_mesh_and_its_fallback():
    import torch

    w = workloads.get("field2_mirrors_small")
    frames = []
    for build in (dict(), dict(builder=L.BUILDER_DEVICE_LBVH)):
        fly, params, cam = workloads.make_flyscene(w, **build)
        rgb = torch.zeros((w["height"], w["width"], 3), dtype=torch.float32, device="cuda")
        fid = torch.zeros((w["height"], w["width"]), dtype=torch.int32, device="cuda")
        fly.render_device(rgb, fid, camera=cam)
        torch.cuda.synchronize()
        frames.append((fly.info(), rgb.cpu().numpy(), fid.cpu().numpy(), fly.stats()))
    (ia, rgb_a, fid_a, sa), (ib, rgb_b, fid_b, sb) = frames
    assert ia.builder == L.BUILDER_HOST_SAH and ib.builder == L.BUILDER_DEVICE_LBVH and ib.num_device_faces == ia.num_device_faces
    assert ib.hierarchy_ms < ia.hierarchy_ms
    assert ib.shadow_grid_entries == ia.shadow_grid_entries > 0
    assert np.array_equal(fid_a, fid_b)
    assert np.array_equal(rgb_a.view(np.uint32), rgb_b.view(np.uint32))
    assert sa.rays_bounce == sb.rays_bounce > 0 and sa.rays_shadow == sb.rays_shadow
    # one face: nothing to sort, the host builder takes over and says so
    one = R.Flyscene(arrays=(np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32), None, np.array([[0, 1, 2]], np.int32),
                             np.array([0], np.int32), [dict(ka=(0, 0, 0), kd=(1, 0, 0), ks=(0, 0, 0), shininess=1.0)]),
                     builder=L.BUILDER_DEVICE_LBVH, output="").initialize(32, 32)
    assert one.info().builder == L.BUILDER_HOST_SAH
    assert (one.render()["face_id"] >= 0).any()


def test_frames_in_flight_on_several_streams():
    """Every stream a caller renders on gets a workspace of its own: frames enqueued on different streams may run
    at the same time and still give the frames a single stream gives, also with more streams than workspaces."""
    import torch

    fly, _ = make_pair("reflect_test", width=160, height=120)
    cams = []
    for k in range(18):
        fly.camera_translate(0.13, -0.07, 0.33)
        fly.camera_rotate(0.028, -0.017)
        cams.append(fly.getCamera())
    shape = (120, 160)
    want = []
    for cam in cams:
        rgb = torch.zeros(shape + (3,), dtype=torch.float32, device="cuda")
        fid = torch.zeros(shape, dtype=torch.int32, device="cuda")
        fly.render_device(rgb, fid, camera=cam)
        torch.cuda.synchronize()
        want.append((rgb.cpu().numpy(), fid.cpu().numpy()))
    assert fly.stats().rays_bounce > 0
    assert not np.array_equal(want[0][1], want[17][1])
    for nstreams in (2, 3, 18):  # 18 > the sixteen workspaces a scene keeps: the least recently used one changes hands
        streams = [torch.cuda.Stream() for _ in range(nstreams)]
        outs = [(torch.zeros(shape + (3,), dtype=torch.float32, device="cuda"), torch.zeros(shape, dtype=torch.int32, device="cuda"))
                for _ in cams]
        torch.cuda.synchronize()
        for rounds in range(3):
            for k, cam in enumerate(cams):
                fly.render_device(outs[k][0], outs[k][1], camera=cam, stream=streams[k % nstreams])
        torch.cuda.synchronize()
        for k in range(len(cams)):
            assert np.array_equal(outs[k][1].cpu().numpy(), want[k][1]), (nstreams, k)
            assert np.array_equal(outs[k][0].cpu().numpy().view(np.uint32), want[k][0].view(np.uint32)), (nstreams, k)


def _random_soup(seed):
    """Triangle soup with the cases that decide ties and rejections: coincident faces with different materials
    (the reference keeps the one it meets last), overlapping coplanar faces, zero-area faces, axis-aligned
    faces (rays inside a face's plane), reflective and dull materials."""
    rng = np.random.default_rng(seed)
    n = [1, 2, 7, 40, 300, 2500, 9000, 64][seed % 8]
    centres = rng.uniform(-1.0, 1.0, (n, 1, 3))
    size = [1.5, 1.0, 0.8, 0.5, 0.25, 0.08, 0.05, 0.6][seed % 8]
    tri = (centres + rng.uniform(-size, size, (n, 3, 3))).astype(np.float32)
    k = max(1, n // 8)
    tri[:k, :, 2] = np.float32(0.25)                      # axis-aligned, coplanar, overlapping
    if n >= 4:
        tri[k:2 * k] = tri[:k]                            # coincident copies (other materials below)
        tri[2 * k] = tri[2 * k][[0, 0, 1]]                # zero area
        tri[2 * k + 1 if 2 * k + 1 < n else 0, :, 0] = np.float32(-0.3)
    verts = tri.reshape(-1, 3)
    faces = np.arange(3 * n, dtype=np.int32).reshape(n, 3)
    nmat = 4
    mats = []
    for m in range(nmat):
        ks = float(rng.choice([0.0, 0.1, 0.3, 0.9]))
        mats.append(dict(ka=rng.uniform(0, 0.2, 3).tolist(), kd=rng.uniform(0.1, 1.0, 3).tolist(), ks=[ks] * 3,
                         shininess=float(rng.choice([1.0, 8.0, 64.0]))))
    fmat = rng.integers(0, nmat, n).astype(np.int32)
    lights = [tuple(rng.uniform(-2.0, 2.0, 3).tolist()) for _ in range(1 + seed % 3)]
    return (verts, None, faces, fmat, mats), lights


@pytest.mark.parametrize("seed", range(16))
def test_random_triangle_soups_every_path(seed):
    """Random soups through (a) the default path (screen bins, shadow grids, host SAH tree), (b) the tree for every
    ray, (c) the device-built tree: each against the oracle, and all three bit-identical to each other."""
    arrays, lights = _random_soup(seed)
    kw = dict(arrays=arrays, lights=lights, soft_points=seed % 4, width=72, height=56)
    fly, orc = make_pair(**kw)
    r = compare(fly, orc, 72, 56, max_reflections=3)
    assert_parity(r)
    base = r["got"]
    for build in (dict(primary_rays=L.PRIMARY_RAYS_TREE, shadow_rays=L.SHADOW_RAYS_TREE),
                  dict(builder=L.BUILDER_DEVICE_LBVH, primary_rays=L.PRIMARY_RAYS_TREE, shadow_rays=L.SHADOW_RAYS_TREE),
                  dict(builder=L.BUILDER_DEVICE_LBVH, max_leaf_faces=1)):
        other = R.Flyscene(arrays=arrays, lights=lights, soft_points=seed % 4, output="", **build).initialize(72, 56)
        other.set_params(max_reflections=3)
        got = other.render()
        assert np.array_equal(got["face_id"], base["face_id"]), build
        assert np.array_equal(got["rgb"].view(np.uint32), base["rgb"].view(np.uint32)), build
        so, sb = got["stats"], base["stats"]
        assert (so.rays_shadow, so.rays_bounce, so.hits) == (sb.rays_shadow, sb.rays_bounce, sb.hits), build


def test_degenerate_sample_direction_falls_back_to_the_tree():
    # a light at the origin gives a zero shadow-ray direction (its hard-shadow sample): no grid can be built for it
    fly, orc = make_pair("glass", width=48, height=48, lights=[(0.0, 0.0, 0.0)], soft_points=0)
    assert fly.info().shadow_grid_entries == 0
    assert_parity(compare(fly, orc, 48, 48))


def test_non_square_and_ragged_sizes():
    fly, orc = make_pair("toy", width=200, height=120)
    assert_parity(compare(fly, orc, 200, 120))
    for w, h in [(1, 1), (7, 3), (17, 33), (65, 15)]:
        fly, orc = make_pair("reflect_test", width=w, height=h)
        assert_parity(compare(fly, orc, w, h))


@pytest.mark.parametrize("size", [(65535, 9), (65536, 9), (9, 65537), (70001, 5)])
def test_images_at_and_beyond_the_screen_space_limit(size):
    """The face pass and the screen bins pack pixel columns and rows into 16 bits: an image 65 536 wide or high goes through
    the tree walk, one of 65 535 still through the face pass.  Both against the oracle."""
    w, h = size
    fly, orc = make_pair("reflect_test", width=w, height=h)
    r = compare(fly, orc, w, h, max_reflections=2)
    assert_parity(r)
    assert r["got"]["stats"].primary_path == (1 if max(w, h) < 65536 else 0)
    assert (r["got"]["face_id"] >= 0).sum() >= 2  # (a strip that wide sees the scene in its middle pixels only)


def test_hard_shadows_two_lights_no_bounce():
    lights = [(-1.0, 1.0, 1.0), (1.0, 1.0, 1.0)]
    fly, orc = make_pair("bunny", lights=lights, soft_points=0, width=128, height=128)
    assert_parity(compare(fly, orc, 128, 128, max_reflections=0))


def test_many_lights_deep_reflections():
    import math

    ring = [(1.5 * math.cos(2 * math.pi * k / 8), 1.0, 1.5 * math.sin(2 * math.pi * k / 8)) for k in range(8)]
    fly, orc = make_pair("reflect_test", lights=ring, soft_points=0, width=96, height=96)
    assert_parity(compare(fly, orc, 96, 96, max_reflections=4))
    fly, orc = make_pair("toy", lights=ring[:3], soft_points=2, width=64, height=64)
    assert_parity(compare(fly, orc, 64, 64, max_reflections=3))


def test_more_lights_than_fit_the_shading_kernels_lds_and_the_shadow_ray_limit():
    """The shading kernel keeps up to 64 light positions in LDS and reads more from memory; a hit casts at most 512 shadow
    rays (lights x samples).  70 lights with hard shadows, 39 lights with 13 samples each (507 rays per hit) and lights added
    after the samples were made, against the oracle; 40 x 13 = 520 is refused at scene creation."""
    rng = np.random.default_rng(3)
    many = [tuple(p) for p in (rng.uniform(-1.0, 1.0, (70, 3)) * [2.0, 0.5, 2.0] + [0.0, 1.5, 0.5]).tolist()]
    fly, orc = make_pair("reflect_test", lights=many, soft_points=0, width=48, height=40)
    r = compare(fly, orc, 48, 40, max_reflections=2)
    assert_parity(r)
    assert r["rays_gpu"][1] == 70 * r["got"]["stats"].hits
    fly, orc = make_pair("toy", lights=many[:39], soft_points=12, width=40, height=32)
    r = compare(fly, orc, 40, 32, max_reflections=1)
    assert_parity(r)
    assert r["rays_gpu"][1] == 507 * r["got"]["stats"].hits
    # lights added after initialize are shaded but cast no shadow rays: 60 sampled + 10 unsampled = 70 in the Phong loop
    fly, orc = make_pair("reflect_test", lights=many[:60], soft_points=1, width=48, height=40)
    for k in range(10):  # (Flyscene::addLight puts the light where the camera is, src/flyscene.hpp:76)
        fly.camera_translate(0.3 * (k % 3 - 1), 0.2, -0.5 * (k % 2))
        fly.addLight()
        orc.add_light(list(fly.getCamera().origin))
    r = compare(fly, orc, 48, 40, max_reflections=1)
    assert_parity(r)
    assert r["rays_gpu"][1] == 120 * r["got"]["stats"].hits
    with pytest.raises(RuntimeError, match="shadow rays per hit"):
        make_pair("toy", lights=many[:40], soft_points=12, width=40, height=32)


def test_as_many_passes_and_levels_as_the_stage_counters_hold():
    """A frame has ssaa^2 * (max_reflections + 1) stages and counters for 64 of them: 16 passes of four levels and one pass of
    64 levels against the oracle, one stage more refused."""
    fly, orc = make_pair("reflect_test", width=40, height=36)
    assert_parity(compare(fly, orc, 40, 36, ssaa_level=4, max_reflections=3))
    assert_parity(compare(fly, orc, 40, 36, ssaa_level=1, max_reflections=63))
    assert_parity(compare(fly, orc, 40, 36, ssaa_level=8, max_reflections=0))
    for kw, why in ((dict(ssaa_level=4, max_reflections=4), "exceeds 64"), (dict(ssaa_level=1, max_reflections=64), "exceeds 64"),
                    (dict(ssaa_level=8, max_reflections=1), "exceeds 64"), (dict(ssaa_level=9, max_reflections=0), "outside 1..8")):
        fly.set_params(**kw)
        with pytest.raises(RuntimeError, match=why):
            fly.render()


def test_bounce_depth_sweep():
    fly, orc = make_pair("reflect_test", width=64, height=64)
    rays = []
    for depth in (0, 1, 2, 5):
        r = compare(fly, orc, 64, 64, max_reflections=depth)
        assert_parity(r)
        rays.append(r["rays_gpu"][2])
    assert rays[0] == 0 and rays == sorted(rays)


def test_ssaa():
    fly, orc = make_pair("shadowmodel", width=48, height=48)
    assert_parity(compare(fly, orc, 48, 48, ssaa_level=2))
    fly, orc = make_pair("toy", width=40, height=40)
    assert_parity(compare(fly, orc, 40, 40, ssaa_level=3, max_reflections=1))


def test_spheres():
    spheres = [(0.0, 0.0, 0.0, 0.3), (0.45, 0.2, 0.2, 0.15)]
    fly, orc = make_pair("reflect_test", spheres=spheres, width=96, height=96)
    r = compare(fly, orc, 96, 96)
    assert_parity(r)
    assert (r["got"]["face_id"] == L.FACE_SPHERE).sum() > 100


def test_changed_literals():
    fly, orc = make_pair("glass", width=64, height=64)
    assert_parity(compare(fly, orc, 64, 64, reflect_threshold=0.6, bounce_offset=0.02, t_epsilon=1e-4,
                          background=(0.1, 0.2, 0.3), background_deep=(0.05, 0.0, 0.1)))


def test_moved_and_rotated_camera():
    fly, orc = make_pair("toy", width=96, height=96)
    fly.camera_translate(2.0, -1.0, 6.0)
    assert_parity(compare(fly, orc, 96, 96))
    fly.camera_rotate(0.2, -0.3)
    assert_parity(compare(fly, orc, 96, 96))


def test_scene_without_materials_and_empty_scene():
    v = np.array([[-1, -1, 0], [1, -1, 0], [1, 1, 0], [-1, 1, 0]], np.float32)
    f = np.array([[0, 1, 2], [0, 2, 3]], np.int32)
    arrays = (v, None, f, np.full(2, -1, np.int32), [])
    fly, orc = make_pair(arrays=arrays, width=32, height=32)
    r = compare(fly, orc, 32, 32)
    assert_parity(r)
    assert np.allclose(r["got"]["rgb"][16, 16], 0.5)  # material -1 -> grey (flyscene.cpp:469-472)
    # equal distance on the shared diagonal: the later face wins (flyscene.cpp:760)
    assert r["got"]["face_id"][16, 16] == r["ref"]["face_id"][16, 16]
    empty = (np.zeros((3, 3), np.float32), None, np.zeros((0, 3), np.int32), np.zeros(0, np.int32), [])
    fly = R.Flyscene(arrays=empty, output="").initialize(16, 16)
    got = fly.render()
    assert (got["face_id"] == L.FACE_MISS).all() and np.allclose(got["rgb"], np.float32(0.4))


def test_device_entry_point_matches_host_entry_point_and_is_repeatable():
    import torch

    fly, _ = make_pair("toy", width=128, height=72)
    host = fly.render()
    rgb = torch.zeros((72, 128, 3), dtype=torch.float32, device="cuda")
    fid = torch.zeros((72, 128), dtype=torch.int32, device="cuda")
    t = torch.zeros((72, 128), dtype=torch.float32, device="cuda")
    for _ in range(3):
        fly.render_device(rgb, fid, t)
    torch.cuda.synchronize()
    assert np.array_equal(rgb.cpu().numpy().view(np.uint32), host["rgb"].view(np.uint32))
    assert np.array_equal(fid.cpu().numpy(), host["face_id"])
    tt = t.cpu().numpy()
    assert ((tt > 0) == (host["face_id"] != L.FACE_MISS)).all()
    st = fly.stats()
    assert st.rays_shadow == st.hits * 13


@pytest.mark.parametrize("parts,tile,kw", [(2, 16, {}), (3, 8, {}), (8, 16, {}), (5, 7, {}), (3, 16, dict(ssaa_level=2, max_reflections=2)),
                                           (4, 5, dict(ssaa_level=3, max_reflections=1))])
def test_row_tile_parts_assemble_to_the_full_frame(parts, tile, kw):
    """Every part rendered on its own and interleaved back must equal the single-part frame bit for bit."""
    import torch

    fly, _ = make_pair("toy", width=96, height=100)
    fly.set_params(**kw)
    full = fly.render()
    base = fly.params_copy(row_tile=tile)
    packed, packed_id = [], []
    for part in range(parts):
        p = tiles.part_params(base, parts, part)
        rows = fly.local_rows(p)
        rgb = torch.zeros((max(rows, 1), 96, 3), dtype=torch.float32, device="cuda")
        fid = torch.zeros((max(rows, 1), 96), dtype=torch.int32, device="cuda")
        fly.render_device(rgb, fid, params=p)
        packed.append(rgb[:rows])
        packed_id.append(fid[:rows])
    torch.cuda.synchronize()
    cat = torch.cat(packed, 0).contiguous()
    img = cat.index_select(0, tiles.row_permutation(base, parts, "cuda"))
    assert np.array_equal(img.cpu().numpy().view(np.uint32), full["rgb"].view(np.uint32))
    # the C ABI's own assembly kernel gives the same image
    out = torch.zeros((100, 96, 3), dtype=torch.float32, device="cuda")
    pp = tiles.part_params(base, parts, 0)
    L.check(L.lib().rtamd_assemble_rows(C.byref(pp), C.c_void_p(cat.data_ptr()), C.c_void_p(out.data_ptr()), 3,
                                        C.c_void_p(torch.cuda.current_stream().cuda_stream)), "assemble")
    torch.cuda.synchronize()
    assert torch.equal(out, img)
    ids = torch.cat(packed_id, 0).index_select(0, tiles.row_permutation(base, parts, "cuda"))
    assert np.array_equal(ids.cpu().numpy(), full["face_id"])


@pytest.mark.parametrize("name,size,kw", [("reflect_test", (200, 203), dict(max_reflections=3)),
                                          ("toy", (96, 100), dict(ssaa_level=2, max_reflections=1)),
                                          ("bunny", (333, 222), dict(max_reflections=0)),
                                          ("textured_quad", (150, 90), dict(max_reflections=1)),
                                          ("glass", (128, 136), dict(max_reflections=2, background=(0.1, 0.2, 0.3)))])
def test_frames_rendered_in_bands_equal_whole_frames(name, size, kw):
    """rtamd_scene_set_workspace_limit: a frame rendered band by band through work buffers of one band's size is the whole
    frame bit for bit, ray counts included, through both entry points and again once the launches are sized by feedback."""
    import torch

    W, H = size
    whole, _ = make_pair(name, width=W, height=H)
    banded, _ = make_pair(name, width=W, height=H)
    whole.set_params(**kw)
    banded.set_params(**kw)
    whole.set_workspace_limit(0)

    def frame(fly, **pk):
        rgb = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
        fid = torch.full((H, W), -7, dtype=torch.int32, device="cuda")
        t = torch.zeros((H, W), dtype=torch.float32, device="cuda")
        fly.render_device(rgb, fid, t, params=fly.params_copy(**pk) if pk else None)
        torch.cuda.synchronize()
        st = fly.stats()
        return rgb.cpu().numpy().view(np.uint32), fid.cpu().numpy(), t.cpu().numpy().view(np.uint32), st

    ref = frame(whole)
    banded.set_workspace_limit(whole.workspace_bytes() // 5)
    for _ in range(3):
        got = frame(banded)
        for a, b in zip(ref[:3], got[:3]):
            assert np.array_equal(a, b)
        for field in ("rays_primary", "hits", "rays_shadow", "rays_bounce"):
            assert getattr(ref[3], field) == getattr(got[3], field), field
    assert banded.workspace_bytes() < whole.workspace_bytes() // 2
    host = banded.render()
    assert np.array_equal(host["rgb"].view(np.uint32), ref[0]) and np.array_equal(host["face_id"], ref[1])
    assert host["stats"].rays_shadow == ref[3].rays_shadow and host["stats"].rays_bounce == ref[3].rays_bounce
    # a frame timed kernel by kernel is still one frame, whatever the number of bands
    timed = frame(banded, flags=L.FLAG_TIME_KERNELS)
    assert np.array_equal(timed[0], ref[0]) and timed[3].frames == 1 and timed[3].ms_frames > 0
    # the tree walk for the primary rays, band by band
    tree, _ = make_pair(name, width=W, height=H, primary_rays=L.PRIMARY_RAYS_TREE)
    tree.set_params(**kw)
    tree.set_workspace_limit(whole.workspace_bytes() // 5)
    got = frame(tree)
    assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1]) and got[3].rays_bounce == ref[3].rays_bounce


def test_reflective_frames_beside_more_foreign_streams_than_hardware_queues():
    """A reflective scene rendered on one stream while six foreign streams carry work of their own, under HIP's default of
    four hardware queues: every frame runs entirely on its caller's stream (the library owns no streams and waits on no
    event of another stream), so nothing can stall behind a queue it shares; the frames equal the one rendered alone."""
    import torch

    fly, _ = make_pair("reflect_test", width=160, height=120)
    fly.set_params(max_reflections=3)
    want = fly.render()
    foreign = [torch.cuda.Stream() for _ in range(6)]
    mine = torch.cuda.Stream()
    a = [torch.rand((512, 512), device="cuda") for _ in foreign]
    rgb = [torch.zeros((120, 160, 3), dtype=torch.float32, device="cuda") for _ in range(3)]
    fid = torch.zeros((120, 160), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    for turn in range(12):
        for s, m in zip(foreign, a):
            with torch.cuda.stream(s):
                for _ in range(4):
                    m = torch.tanh(m @ m * 1e-3)
        fly.render_device(rgb[turn % 3], fid, stream=mine)
    torch.cuda.synchronize()
    for r in rgb:
        assert np.array_equal(r.cpu().numpy().view(np.uint32), want["rgb"].view(np.uint32))
    assert np.array_equal(fid.cpu().numpy(), want["face_id"])


def test_device_ppm_quantisation_matches_the_file_writer(tmp_path):
    """rtamd_quantize_ppm gives the bytes of the reference's writePPMImage (utils/ppmIO.hpp:203), also at channel
    values of exactly k/255, above 1, and for lengths that are no multiple of four."""
    import torch

    rng = np.random.default_rng(5)
    vals = np.concatenate([rng.random(100003, dtype=np.float32) * 1.2, (np.arange(256) / 255.0).astype(np.float32),
                           np.array([0.0, 1.0, 1.0000001, 0.99999994, 3.5, 1e-30], np.float32)]).astype(np.float32)
    want = np.minimum(255, (np.float32(255) * vals).astype(np.int32)).astype(np.uint8)
    for n in (len(vals), 1, 2, 3, 1001):
        src = torch.from_numpy(vals[:n].copy()).cuda()
        out = torch.zeros(n + 8, dtype=torch.uint8, device="cuda")
        tiles.quantize_ppm(src, out, n)
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        assert np.array_equal(got[:n], want[:n]) and not got[n:].any()
    # and the frame a Flyscene writes as P6 holds exactly these bytes
    fly, _ = make_pair("toy", width=64, height=48)
    frame = fly.render()["rgb"]
    path = str(tmp_path / "frame.p6.ppm")
    from raytracer_amd.flyscene import write_ppm

    write_ppm(path, frame, binary=True)
    payload = np.frombuffer(open(path, "rb").read()[-64 * 48 * 3:], np.uint8)
    dev = torch.zeros(64 * 48 * 3, dtype=torch.uint8, device="cuda")
    tiles.quantize_ppm(torch.from_numpy(frame).cuda(), dev)
    assert np.array_equal(dev.cpu().numpy(), payload)


def test_bvh_leaf_size_does_not_change_the_image():
    a = R.Flyscene(model=scene_path("toy"), output="", max_leaf_faces=1).initialize(96, 96).render()
    b = R.Flyscene(model=scene_path("toy"), output="", max_leaf_faces=8, sah_bins=8).initialize(96, 96).render()
    assert np.array_equal(a["face_id"], b["face_id"]) and np.array_equal(a["rgb"].view(np.uint32), b["rgb"].view(np.uint32))


def test_counted_frame_matches_plain_frame():
    fly, _ = make_pair("toy", width=96, height=96)
    plain = fly.render()
    fly.set_params(flags=L.FLAG_COUNT_WORK | L.FLAG_TIME_KERNELS)
    counted = fly.render()
    assert np.array_equal(plain["rgb"].view(np.uint32), counted["rgb"].view(np.uint32))
    st = counted["stats"]
    # [0] BVH walk for closest hits, [1] shadow rays, [2] screen-bin pass of the primary rays (when it is used)
    assert st.box_tests[0] > 0 and st.box_tests[1] > 0
    assert st.tri_plane[0] + st.tri_plane[2] >= st.tri_edge[0] + st.tri_edge[2] > 0
    assert st.launches_trace + st.launches_raster >= 2 and st.launches_shadow >= 1
    assert st.ms_frames > 0 and st.ms_shade > 0 and st.ms_compact > 0 and st.ms_trace + st.ms_raster > 0


def test_raytrace_scene_writes_the_reference_ppm(tmp_path):
    from oracle.oracle_py import OracleScene, write_ppm

    out = tmp_path / "result.ppm"
    fly = R.Flyscene(model=scene_path("reflect_test"), output=str(out)).initialize(64, 64)
    fly.raytraceScene()
    ref = OracleScene(scene_path("reflect_test")).render(64, 64)
    write_ppm(str(tmp_path / "ref.ppm"), ref["rgb"])
    a, b = out.read_text().split(), (tmp_path / "ref.ppm").read_text().split()
    assert a[:4] == ["P3", "64", "64", "255"] and len(a) == len(b) == 4 + 64 * 64 * 3
    diff = np.abs(np.array(a[4:], int) - np.array(b[4:], int))
    assert diff.max() <= 1  # a last-bit colour difference can cross an integer boundary of 255*c
    # raytraceScene fetches the frame as the file's 8-bit pixels (rtamd_render_ppm): the text is what the float writer makes of
    # the float framebuffer of the same frame, byte for byte
    got = fly.render()
    R.flyscene.write_ppm(str(tmp_path / "floats.ppm"), got["rgb"])
    assert out.read_bytes() == (tmp_path / "floats.ppm").read_bytes()
    pixels = np.zeros((64, 64, 3), np.uint8)
    exact = C.c_int32(-1)
    L.check(L.lib().rtamd_render_ppm(fly.scene, C.byref(fly.getCamera()), C.byref(fly.params), pixels.ctypes.data_as(C.c_void_p), C.byref(exact), None), "rtamd_render_ppm")
    assert exact.value == 1
    assert np.array_equal(pixels, np.minimum(255, (np.float32(255) * got["rgb"]).astype(np.int32)).astype(np.uint8))
    # a channel the bytes cannot express (the reference prints a negative number): the frame is fetched as floats instead
    fly.set_params(background=(-0.5, 0.25, 2.0))
    fly.raytraceScene()
    neg = fly.render()
    R.flyscene.write_ppm(str(tmp_path / "neg.ppm"), neg["rgb"])
    assert out.read_bytes() == (tmp_path / "neg.ppm").read_bytes() and b"-127 63 255 " in out.read_bytes()
    L.check(L.lib().rtamd_render_ppm(fly.scene, C.byref(fly.getCamera()), C.byref(fly.params), pixels.ctypes.data_as(C.c_void_p), C.byref(exact), None), "rtamd_render_ppm")
    assert exact.value == 0


# ---- the benchmark configurations at full size: properties that do not need the oracle on every pixel

def _rows_sample_parity(wname, rows_step, tol_pixels=0):
    """Full-size frame on the GPU; the oracle renders every rows_step-th row of it."""
    from oracle.oracle_py import OracleScene

    w = workloads.get(wname)
    fly, params, cam = workloads.make_flyscene(w)
    import torch

    W, H = w["width"], w["height"]
    rgb = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    fid = torch.zeros((H, W), dtype=torch.int32, device="cuda")
    fly.render_device(rgb, fid, camera=cam)
    torch.cuda.synchronize()
    st = fly.stats()
    assert st.rays_primary == W * H and st.rays_shadow == st.hits * len(w["lights"]) * (1 + w["soft_points"])
    lights = [tuple(l) + (1.0, 1.0, 1.0) for l in w["lights"]]
    orc = OracleScene(arrays=workloads.mesh_arrays(w), lights=lights, soft_points=w["soft_points"])
    ref = orc.render(W, H, camera=oracle_camera(orc, cam), max_reflections=w["max_reflections"], row_offset=rows_step // 2,
                     row_step=rows_step, reject_nan=1)
    rows = list(range(rows_step // 2, H, rows_step))
    g_id, g_rgb = fid.cpu().numpy()[rows], rgb.cpu().numpy()[rows]
    same = g_id == ref["face_id"][rows]
    assert (~same).sum() <= tol_pixels and same.mean() >= FACE_ID_MIN_MATCH
    assert np.abs(g_rgb - ref["rgb"][rows])[same].max() <= PIXEL_TOL
    return fly, rgb, fid


def test_config1_bunny_1000_rows_sample():
    _rows_sample_parity("bunny_1000", 25)


def test_config2_bunny_1080p_bounce_rows_sample():
    _rows_sample_parity("bunny_1080p", 40)


def test_config3_bunny_4k_two_lights_hard_shadows_rows_sample():
    _rows_sample_parity("bunny_4k", 90)


def test_config3_subdivided_bunny_1m_faces_4k_rows_sample():
    # 1.1 M faces in ONE object (config 3's upper end): every face smaller than a pixel
    _rows_sample_parity("bunny_sub2_4k", 180)


def test_config5_mirror_field_8k_rows_sample_and_partition():
    """Mirrors + 8 lights, depth-4 reflections at 7680x4320: a 1/360 sample of rows against the oracle, ray
    bookkeeping, and the frame rendered as 8 row-tile parts equals the single-part frame bit for bit."""
    import torch

    fly, rgb, fid = _rows_sample_parity("field_mirrors_8k", 360)
    st = fly.stats()
    assert st.rays_bounce > 10 ** 6
    w = workloads.get("field_mirrors_8k")
    cam = workloads.look_at(w["camera"]["eye"], w["camera"]["target"], w["width"], w["height"])
    base = fly.params_copy(row_tile=16)
    index = tiles.row_permutation(base, 8, "cuda")
    offset = 0
    for part in range(8):
        p = tiles.part_params(base, 8, part)
        rows = fly.local_rows(p)
        buf = torch.zeros((rows, w["width"], 3), dtype=torch.float32, device="cuda")
        fly.render_device(buf, params=p, camera=cam)
        torch.cuda.synchronize()
        mine = (index >= offset) & (index < offset + rows)  # image rows this part owns, in its packed order
        assert torch.equal(buf, rgb[mine]), part
        offset += rows
        del buf


def test_config4_like_field_small_full_parity():
    w = workloads.get("field2_small")
    fly, params, cam = workloads.make_flyscene(w)
    lights = [tuple(l) for l in w["lights"]]
    from oracle.oracle_py import OracleScene

    orc = OracleScene(arrays=workloads.mesh_arrays(w), lights=[l + (1.0, 1.0, 1.0) for l in lights], soft_points=0)
    got_rows = fly.local_rows()
    import torch

    rgb = torch.zeros((got_rows, w["width"], 3), dtype=torch.float32, device="cuda")
    fid = torch.zeros((got_rows, w["width"]), dtype=torch.int32, device="cuda")
    fly.render_device(rgb, fid, camera=cam)
    torch.cuda.synchronize()
    ref = orc.render(w["width"], w["height"], camera=oracle_camera(orc, cam), max_reflections=0, reject_nan=1)
    assert np.array_equal(fid.cpu().numpy(), ref["face_id"])
    assert np.abs(rgb.cpu().numpy() - ref["rgb"]).max() <= PIXEL_TOL
    st, c = fly.stats(), ref["counters"]
    assert (st.rays_primary, st.rays_shadow, st.rays_bounce) == (c.rays_primary, c.rays_shadow, c.rays_bounce)


def test_config5_like_mirrors_many_lights_full_parity():
    w = workloads.get("field2_mirrors_small")
    fly, params, cam = workloads.make_flyscene(w)
    from oracle.oracle_py import OracleScene
    import torch

    orc = OracleScene(arrays=workloads.mesh_arrays(w), lights=[tuple(l) + (1.0, 1.0, 1.0) for l in w["lights"]], soft_points=0)
    rgb = torch.zeros((w["height"], w["width"], 3), dtype=torch.float32, device="cuda")
    fid = torch.zeros((w["height"], w["width"]), dtype=torch.int32, device="cuda")
    fly.render_device(rgb, fid, camera=cam)
    torch.cuda.synchronize()
    ref = orc.render(w["width"], w["height"], camera=oracle_camera(orc, cam), max_reflections=4, reject_nan=1)
    assert np.array_equal(fid.cpu().numpy(), ref["face_id"])
    st, c = fly.stats(), ref["counters"]
    assert (st.rays_primary, st.rays_shadow, st.rays_bounce) == (c.rays_primary, c.rays_shadow, c.rays_bounce)
    assert st.rays_bounce > 1000
    assert np.abs(rgb.cpu().numpy() - ref["rgb"]).max() <= PIXEL_TOL


def test_config4_field_4k_invariants():
    """1.1 M triangles at 3840x2160: the frame split into 8 row-tile parts equals the single-part frame, and
    the ray bookkeeping is consistent; a 1/60 sample of rows is checked against the oracle."""
    import torch

    fly, rgb, fid = _rows_sample_parity("field_4k", 60)
    w = workloads.get("field_4k")
    cam = workloads.look_at(w["camera"]["eye"], w["camera"]["target"], w["width"], w["height"])
    base = fly.params_copy(row_tile=16)
    parts = []
    for part in range(8):
        p = tiles.part_params(base, 8, part)
        rows = fly.local_rows(p)
        buf = torch.zeros((rows, w["width"], 3), dtype=torch.float32, device="cuda")
        fly.render_device(buf, params=p, camera=cam)
        parts.append(buf)
    torch.cuda.synchronize()
    img = torch.cat(parts, 0).index_select(0, tiles.row_permutation(base, 8, "cuda"))
    assert torch.equal(img, rgb)


@pytest.mark.gpu
def test_gather_pipeline_rehearsal_on_streams():
    """The render loop of a several-GPU run on one GPU: frames on two streams, a device copy on a third stream where
    the collective would be, the same waits between them; every frame comes back, in order."""
    import torch

    from raytracer_amd import tiles

    fly, _ = make_pair("NthTest", width=96, height=64)
    p = fly.params_copy(row_tile=16)
    pipe = tiles.GatherPipeline(p, 96, 3, torch.device("cuda", 0), dtype=torch.uint8, depth=2, rehearse=True)
    streams = [torch.cuda.current_stream(), torch.cuda.Stream()]
    rgb = [torch.zeros((64, 96, 3), dtype=torch.float32, device="cuda") for _ in range(2)]
    fly.render_device(rgb[0], params=p)
    torch.cuda.synchronize()
    want = np.minimum(255, (255.0 * rgb[0].cpu().numpy()).astype(np.int32)).astype(np.uint8)
    got = []
    for k in range(7):
        s = streams[k % 2]
        buf = pipe.acquire(s)
        fly.render_device(rgb[k % 2], params=p, stream=s)
        tiles.quantize_ppm(rgb[k % 2], buf, 64 * 96 * 3, stream=s)
        with torch.cuda.stream(s):
            buf[0, 0, 0] = k  # marks the frame
        img = pipe.submit(s)
        if img is not None:
            got.append(img)
    got += pipe.drain()
    torch.cuda.synchronize()
    assert len(got) == 7
    for k, img in enumerate(got):
        img = img.cpu().numpy()
        assert img[0, 0, 0] == k
        img[0, 0, 0] = want[0, 0, 0]
        assert np.array_equal(img, want)


def test_all_to_all_pipeline_rehearsal_on_streams():
    """The same loop through the batched exchange (one process: batches of one frame, a device copy on a side stream where
    the all-to-all would be): frames on three streams into the chunks of the send buffers, every frame back in order."""
    import torch

    from raytracer_amd import tiles

    fly, _ = make_pair("NthTest", width=96, height=64)
    p = fly.params_copy(row_tile=16)
    pipe = tiles.AllToAllPipeline(p, 96, 3, torch.device("cuda", 0), dtype=torch.float32, depth=3, rehearse=True)
    streams = [torch.cuda.current_stream(), torch.cuda.Stream(), torch.cuda.Stream()]
    ref = torch.zeros((64, 96, 3), dtype=torch.float32, device="cuda")
    fly.render_device(ref, params=p)
    torch.cuda.synchronize()
    want = ref.cpu().numpy()
    got = []
    for k in range(8):
        s = streams[k % 3]
        buf = pipe.acquire(s)
        fly.render_device(buf, params=p, stream=s)
        with torch.cuda.stream(s):
            buf[0, 0, 0] = k  # marks the frame
        img = pipe.submit(s)
        if img is not None:
            got.append(img)
    got += pipe.drain()
    torch.cuda.synchronize()
    assert len(got) == 8
    for k, img in enumerate(got):
        img = img.cpu().numpy()
        assert img[0, 0, 0] == k
        img[0, 0, 0] = want[0, 0, 0]
        assert np.array_equal(img.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("seed,size", [(0, (640, 480)), (1, (1920, 1080)), (2, (333, 777))])
def test_faces_smaller_than_a_pixel_bins_and_grids_against_the_tree(seed, size):
    """Dust: tens of thousands of faces of about a pixel and less, several per pixel.  Their projected boxes are
    widened by a sixteenth of a pixel only (and their cells in the shadow grids by a sliver of a cell), so this is
    where a candidate list that is too tight would lose a hit: screen bins and shadow grids against the tree walk,
    bit for bit, also with sub-samples and a row partition."""
    rng = np.random.default_rng(100 + seed)
    n = 40000
    centres = rng.uniform(-1.0, 1.0, (n, 1, 3)) * np.array([1.0, 1.0, 0.3])
    extent = rng.choice([0.0004, 0.0015, 0.004, 0.012], (n, 1, 1))
    tri = (centres + rng.uniform(-1.0, 1.0, (n, 3, 3)) * extent).astype(np.float32)
    verts = tri.reshape(-1, 3)
    faces = np.arange(3 * n, dtype=np.int32).reshape(n, 3)
    mats = [dict(ka=[0.1] * 3, kd=[0.2 + 0.2 * m, 0.9 - 0.2 * m, 0.5], ks=[0.0] * 3, shininess=8.0) for m in range(4)]
    arrays = (verts, None, faces, rng.integers(0, 4, n).astype(np.int32), mats)
    lights = [(-1.0, 1.0, 1.0), (0.7, 1.3, -0.4)]
    W, H = size
    kw = dict(arrays=arrays, lights=lights, soft_points=1, output="")
    a = R.Flyscene(primary_rays=L.PRIMARY_RAYS_BINS, **kw).initialize(W, H)
    b = R.Flyscene(primary_rays=L.PRIMARY_RAYS_TREE, shadow_rays=L.SHADOW_RAYS_TREE, **kw).initialize(W, H)
    d = R.Flyscene(primary_rays=L.PRIMARY_RAYS_BINS, builder=L.BUILDER_DEVICE_LBVH, **kw).initialize(W, H)
    c = R.Flyscene(**kw).initialize(W, H)  # the automatic choice: every face against the pixels of its own box
    assert a.info().shadow_grid_entries > 0 and d.info().shadow_grid_entries == a.info().shadow_grid_entries
    for params in (dict(), dict(ssaa_level=2), dict(row_tile=16, num_parts=4, part=3), dict(row_tile=5, num_parts=3, part=1)):
        if W * H > 10**6 and "ssaa_level" in params:
            continue
        for f in (a, b, c, d):
            f.set_params(max_reflections=0, **params)
        ra, rb, rc, rd = a.render(), b.render(), c.render(), d.render()
        assert ra["stats"].primary_path == 1 and rb["stats"].primary_path == 0 and rc["stats"].primary_path == 1
        assert (ra["face_id"] >= 0).sum() > 200
        for r in (ra, rc, rd):
            assert np.array_equal(r["face_id"], rb["face_id"]), (size, params)
            assert np.array_equal(r["rgb"].view(np.uint32), rb["rgb"].view(np.uint32)), (size, params)
            assert (r["stats"].rays_shadow, r["stats"].hits) == (rb["stats"].rays_shadow, rb["stats"].hits)


def test_ply_scene_renders_like_its_obj_and_p6_output_matches_the_device_bytes(tmp_path):
    """Formats either side of the path, on the GPU: a binary PLY holding toy.obj's vertices, normals and faces (with toy.mtl
    beside it) must render the frame of the OBJ -- checked against the oracle's frame of the OBJ -- and the binary PPM
    written from that frame must hold the bytes rtamd_quantize_ppm produces on the device (utils/ppmIO.hpp:188-209)."""
    import shutil
    import torch

    from test_formats import write_ply

    obj = R.Flyscene(model=scene_path("toy"), output="").initialize(16, 16, upload=False).arrays()
    v, n, f = obj["raw_vertices"].reshape(-1, 3), obj["raw_normals"].reshape(-1, 3), obj["face_vertex_ids"].reshape(-1, 3)
    write_ply(str(tmp_path / "toy.ply"), v, n, f, binary=True, colors=True)
    shutil.copy(scene_path("toy")[:-4] + ".mtl", tmp_path / "toy.mtl")
    W, H = 128, 96
    out = tmp_path / "frame.p6.ppm"
    fly = R.Flyscene(model=str(tmp_path / "toy.ply"), output=str(out)).initialize(W, H)
    _, orc = make_pair("toy", width=W, height=H)  # the oracle reads the OBJ
    r = compare(fly, orc, W, H)
    assert_parity(r)
    assert r["got"]["stats"].rays_bounce > 0  # toy's Ks 0.485 reflects: every kernel of the frame took part
    # the file Flyscene::raytraceScene writes (binary form by the file name) against the device's quantisation of the frame
    fly.raytraceScene()
    raw = out.read_bytes()
    head = ("P6\n%d %d\n255\n" % (W, H)).encode()
    assert raw.startswith(head) and len(raw) == len(head) + W * H * 3
    dev = torch.zeros(W * H * 3, dtype=torch.uint8, device="cuda")
    tiles.quantize_ppm(torch.from_numpy(r["got"]["rgb"]).cuda(), dev)
    assert np.array_equal(np.frombuffer(raw[len(head):], np.uint8), dev.cpu().numpy())
