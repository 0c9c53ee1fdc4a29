"""The CPU oracle against the committed vectors of tests/golden (made by tools/make_golden.py), and its
internal consistency.  PARITY UNPINNED: the reference has no tests or golden images and cannot be run,
so these vectors pin the oracle to itself; what ties the oracle to the reference is the Eigen pin
(oracle/eigen_pin.cpp) and the line-by-line citations in oracle/oracle.c.  No GPU."""
import glob
import hashlib
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, scene_path
from oracle.oracle_py import OracleScene, screen_ray, write_ppm

FILES = sorted(glob.glob(os.path.join(GOLDEN, "*.npz")))


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(p)[:-4] for p in FILES])
def test_oracle_reproduces_golden(path, tmp_path):
    g = np.load(path)
    name, size = os.path.basename(path)[:-4].rsplit("_", 1)
    s = OracleScene(scene_path(name))
    r = s.render(int(size), int(size))
    assert np.array_equal(r["face_id"], g["face_id"])
    assert np.array_equal(r["rgb"].view(np.uint32), g["rgb"].view(np.uint32))
    assert np.array_equal(r["nan_mask"].astype(np.uint8), g["nan_mask"])
    c = r["counters"]
    assert [c.rays_primary, c.rays_shadow, c.rays_bounce] == g["rays"].tolist()
    assert [s.bvh_stats()[k] for k in ("nodes", "leaves", "depth")] == g["bvh_stats"].tolist()
    assert digest(s.bvh_face_order()) == str(g["face_order_sha256"])
    a = s.arrays()
    for k in ("vertices", "normals", "face_normals", "face_origin_distance", "face_vertex_ids", "face_material_ids", "light_samples"):
        assert digest(a[k]) == str(g["sha256_" + k]), k
    write_ppm(str(tmp_path / "result.ppm"), r["rgb"])  # the file the reference leaves behind
    assert hashlib.sha256((tmp_path / "result.ppm").read_bytes()).hexdigest() == str(g["ppm_sha256"])
    if "t" in g.files:
        assert np.array_equal(r["t"].view(np.uint32), g["t"].view(np.uint32))
        assert np.array_equal(r["light"].view(np.uint32), g["light"].view(np.uint32))
        boxes, info = s.bvh_dump()
        assert np.array_equal(boxes.view(np.uint32), g["bvh_boxes"].view(np.uint32)) and np.array_equal(info, g["bvh_info"])


def test_thread_count_does_not_change_the_image():
    s = OracleScene(scene_path("toy"))
    a = s.render(48, 48, num_threads=1)
    b = s.render(48, 48, num_threads=5)
    assert np.array_equal(a["rgb"].view(np.uint32), b["rgb"].view(np.uint32)) and np.array_equal(a["face_id"], b["face_id"])


def test_row_sampling_renders_exactly_those_rows():
    s = OracleScene(scene_path("reflect_test"))
    full = s.render(40, 40)
    part = s.render(40, 40, row_offset=3, row_step=7)
    rows = list(range(3, 40, 7))
    assert np.array_equal(part["rgb"][rows].view(np.uint32), full["rgb"][rows].view(np.uint32))
    assert part["counters"].rays_primary == len(rows) * 40


def test_bvh_is_the_reference_tree_shape():
    # splitBox: leaves have < 10 faces unless a split failed the 0.8 rule; faces are partitioned
    s = OracleScene(scene_path("bunny"))
    boxes, info = s.bvh_dump()
    leaves = info[info[:, 0] == 1]
    assert leaves[:, 1].sum() == s.nf
    assert sorted(s.bvh_face_order().tolist()) == list(range(s.nf))
    st = s.bvh_stats()
    assert st["nodes"] == len(info) and st["leaves"] == len(leaves) and st["nodes"] == 2 * st["leaves"] - 1


def test_primary_ray_is_at_the_pixel_corner():
    s = OracleScene(scene_path("glass"))
    cam = s.default_camera(100, 100)
    d = screen_ray(cam, 50.0, 50.0)
    assert d.tolist() == [0.0, 0.0, -1.0]  # pixel (W/2, H/2) looks straight down -z: sample at the corner, not the centre
    d0 = screen_ray(cam, 0.0, 0.0)
    t = np.tan(np.radians(30.0))
    assert np.allclose(d0 / -d0[2], [-t, t, -1.0], atol=1e-6)


def test_shadow_ray_direction_is_the_sample_position():
    # One triangle above the origin region and a floor: a point is shadowed iff the ray from it along the
    # light's POSITION VECTOR (not light - point) hits something (flyscene.cpp:884).
    v = np.array([[-5, 0, -5], [5, 0, -5], [5, 0, 5], [-5, 0, 5], [0.9, 1.0, 0.9], [1.1, 1.0, 0.9], [1.0, 1.0, 1.1]], np.float32)
    f = np.array([[0, 2, 1], [0, 3, 2], [4, 5, 6]], np.int32)
    mats = [dict(ka=(0, 0, 0), kd=(1, 1, 1), ks=(0, 0, 0), shininess=1.0)]
    s = OracleScene(arrays=(v, None, f, np.zeros(3, np.int32), mats), lights=[(1.0, 1.0, 1.0, 1, 1, 1)], soft_points=0, normalize_model=False)
    # from the floor point (0,0,0) the position vector (1,1,1) passes through the small triangle at (1,1,1)
    hit, fid, t, _, _ = s.intersects([0, 0, 0], [1, 1, 1])
    assert hit and fid == 2 and abs(t - 1.0) < 1e-6
    # from (2,0,2) the true direction to the light would be (-1,1,-1) (free), but the reference shoots (1,1,1): also free
    assert not s.intersects([2, 0, 2], [1, 1, 1])[0]
    # from (-1,0,-1) along (1,1,1): reaches y=1 at (0,1,0): misses the triangle
    assert not s.intersects([-1, 0, -1], [1, 1, 1])[0]


def test_equal_distance_later_face_wins():
    # two coplanar triangles sharing the diagonal of a quad: on the diagonal both accept with the same t and
    # the face met later (higher index here: one leaf) is reported (flyscene.cpp:760,782-789)
    v = np.array([[-1, -1, 0], [1, -1, 0], [1, 1, 0], [-1, 1, 0]], np.float32)
    f = np.array([[0, 1, 2], [0, 2, 3]], np.int32)
    mats = [dict(ka=(0, 0, 0), kd=(1, 1, 1), ks=(0, 0, 0), shininess=1.0)]
    s = OracleScene(arrays=(v, None, f, np.zeros(2, np.int32), mats), normalize_model=False)
    hit, fid, t, _, _ = s.intersects([0.5, 0.5, 2], [0, 0, -1])
    assert hit and fid == 1 and t == 2.0
    assert s.intersects([0.5, -0.5, 2], [0, 0, -1])[1] == 0


def test_zero_over_zero_in_the_box_test_and_the_modes_that_settle_it():
    # boundingBox::boxIntersection divides (side - origin) by the direction (boundingBox.h:258-271).  A ray with direction.x == 0
    # that starts at x == 0 runs IN the plane of a box side at x == 0: 0 / 0.  With the reference's min / max a box that ENDS at
    # that plane is turned away (tout becomes -inf) and one that STARTS there is taken (everything compares false).
    mats = [dict(ka=(0, 0, 0), kd=(1, 1, 1), ks=(0, 0, 0), shininess=1.0)]
    ends = np.array([[-1, -1, 0], [0, -1, 0], [0, 1, 0]], np.float32)    # x in [-1, 0]: the ray along x = 0 runs in its side
    starts = np.array([[0, -1, 0], [1, -1, 0], [0, 1, 0]], np.float32)   # x in [0, 1]
    f = np.array([[0, 1, 2]], np.int32)
    o, d = [0.0, 0.0, 2.0], [0.0, 0.0, -1.0]
    for verts, as_reference in ((ends, False), (starts, True)):
        s = OracleScene(arrays=(verts, None, f, np.zeros(1, np.int32), mats), normalize_model=False)
        assert s.intersects(o, d)[0] == as_reference                      # the reference: depends on the side
        assert s.intersects(o, d, reject_nan=1)[:2] == (True, 0)          # exact arithmetic: the ray touches the box; the edge x = 0 is hit
        assert s.intersects(o, d, all_faces=1)[:2] == (True, 0)           # no box asked at all
        assert s.intersects([0.25 if as_reference else -0.25, 0.0, 2.0], d)[:2] == (True, 0)  # an ordinary ray: every mode alike
        # a frame flags the pixels whose rays met the 0 / 0 (here column W / 2 of an even-width image: its samples have n.x == 0)
        r = s.render(8, 5)
        assert r["nan_mask"][:, 4].any() and not r["nan_mask"][:, :4].any() and not r["nan_mask"][:, 5:].any()


def test_all_faces_mode_changes_nothing_on_the_bundled_scenes():
    # No box turns a ray away in this mode; on the reference's own scenes no box of the reference decides a pixel
    for name, size in (("toy", (48, 40)), ("reflect_test", (64, 64)), ("NthTest", (40, 48))):
        s = OracleScene(scene_path(name))
        a = s.render(*size, max_reflections=2)
        b = s.render(*size, max_reflections=2, all_faces=1)
        assert np.array_equal(a["face_id"], b["face_id"]) and np.array_equal(a["rgb"].view(np.uint32), b["rgb"].view(np.uint32))
        assert b["counters"].leaves_visited > a["counters"].leaves_visited


def test_ppm_writer(tmp_path):
    rgb = np.zeros((1, 2, 3), np.float32)
    rgb[0, 0] = (0.4, 0.4, 0.4)
    rgb[0, 1] = (1.5, 0.999, 0.0)
    write_ppm(str(tmp_path / "a.ppm"), rgb)
    assert (tmp_path / "a.ppm").read_text() == "P3\n2 1\n255\n102 102 102 255 254 0 \n"


@pytest.mark.skipif(not os.path.isdir("/root/reference/dependencies/eigen/include"), reason="reference not present")
def test_vector_primitives_match_vendored_eigen():
    exe = os.path.join(ROOT, "oracle", "_ref", "eigen_pin")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "pin"])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout
    assert "0 mismatches" in out.stdout
