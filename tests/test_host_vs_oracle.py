"""Host side of the C ABI (OBJ/MTL loader, model matrix, precomputeData, precomputeLights, replay of
splitBox's face order) against the CPU oracle, bit for bit.  Two independent implementations of the
reference's initialisation (src/flyscene.cpp:18-250, utils/objimporter.hpp, utils/mtlIO.hpp) must
produce identical arrays.  No GPU."""
import ctypes as C

import numpy as np
import pytest

from conftest import scene_path
from oracle.oracle_py import OracleScene

import raytracer_amd as R
from raytracer_amd import _lib as L

SCENES = ["reflect_test", "reflect_test_2", "shadowmodel", "NthTest", "glass", "toy", "toy_on_plane", "dodgeColorTest", "bunny",
          "textured_quad"]
KEYS = ["raw_vertices", "raw_normals", "vertices", "normals", "face_vertex_ids", "face_normals", "face_origin_distance",
        "face_material_ids", "materials", "light_samples", "model_matrix"]


def same_bits(a, b):
    a, b = np.ascontiguousarray(a).reshape(-1), np.ascontiguousarray(b).reshape(-1)
    return a.shape == b.shape and a.dtype == b.dtype and np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.parametrize("name", SCENES)
def test_flattened_scene_is_bit_identical(name):
    fly = R.Flyscene(model=scene_path(name), output="").initialize(64, 64, upload=False)
    got = fly.arrays()
    ref = OracleScene(scene_path(name)).arrays()
    for k in KEYS:
        assert same_bits(got[k], ref[k]), "%s: %s differs" % (name, k)


def test_texture_inputs_are_bit_identical(tmp_path):
    """`vt` / texture ids of `f` records / map_kd / ASCII PPM with flipped rows (objimporter.hpp:51-79,222-256;
    mtlIO.hpp:122-125; ppmIO.hpp:36-86): host loader vs oracle loader."""
    from raytracer_amd.flyscene import texture_arrays

    fly = R.Flyscene(model=scene_path("textured_quad"), output="").initialize(16, 16, upload=False)
    got, ref = texture_arrays(fly), OracleScene(scene_path("textured_quad")).texture_arrays()
    assert same_bits(got["face_texcoords"], ref["face_texcoords"]) and got["face_texcoords"].shape == (5, 3, 2)
    assert got["material_textures"].tolist() == ref["material_textures"].tolist() == [0, 1, -1]
    assert len(got["images"]) == len(ref["images"]) == 2
    for a, b in zip(got["images"], ref["images"]):
        assert same_bits(a, b)
    # checker.ppm: 8x4, first value of the file's LAST row comes first (40/255, 180/255, 255/255)
    assert got["images"][0].shape == (4, 8, 3)
    assert got["images"][0][0, 0].tolist() == [np.float32(40) / np.float32(255), np.float32(180) / np.float32(255), np.float32(1)]
    # faces of a textured group whose records carry no texture id keep the default (-1,-1)
    assert (got["face_texcoords"][3] == -1).all()
    # a scene without `vt` has no texture coordinates at all, and a missing image is an error, not a silent skip
    assert texture_arrays(R.Flyscene(model=scene_path("toy"), output="").initialize(8, 8, upload=False))["face_texcoords"].size == 0
    import shutil
    for f in ("textured_quad.obj", "textured_quad.mtl", "checker.ppm"):
        shutil.copy(scene_path(f[:-4])[:-4] + f[-4:], tmp_path / f)
    with pytest.raises(L.RtamdError):
        R.Flyscene(model=str(tmp_path / "textured_quad.obj"), output="").initialize(8, 8, upload=False)


def test_face_count_and_order_follow_the_file():
    a = R.Flyscene(model=scene_path("toy"), output="").initialize(8, 8, upload=False).arrays()
    assert a["face_vertex_ids"].shape == (12134, 3) and a["vertices"].shape == (6069, 3)
    # first record of toy.obj: f 228//228 3//3 242//242
    assert a["face_vertex_ids"][0].tolist() == [227, 2, 241]
    # unit cube: largest extent exactly 1 up to rounding, centred
    ext = a["vertices"].max(0) - a["vertices"].min(0)
    assert abs(ext.max() - 1.0) < 1e-6 and np.abs(a["vertices"].max(0) + a["vertices"].min(0)).max() < 1e-6


def test_light_samples_follow_precompute_lights():
    fly = R.Flyscene(model=scene_path("glass"), lights=[(-1, 1, 1), (0.5, 2.0, -1.0)], output="").initialize(8, 8, upload=False)
    s = fly.arrays()["light_samples"].reshape(2, 13, 3)
    assert s[0, 0].tolist() == [-1.0, 1.0, 1.0] and s[1, 0].tolist() == [0.5, 2.0, -1.0]
    r = np.linalg.norm(s[:, 1:] - s[:, :1], axis=-1)
    assert np.allclose(r, 0.3, atol=1e-6)  # SOFTSHADOWRADIUS
    ref = OracleScene(scene_path("glass"), lights=[(-1, 1, 1, 1, 1, 1), (0.5, 2.0, -1.0, 1, 1, 1)]).arrays()["light_samples"]
    assert same_bits(s, ref)
    hard = R.Flyscene(model=scene_path("glass"), soft_points=0, output="").initialize(8, 8, upload=False).arrays()["light_samples"]
    assert hard.shape == (1, 3)


def _desc(a):
    d = L.SceneDesc()
    keep = [np.ascontiguousarray(a["vertices"]), np.ascontiguousarray(a["normals"]), np.ascontiguousarray(a["face_vertex_ids"]),
            np.ascontiguousarray(a["face_normals"]), np.ascontiguousarray(a["face_origin_distance"]),
            np.ascontiguousarray(a["face_material_ids"])]
    fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int32)
    d.vertices, d.normals = keep[0].ctypes.data_as(fp), keep[1].ctypes.data_as(fp)
    d.num_vertices = len(keep[0])
    d.face_vertex_ids = keep[2].ctypes.data_as(ip)
    d.face_normals, d.face_origin_distance = keep[3].ctypes.data_as(fp), keep[4].ctypes.data_as(fp)
    d.face_material_ids = keep[5].ctypes.data_as(ip)
    d.num_faces = len(keep[2])
    return d, keep


@pytest.mark.parametrize("name", SCENES)
def test_reference_face_rank_matches_split_box_replay(name):
    orc = OracleScene(scene_path(name))
    a = orc.arrays()
    d, keep = _desc(a)
    rank = np.full(len(a["face_vertex_ids"]), -1, np.int32)
    L.check(L.lib().rtamd_reference_face_rank(C.byref(d), rank.ctypes.data_as(C.POINTER(C.c_int32))), "rank")
    assert sorted(rank.tolist()) == list(range(len(rank)))
    assert np.array_equal(rank, orc.bvh_face_order())


def test_arrays_entry_point_matches_obj_path():
    obj = R.Flyscene(model=scene_path("toy"), output="").initialize(8, 8, upload=False).arrays()
    mats = [dict(ka=m[0:3], kd=m[3:6], ks=m[6:9], shininess=m[9]) for m in obj["materials"]]
    arr = R.Flyscene(arrays=(obj["raw_vertices"], obj["raw_normals"], obj["face_vertex_ids"], obj["face_material_ids"], mats),
                     output="").initialize(8, 8, upload=False).arrays()
    for k in KEYS:
        assert same_bits(obj[k], arr[k]), k
    # without normals they are recomputed from the faces (objimporter.hpp:81-106), as in the oracle
    arr2 = R.Flyscene(arrays=(obj["raw_vertices"], None, obj["face_vertex_ids"], obj["face_material_ids"], mats),
                      output="").initialize(8, 8, upload=False).arrays()
    ref2 = OracleScene(arrays=(obj["raw_vertices"], None, obj["face_vertex_ids"], obj["face_material_ids"], mats)).arrays()
    assert same_bits(arr2["normals"], ref2["normals"]) and not same_bits(arr2["normals"], obj["normals"])


def test_camera_mirror():
    fly = R.Flyscene(model=scene_path("glass"), output="").initialize(640, 480, upload=False)
    cam = fly.getCamera()
    assert list(cam.origin) == [0.0, 0.0, 2.0] and cam.aspect_ratio == np.float32(640) / np.float32(480)
    fly.camera_translate(0.0, 0.0, 4.0)  # dz * speed(0.05) towards the scene
    cam = fly.getCamera()
    assert np.allclose(list(cam.origin), [0.0, 0.0, 1.8], atol=1e-6)


def test_add_light_appends_to_the_list_and_leaves_the_samples_alone():
    """Flyscene::addLight (flyscene.hpp:76) is `lights.push_back(getCenter())`; precomputeLights ran once, in initialize
    (flyscene.cpp:117).  Host: the sample sets stay those of initialize.  Oracle: the extra light changes the shading
    (shadeOffFace loops over `lights`, :505) but not the shadow rays (getLightIntensity loops over the sample sets, :880)."""
    fly = R.Flyscene(model=scene_path("glass"), output="").initialize(8, 8, upload=False)
    before = fly.arrays()["light_samples"].copy()
    fly.camera_translate(3.0, 1.0, 5.0)
    fly.addLight()
    fly.addLight()
    assert same_bits(fly.arrays()["light_samples"], before) and before.shape == (13, 3)
    orc = OracleScene(scene_path("glass"))
    plain = orc.render(48, 48)
    orc.add_light(tuple(fly.getCamera().origin))
    lit = orc.render(48, 48)
    assert orc.lib.orc_num_lights(orc.h) == 1 and orc.lib.orc_num_shading_lights(orc.h) == 2
    assert lit["counters"].rays_shadow == plain["counters"].rays_shadow and np.array_equal(lit["light"], plain["light"])
    hit = plain["face_id"] >= 0
    assert (lit["rgb"][hit] >= plain["rgb"][hit]).all() and (lit["rgb"][hit] > plain["rgb"][hit] + 0.01).any()
    # a light that IS sampled casts shadow rays of its own
    two = OracleScene(scene_path("glass"), lights=[(-1, 1, 1, 1, 1, 1), (*fly.getCamera().origin, 1, 1, 1)]).render(48, 48)
    assert two["counters"].rays_shadow == 2 * plain["counters"].rays_shadow
