"""The C ABI library loads and exports every function the headers in include/ declare (no compute:
runs without a GPU), and the host-only entry points behave."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT, scene_path

import raytracer_amd as R
from raytracer_amd import _lib as L


def declared_functions():
    names = []
    for hdr in ("rtamd.h", "rtamd_flyscene.h"):
        text = open(os.path.join(ROOT, "include", hdr)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        for m in re.finditer(r"\b((?:rtamd|rtfly)_[a-z0-9_]+)\s*\(", text):
            names.append(m.group(1))
    return sorted(set(names))


def test_every_declared_symbol_is_exported_and_bound():
    L.lib()  # makes one HIP runtime global first: the library itself carries no DT_NEEDED entry for it
    lib = C.CDLL(L.LIB_PATH)
    declared = declared_functions()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), "librtamd.so does not export %s" % name
    assert sorted(L.SYMBOLS) == declared, "raytracer_amd/_lib.py and include/*.h disagree"


def test_struct_sizes_match_header_layout():
    # sizes implied by the field lists of include/rtamd.h
    assert C.sizeof(L.Material) == 40
    assert C.sizeof(L.Sphere) == 16
    assert C.sizeof(L.Camera) == (3 + 12 + 4 + 2) * 4
    assert C.sizeof(L.RenderParams) == (4 + 3 + 6 + 3 + 1 + 4) * 4
    assert C.sizeof(L.Stats) == 4 * 8 + 9 * 8 + 5 * 4 + 3 * 4 + 2 * 4 + 4 * 4


def test_defaults_are_the_reference_macros():
    p = R.default_params(1000, 1000)
    assert (p.ssaa_level, p.max_reflections) == (1, 5)             # flyscene.hpp:36,40
    assert p.reflect_threshold == np.float32(0.25) and p.bounce_offset == np.float32(0.01)
    assert p.t_epsilon == np.float32(0.00001)
    assert list(p.background) == [np.float32(0.4)] * 3 and list(p.background_deep) == [0.0] * 3
    cam = R.default_camera(1000, 500)
    assert list(cam.origin) == [0.0, 0.0, 2.0]                      # Flycamera::reset
    assert cam.aspect_ratio == 2.0 and cam.fovy == 60.0
    assert list(cam.viewport) == [0.0, 0.0, 1000.0, 500.0]


@pytest.mark.parametrize("height,tile,parts", [(1080, 16, 8), (1000, 16, 3), (37, 8, 5), (5, 16, 8), (64, 1, 2)])
def test_row_partition_is_a_bijection(height, tile, parts):
    lib = L.lib()
    seen = np.zeros(height, np.int32)
    for part in range(parts):
        p = R.default_params(8, height, row_tile=tile, num_parts=parts, part=part)
        n = lib.rtamd_local_rows(C.byref(p))
        rows = [lib.rtamd_global_row(C.byref(p), i) for i in range(n)]
        assert rows == sorted(rows)
        for y in rows:
            assert 0 <= y < height
            assert (y // tile) % parts == part
            seen[y] += 1
    assert (seen == 1).all()


def test_errors_are_reported_not_swallowed():
    lib = L.lib()
    out = C.c_void_p()
    assert lib.rtamd_scene_create(None, None, C.byref(out)) == L.ERR_INVALID
    assert b"null" in lib.rtamd_last_error()
    d = L.SceneDesc()
    d.num_faces = -1
    assert lib.rtamd_scene_create(C.byref(d), None, C.byref(out)) == L.ERR_INVALID
    with pytest.raises(L.RtamdError):
        R.Flyscene(model="/nonexistent/file.obj").initialize(8, 8, upload=False)
    p = R.default_params(8, 8, num_parts=2, part=5)
    assert lib.rtamd_local_rows(C.byref(p)) == 0


def test_write_ppm_format(tmp_path):
    rgb = np.zeros((2, 3, 3), np.float32)
    rgb[0, 0] = (1.0, 0.5, 0.25)
    rgb[1, 2] = (2.0, -0.0, 0.999)
    path = tmp_path / "o.ppm"
    R.write_ppm(str(path), rgb)
    text = path.read_text()
    # ppmIO.hpp:195-205: header, then one text line per row, every value followed by a blank
    assert text == "P3\n3 2\n255\n255 127 63 0 0 0 0 0 0 \n0 0 0 0 0 0 255 0 254 \n"


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "raytracer_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".h", ".hip")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.replace("the oracle", "").lower() or f == "workloads.py", (dirpath, f)


def test_scene_desc_layout_and_argument_checks_of_the_light_and_ppm_entry_points():
    # rtamd_scene_desc on LP64: num_unsampled_lights took the padding behind num_textures, so the struct kept its size and a
    # caller that zero-fills it (as every caller of the earlier layout did) says "every light is sampled"
    assert C.sizeof(L.SceneDesc) == 168 and L.SceneDesc.num_unsampled_lights.offset == 164
    lib = L.lib()
    out = C.c_void_p()
    d = L.SceneDesc()
    d.num_lights, d.num_unsampled_lights = 1, 2
    assert lib.rtamd_scene_create(C.byref(d), None, C.byref(out)) == L.ERR_INVALID
    assert b"without shadow samples" in lib.rtamd_last_error()
    d.num_unsampled_lights = -1
    assert lib.rtamd_scene_create(C.byref(d), None, C.byref(out)) == L.ERR_INVALID
    assert lib.rtamd_scene_set_lights(None, None, None, 0) == L.ERR_INVALID
    cam, p = R.default_camera(8, 8), R.default_params(8, 8)
    pixels = (C.c_uint8 * (8 * 8 * 3))()
    assert lib.rtamd_render_ppm(None, C.byref(cam), C.byref(p), pixels, None, None) == L.ERR_INVALID
