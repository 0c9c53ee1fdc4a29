"""The legs of the path that CAN be checked against the reference's own compiled code, bit for bit.  No GPU.

oracle/_ref/camera_pin and oracle/_ref/io_pin are built by oracle/Makefile from the reference's headers where they lie
(Tucano::Camera, camera.hpp; ImageImporter::writePPMImage, utils/ppmIO.hpp:188-209), with nothing replaced or stubbed.
They exist only where /root/reference does (this container); elsewhere these tests are skipped.  What they pin:

* camera: getCenter(), the inverse view matrix and screenToWorld() of every pixel corner -- against the product's
  GL-free Flycamera (rtfly_get_camera) and the oracle's restatement (orc_screen_points), for the reference's default
  pose at the benchmark sizes and for moved and turned cameras;
* image output: the bytes of result.ppm -- against the oracle's and the product's writers.

Not pinnable this way (they need OpenGL symbols at link time, see oracle/io_pin.cpp): loadPPMImageData, loadMTL,
Tucano::Flycamera, boundingBox.h, src/flyscene.cpp."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, scene_path
from oracle.oracle_py import OracleScene, screen_points
from oracle.oracle_py import write_ppm as oracle_write_ppm
from parity_util import oracle_camera

import raytracer_amd as R
from raytracer_amd import _lib as L
from raytracer_amd.flyscene import write_ppm as product_write_ppm

REF = "/root/reference/dependencies/tucano/tucano/camera.hpp"
pytestmark = pytest.mark.skipif(not os.path.exists(REF), reason="reference not present")


def pin(name):
    exe = os.path.join(ROOT, "oracle", "_ref", name)
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "pin"])
    return exe


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


CAMERAS = [
    # (width, height, every n-th row, operations in the order a user performs them)
    (1000, 1000, 1, []),                    # the reference's window, src/main.cpp:9-10
    (1920, 1080, 1, []),
    (3840, 2160, 3, []),
    (7680, 4320, 7, []),
    (640, 480, 1, [("t", 2.0, -1.0, 6.0)]),
    (333, 777, 1, [("t", 0.0, 0.0, 38.0), ("r", 0.3, -0.7)]),
    (800, 600, 1, [("r", 0.2, -0.3), ("t", 2.0, -1.0, 6.0), ("r", -0.45, 1.9), ("t", -3.5, 0.25, 1.0)]),
    (512, 512, 1, [("r", 5.9, 3.3), ("t", 10.0, 10.0, -20.0)]),
]


@pytest.mark.parametrize("width,height,ystep,ops", CAMERAS, ids=["%dx%d_%dops" % (c[0], c[1], len(c[3])) for c in CAMERAS])
def test_camera_matches_the_reference_camera(width, height, ystep, ops, tmp_path):
    out = str(tmp_path / "cam.bin")
    args = [pin("camera_pin"), out, str(width), str(height), str(ystep)]
    for op in ops:
        args += [op[0]] + [repr(float(v)) for v in op[1:]]
    subprocess.check_call(args)
    raw = np.fromfile(out, np.float32)
    rows = (height + ystep - 1) // ystep
    assert raw.size == 15 + rows * width * 3
    ref_origin, ref_inv, ref_points = raw[:3], raw[3:15], raw[15:].reshape(rows, width, 3)

    # the product's GL-free Flycamera, driven through the C ABI like the reference's by its key handlers
    fly = R.Flyscene(model=scene_path("glass"), output="").initialize(width, height, upload=False)
    for op in ops:
        if op[0] == "t":
            L.check(fly.lib.rtfly_camera_translate(fly.h, *op[1:]), "translate")
        else:
            L.check(fly.lib.rtfly_camera_rotate(fly.h, *op[1:]), "rotate")
    L.check(fly.lib.rtfly_camera_update_view(fly.h), "update_view")
    cam = fly.getCamera()
    # (a zero may carry either sign: the product adds +0.0 so that no negative zero reaches a ray origin)
    assert np.array_equal(bits(np.array(list(cam.origin)) + 0.0), bits(ref_origin + np.float32(0.0)))
    assert np.array_equal(bits(np.array(list(cam.inv_view), np.float32) + np.float32(0.0)), bits(ref_inv + np.float32(0.0)))
    assert list(cam.viewport) == [0.0, 0.0, float(width), float(height)] and cam.fovy == 60.0
    assert cam.aspect_ratio == np.float32(width) / np.float32(height)

    # the oracle's screenToWorld on that camera
    got = screen_points(oracle_camera(None, cam), width, height, ystep)
    assert np.array_equal(bits(got + np.float32(0.0)), bits(ref_points + np.float32(0.0)))
    if not ops:  # Flycamera::reset: centre (0, 0, 2), looking down -z
        assert ref_origin.tolist() == [0.0, 0.0, 2.0] and np.all(ref_points[..., 2] == 1.0)
        oc = OracleScene(scene_path("glass")).default_camera(width, height)
        assert list(oc.origin) == ref_origin.tolist() and np.array_equal(np.array(list(oc.inv_view), np.float32), ref_inv + np.float32(0.0))


def test_ppm_writer_matches_the_reference_writer(tmp_path):
    frame = OracleScene(scene_path("reflect_test")).render(64, 48)["rgb"]
    # values around every quantisation step, above 1, below 0 (the reference clamps above only), tiny and huge
    edge = np.array([0.0, 1.0, 0.999999, 1.0000001, 254.0 / 255.0, 254.5 / 255.0, 0.5, 127.0 / 255.0, 128.0 / 255.0, 2.5,
                     -0.3, -1e-9, 1e-30, 3.9215687e-3, 3.9215684e-3, 1e6], np.float32)
    steps = (np.arange(0, 256, dtype=np.float32)[:, None] / np.float32(255.0) + np.array([-1e-7, 0.0, 1e-7], np.float32)).reshape(-1)
    extra = np.resize(np.concatenate([edge, steps]), (16, 64, 3)).astype(np.float32)
    rgb = np.ascontiguousarray(np.concatenate([frame, extra], 0))
    h, w, _ = rgb.shape
    src = tmp_path / "in.bin"
    with open(src, "wb") as f:
        f.write(np.array([w, h], np.int32).tobytes())
        f.write(rgb.tobytes())
    subprocess.check_call([pin("io_pin"), str(src), str(tmp_path / "ref.ppm")])
    ref = (tmp_path / "ref.ppm").read_bytes()
    assert ref.startswith(b"P3\n64 64\n255\n")
    assert oracle_write_ppm(str(tmp_path / "orc.ppm"), rgb) == 0
    assert (tmp_path / "orc.ppm").read_bytes() == ref
    product_write_ppm(str(tmp_path / "got.ppm"), rgb)
    assert (tmp_path / "got.ppm").read_bytes() == ref
