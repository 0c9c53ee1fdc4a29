"""File formats either side of the path: PLY ingestion (binary little-endian and ascii) and binary PPM output.
The reference reads OBJ only on this path and previews PLY (utils/plyimporter.hpp:213-284); a PLY holding the
same vertices, normals and faces as an OBJ must flatten to bit-identical arrays.  No GPU."""
import os
import struct

import numpy as np
import pytest

from conftest import scene_path

import raytracer_amd as R
from raytracer_amd import _lib as L

KEYS = ["raw_vertices", "raw_normals", "vertices", "normals", "face_vertex_ids", "face_normals", "face_origin_distance", "light_samples",
        "model_matrix"]


def write_ply(path, v, n, f, binary=True, colors=True):
    props = ["property float x", "property float y", "property float z"]
    if n is not None:
        props += ["property float nx", "property float ny", "property float nz"]
    if colors:
        props += ["property uchar red", "property uchar green", "property uchar blue", "property uchar alpha"]
    header = ["ply", "format %s 1.0" % ("binary_little_endian" if binary else "ascii"), "comment synthetic",
              "element vertex %d" % len(v)] + props + ["element face %d" % len(f), "property list uchar int vertex_indices", "end_header"]
    with open(path, "wb") as out:
        out.write(("\n".join(header) + "\n").encode())
        for i in range(len(v)):
            row = list(v[i]) + (list(n[i]) if n is not None else [])
            if binary:
                out.write(struct.pack("<%df" % len(row), *row))
                if colors:
                    out.write(bytes([10, 20, 30, 255]))
            else:
                out.write((" ".join("%.9g" % x for x in row) + (" 10 20 30 255" if colors else "") + "\n").encode())
        for tri in f:
            if binary:
                out.write(struct.pack("<B3i", 3, *tri))
            else:
                out.write(("3 %d %d %d\n" % tuple(tri)).encode())


@pytest.mark.parametrize("binary", [True, False])
@pytest.mark.parametrize("with_normals", [True, False])
def test_ply_gives_the_same_arrays_as_obj(tmp_path, binary, with_normals):
    ref = R.Flyscene(model=scene_path("toy"), output="").initialize(16, 16, upload=False).arrays()
    v = ref["raw_vertices"].reshape(-1, 3)
    n = ref["raw_normals"].reshape(-1, 3) if with_normals else None
    f = ref["face_vertex_ids"].reshape(-1, 3)
    path = tmp_path / "toy.ply"
    write_ply(str(path), v, n, f, binary=binary, colors=binary)
    got = R.Flyscene(model=str(path), output="").initialize(16, 16, upload=False).arrays()
    for k in KEYS:
        if k in ("raw_normals", "normals") and not with_normals:
            continue  # toy.obj ships normals; without them both loaders recompute, compared below
        a, b = np.ascontiguousarray(got[k]).reshape(-1), np.ascontiguousarray(ref[k]).reshape(-1)
        assert a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32)), k
    if not with_normals:
        recomputed = R.Flyscene(arrays=(v, None, f, np.full(len(f), -1, np.int32), []), output="").initialize(16, 16, upload=False).arrays()
        assert np.array_equal(got["normals"].view(np.uint32), recomputed["normals"].view(np.uint32))
    # no <name>.mtl beside the file: no material (the reference shades such faces grey, flyscene.cpp:469-472)
    assert (got["face_material_ids"] == -1).all()


def test_ply_picks_up_a_material_file_beside_it(tmp_path):
    v = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    write_ply(str(tmp_path / "tri.ply"), v, None, np.array([[0, 1, 2]], np.int32))
    (tmp_path / "tri.mtl").write_text("newmtl m\nNs 20\nKa 0.1 0.2 0.3\nKd 0.4 0.5 0.6\nKs 0.7 0.8 0.9\n")
    a = R.Flyscene(model=str(tmp_path / "tri.ply"), output="").initialize(8, 8, upload=False).arrays()
    assert a["face_material_ids"].tolist() == [0]
    assert np.allclose(a["materials"].reshape(-1, 10)[0], [0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 20])


def test_ply_errors_are_reported(tmp_path):
    (tmp_path / "bad.ply").write_bytes(b"ply\nformat binary_big_endian 1.0\nelement vertex 0\nend_header\n")
    with pytest.raises(L.RtamdError):
        R.Flyscene(model=str(tmp_path / "bad.ply"), output="").initialize(8, 8, upload=False)
    v = np.zeros((3, 3), np.float32)
    write_ply(str(tmp_path / "oob.ply"), v, None, np.array([[0, 1, 7]], np.int32))
    with pytest.raises(L.RtamdError):
        R.Flyscene(model=str(tmp_path / "oob.ply"), output="").initialize(8, 8, upload=False)


def test_binary_ppm_holds_the_same_bytes_as_the_text_form(tmp_path):
    rng = np.random.RandomState(3)
    rgb = rng.uniform(-0.2, 1.3, (5, 7, 3)).astype(np.float32)
    R.write_ppm(str(tmp_path / "a.ppm"), rgb)
    R.write_ppm(str(tmp_path / "b.pnm"), rgb, binary=True)
    text = (tmp_path / "a.ppm").read_text().split()
    assert text[:4] == ["P3", "7", "5", "255"]
    raw = (tmp_path / "b.pnm").read_bytes()
    assert raw.startswith(b"P6\n7 5\n255\n") and len(raw) == 11 + 5 * 7 * 3
    # ppmIO.hpp:203 writes min(255, (int)(255*c)) unclamped below; the binary form clamps negatives to 0
    assert list(raw[11:]) == [max(int(x), 0) for x in text[4:]]


@pytest.mark.parametrize("name", ["reflect_test_64", "NthTest_256", "bunny_256", "textured_quad_64"])
def test_product_ppm_writer_gives_the_golden_file_bytes(name, tmp_path):
    """The host writer of the product (ASCII P3, utils/ppmIO.hpp:188-209) turns the golden float frame into the
    same bytes as the oracle's writer did when the fixture was made."""
    import hashlib

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz"))
    R.write_ppm(str(tmp_path / "result.ppm"), g["rgb"])
    assert hashlib.sha256((tmp_path / "result.ppm").read_bytes()).hexdigest() == str(g["ppm_sha256"])
