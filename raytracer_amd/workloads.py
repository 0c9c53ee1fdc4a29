"""The five benchmark configurations as concrete inputs (synthetic where the reference's own files are
missing from its snapshot).

Every workload is a dict: mesh arrays (file coordinates, as an OBJ would give them), materials,
lights, shadow sampling, image size, bounce depth and camera.  The same dict feeds the GPU path
(`make_flyscene`) and, in tests and in the benchmark's CPU leg, the oracle.

Scenes:
  bunny        the Stanford bunny of the reference's resources/models/bunny.ply, 69 451 faces
               (tests/scenes/bunny.obj is that file converted by tools/ply_to_obj.py because the
               reference's default resources/models/bunny.obj is absent from its snapshot)
  bunny_sub2   the same mesh after two rounds of 1-to-4 midpoint subdivision (1 111 216 faces),
               standing in for the missing dragon.ply
  field        a 4x4 grid of bunnies (16 x 69 451 faces) on a two-triangle ground plane, each
               instance turned about the vertical axis by 2*pi*u_k, u_k = k-th 32-bit output of
               mt19937(1234) / 2^32; baked into one mesh (the reference has no instancing)
"""
import math
import os

import numpy as np

from . import _lib as L
from .flyscene import Flyscene

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUNNY_OBJ = os.path.join(_ROOT, "tests", "scenes", "bunny.obj")

BUNNY_MTL = {"ka": (0.0, 0.0, 0.0), "kd": (0.714057, 0.296761, 0.8), "ks": (0.1, 0.1, 0.1), "shininess": 32.999994}
DEFAULT_LIGHT = (-1.0, 1.0, 1.0)


def _mat(ka, kd, ks, shininess):
    return {"ka": tuple(ka), "kd": tuple(kd), "ks": tuple(ks), "shininess": float(shininess)}


_bunny_cache = {}


def load_bunny():
    """(vertices[nv,3], normals[nv,3], faces[nf,3]) of the bunny in file coordinates, read with the
    product's own OBJ loader (host only, no GPU)."""
    if "b" not in _bunny_cache:
        f = Flyscene(model=BUNNY_OBJ, output="").initialize(8, 8, upload=False)
        a = f.arrays()
        _bunny_cache["b"] = (a["raw_vertices"].copy(), a["raw_normals"].copy(), a["face_vertex_ids"].copy())
        f.close()
    return _bunny_cache["b"]


def subdivide(vertices, normals, faces, rounds):
    """1-to-4 midpoint subdivision; new vertices sit at edge midpoints (positions and normals averaged)."""
    v, n, f = vertices.astype(np.float32), normals.astype(np.float32), faces.astype(np.int64)
    for _ in range(rounds):
        e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]], axis=0)
        e.sort(axis=1)
        key = e[:, 0] * (len(v) + 1) + e[:, 1]
        uniq, inverse = np.unique(key, return_inverse=True)
        a, b = uniq // (len(v) + 1), uniq % (len(v) + 1)
        mid_v = ((v[a] + v[b]) * np.float32(0.5)).astype(np.float32)
        mid_n = (n[a] + n[b]).astype(np.float32)
        base = len(v)
        nf = len(f)
        m01, m12, m20 = base + inverse[:nf], base + inverse[nf:2 * nf], base + inverse[2 * nf:]
        v = np.concatenate([v, mid_v])
        n = np.concatenate([n, mid_n])
        f = np.concatenate([
            np.stack([f[:, 0], m01, m20], 1), np.stack([m01, f[:, 1], m12], 1),
            np.stack([m20, m12, f[:, 2]], 1), np.stack([m01, m12, m20], 1)])
    return v, n, f.astype(np.int32)


def mt19937_unit(seed, count):
    """First `count` 32-bit outputs of mt19937(seed), scaled to [0, 1)."""
    rs = np.random.RandomState(seed)
    raw = rs.randint(0, 2 ** 32, size=count, dtype=np.uint64)
    return raw.astype(np.float64) / 2.0 ** 32


def bunny_field(grid=4, seed=1234, mirror_alternate=False):
    """grid x grid bunnies standing on the plane y = 0, instance k = row-major (ix, iz)."""
    v, n, f = load_bunny()
    lo, hi = v.min(0), v.max(0)
    size = float((hi - lo).max())
    pitch = 1.25 * size
    c = (lo + hi) * 0.5
    vs, ns, fs, fm = [], [], [], []
    u = mt19937_unit(seed, grid * grid)
    for k in range(grid * grid):
        ix, iz = k % grid, k // grid
        ang = 2.0 * math.pi * u[k]
        ca, sa = np.float32(math.cos(ang)), np.float32(math.sin(ang))
        x, y, z = v[:, 0] - c[0], v[:, 1] - lo[1], v[:, 2] - c[2]
        p = np.stack([ca * x + sa * z + np.float32((ix - (grid - 1) / 2) * pitch), y,
                      -sa * x + ca * z + np.float32((iz - (grid - 1) / 2) * pitch)], 1).astype(np.float32)
        q = np.stack([ca * n[:, 0] + sa * n[:, 2], n[:, 1], -sa * n[:, 0] + ca * n[:, 2]], 1).astype(np.float32)
        fs.append(f + k * len(v))
        vs.append(p)
        ns.append(q)
        fm.append(np.full(len(f), 1 if (mirror_alternate and k % 2 == 1) else 0, np.int32))
    nv = grid * grid * len(v)
    half = np.float32(grid * pitch * 0.6)
    ground_v = np.array([[-half, 0, -half], [half, 0, -half], [half, 0, half], [-half, 0, half]], np.float32)
    ground_n = np.tile(np.array([[0, 1, 0]], np.float32), (4, 1))
    ground_f = np.array([[nv, nv + 2, nv + 1], [nv, nv + 3, nv + 2]], np.int32)
    vs.append(ground_v)
    ns.append(ground_n)
    fs.append(ground_f)
    fm.append(np.full(2, 2, np.int32))
    return (np.concatenate(vs), np.concatenate(ns), np.concatenate(fs).astype(np.int32), np.concatenate(fm))


def look_at(eye, target, width, height, fovy=60.0, up=(0.0, 1.0, 0.0)):
    """rtamd camera looking from `eye` to `target` (inverse view = camera-to-world)."""
    eye, target, up = (np.asarray(a, np.float64) for a in (eye, target, up))
    fwd = target - eye
    fwd /= np.linalg.norm(fwd)
    right = np.cross(fwd, up)
    right /= np.linalg.norm(right)
    cup = np.cross(right, fwd)
    cam = L.Camera()
    # camera looks down -z: columns of the linear part are right, up, -forward
    m = np.zeros((3, 4), np.float32)
    m[:, 0], m[:, 1], m[:, 2], m[:, 3] = right, cup, -fwd, eye
    cam.inv_view[:] = m.reshape(-1).tolist()
    cam.origin[:] = np.asarray(eye, np.float32).tolist()
    cam.viewport[:] = [0.0, 0.0, float(width), float(height)]
    cam.aspect_ratio = np.float32(width) / np.float32(height)
    cam.fovy = fovy
    return cam


def get(name):
    """Workload description by name."""
    ring = [(1.5 * math.cos(2 * math.pi * k / 8), 1.0, 1.5 * math.sin(2 * math.pi * k / 8)) for k in range(8)]
    field_cam = dict(eye=(0.0, 0.42, 0.72), target=(0.0, -0.06, 0.0))
    table = {
        # config 1: the reference's defaults (src/main.cpp:9-10, src/flyscene.hpp:22-41, src/flyscene.cpp:9)
        "bunny_1000": dict(scene="bunny", width=1000, height=1000, lights=[DEFAULT_LIGHT], soft_points=12, max_reflections=5),
        # config 2: 1920x1080, shadows + one bounce; the stock Ks 0.1 never bounces (threshold 0.25), so
        # the specular coefficient is raised to 0.5 to exercise the bounce path
        "bunny_1080p": dict(scene="bunny", ks=0.5, width=1920, height=1080, lights=[DEFAULT_LIGHT], soft_points=12, max_reflections=1),
        "bunny_1080p_stock": dict(scene="bunny", width=1920, height=1080, lights=[DEFAULT_LIGHT], soft_points=12, max_reflections=1),
        # config 3: 4K, two point lights, hard shadows
        "bunny_4k": dict(scene="bunny", width=3840, height=2160, lights=[DEFAULT_LIGHT, (1.0, 1.0, 1.0)], soft_points=0, max_reflections=0),
        "bunny_sub2_4k": dict(scene="bunny_sub2", width=3840, height=2160, lights=[DEFAULT_LIGHT, (1.0, 1.0, 1.0)], soft_points=0,
                              max_reflections=0),
        # config 4: 1.1 M-triangle field at 4K, one light, hard shadows
        "field_4k": dict(scene="field", width=3840, height=2160, lights=[DEFAULT_LIGHT], soft_points=0, max_reflections=0, camera=field_cam),
        # config 5: mirrors + 8 lights, depth 4, 8K
        "field_mirrors_8k": dict(scene="field_mirrors", width=7680, height=4320, lights=ring, soft_points=0, max_reflections=4,
                                 camera=field_cam),
        # small stand-ins used by tests
        "field2_small": dict(scene="field2", width=256, height=144, lights=[DEFAULT_LIGHT], soft_points=0, max_reflections=0, camera=field_cam),
        "field2_mirrors_small": dict(scene="field2_mirrors", width=256, height=144, lights=ring, soft_points=0, max_reflections=4,
                                     camera=field_cam),
    }
    if name not in table:
        raise KeyError("unknown workload %r; known: %s" % (name, ", ".join(sorted(table))))
    w = dict(table[name])
    w["name"] = name
    return w


def mesh_arrays(w):
    """(vertices, normals, faces, face_material, materials) of a workload."""
    scene = w["scene"]
    ks = w.get("ks")
    bunny_mat = dict(BUNNY_MTL)
    if ks is not None:
        bunny_mat["ks"] = (ks, ks, ks)
    if scene in ("bunny", "bunny_sub2"):
        v, n, f = load_bunny()
        if scene == "bunny_sub2":
            v, n, f = subdivide(v, n, f, 2)
        return v, n, f, np.zeros(len(f), np.int32), [bunny_mat]
    if scene in ("field", "field_mirrors", "field2", "field2_mirrors"):
        mirrors = scene.endswith("mirrors")
        grid = 2 if scene.startswith("field2") else 4
        v, n, f, fm = bunny_field(grid, 1234, mirror_alternate=mirrors)
        mats = [bunny_mat,
                _mat((0.0, 0.0, 0.0), (0.3, 0.3, 0.35), (0.8, 0.8, 0.8), 225.0),                       # mirror bunnies
                _mat((0.05, 0.05, 0.05), (0.5, 0.5, 0.5), (0.8, 0.8, 0.8) if mirrors else (0.1, 0.1, 0.1), 225.0)]  # ground
        return v, n, f, fm, mats
    raise KeyError(scene)


def make_flyscene(w, upload=True, **build):
    """Flyscene initialised for the workload; returns (flyscene, params, camera)."""
    arrays = mesh_arrays(w)
    fly = Flyscene(arrays=arrays, lights=w["lights"], hard_shadow=True, soft_points=w["soft_points"], soft_radius=0.3, output="", **build)
    fly.initialize(w["width"], w["height"], upload=upload)
    fly.set_params(max_reflections=w["max_reflections"])
    cam = fly.getCamera()
    if "camera" in w:
        cam = look_at(w["camera"]["eye"], w["camera"]["target"], w["width"], w["height"])
    return fly, fly.params, cam
