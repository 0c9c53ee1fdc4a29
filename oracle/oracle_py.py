"""ctypes binding of the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE ONLY.

Importable from tests/, from __graft_entry__.smoke() and from bench.py's cpu_baseline leg; the
product package raytracer_amd never imports it.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


class Material(C.Structure):
    _fields_ = [("ka", C.c_float * 3), ("kd", C.c_float * 3), ("ks", C.c_float * 3), ("shininess", C.c_float)]


class Camera(C.Structure):
    _fields_ = [
        ("origin", C.c_float * 3),
        ("inv_view", C.c_float * 12),
        ("viewport", C.c_float * 4),
        ("aspect_ratio", C.c_float),
        ("fovy", C.c_float),
    ]


class Params(C.Structure):
    _fields_ = [
        ("width", C.c_int32),
        ("height", C.c_int32),
        ("ssaa_level", C.c_int32),
        ("max_reflections", C.c_int32),
        ("reflect_threshold", C.c_float),
        ("bounce_offset", C.c_float),
        ("t_epsilon", C.c_float),
        ("background", C.c_float * 3),
        ("background_deep", C.c_float * 3),
        ("num_threads", C.c_int32),
        ("row_offset", C.c_int32),
        ("row_step", C.c_int32),
        ("reject_nan", C.c_int32),
        ("all_faces", C.c_int32),
    ]


class Counters(C.Structure):
    _fields_ = [
        ("rays_primary", C.c_uint64),
        ("rays_shadow", C.c_uint64),
        ("rays_bounce", C.c_uint64),
        ("box_tests", C.c_uint64),
        ("leaves_visited", C.c_uint64),
        ("tri_plane", C.c_uint64),
        ("tri_edge", C.c_uint64),
        ("nan_accepts", C.c_uint64),
        ("seconds", C.c_double),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}

    @property
    def rays(self):
        return self.rays_primary + self.rays_shadow + self.rays_bounce


_libs = {}


def load(fast=False):
    """Load liboracle.so (canonical arithmetic) or liboracle_fast.so (reference Release flags)."""
    name = "liboracle_fast.so" if fast else "liboracle.so"
    if name in _libs:
        return _libs[name]
    path = os.path.join(_HERE, name)
    if not os.path.exists(path):
        raise RuntimeError("%s not built: run `make -C oracle`" % path)
    lib = C.CDLL(path)
    P = C.c_void_p
    fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int32)
    lib.orc_scene_load_obj.restype = P
    lib.orc_scene_load_obj.argtypes = [C.c_char_p]
    lib.orc_scene_from_arrays.restype = P
    lib.orc_scene_from_arrays.argtypes = [fp, fp, C.c_int32, ip, ip, C.c_int32, C.POINTER(Material), C.c_int32]
    lib.orc_scene_free.argtypes = [P]
    lib.orc_scene_precompute.argtypes = [P, C.c_int32]
    lib.orc_scene_set_lights.argtypes = [P, fp, C.c_int32, C.c_int32, C.c_int32, C.c_float]
    lib.orc_scene_add_light.argtypes = [P, fp]
    lib.orc_scene_set_spheres.argtypes = [P, fp, C.c_int32]
    lib.orc_scene_build_bvh.argtypes = [P]
    lib.orc_scene_set_texcoords.argtypes = [P, fp]
    lib.orc_scene_add_texture.restype = C.c_int32
    lib.orc_scene_add_texture.argtypes = [P, C.c_int32, C.c_int32, fp]
    lib.orc_scene_set_material_texture.argtypes = [P, C.c_int32, C.c_int32]
    lib.orc_face_texcoords.restype = fp
    lib.orc_face_texcoords.argtypes = [P]
    lib.orc_num_textures.restype = C.c_int32
    lib.orc_num_textures.argtypes = [P]
    lib.orc_texture_data.restype = fp
    lib.orc_texture_data.argtypes = [P, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.orc_material_textures.restype = ip
    lib.orc_material_textures.argtypes = [P]
    for fn in ("orc_num_vertices", "orc_num_faces", "orc_num_materials", "orc_num_lights", "orc_num_shading_lights", "orc_samples_per_light",
               "orc_bvh_node_count", "orc_bvh_leaf_count", "orc_bvh_depth"):
        getattr(lib, fn).restype = C.c_int32
        getattr(lib, fn).argtypes = [P]
    for fn in ("orc_raw_vertices", "orc_raw_normals", "orc_vertices", "orc_normals", "orc_face_normals",
               "orc_face_origin_distance", "orc_light_samples", "orc_model_matrix"):
        getattr(lib, fn).restype = fp
        getattr(lib, fn).argtypes = [P]
    for fn in ("orc_face_vertex_ids", "orc_face_material_ids"):
        getattr(lib, fn).restype = ip
        getattr(lib, fn).argtypes = [P]
    lib.orc_materials.restype = C.POINTER(Material)
    lib.orc_materials.argtypes = [P]
    lib.orc_bvh_face_order.argtypes = [P, ip]
    lib.orc_bvh_dump.argtypes = [P, fp, ip]
    lib.orc_camera_default.argtypes = [C.POINTER(Camera), C.c_int32, C.c_int32]
    lib.orc_params_default.argtypes = [C.POINTER(Params), C.c_int32, C.c_int32]
    lib.orc_render.restype = C.c_int32
    lib.orc_render.argtypes = [P, C.POINTER(Camera), C.POINTER(Params), fp, ip, fp, fp, ip, C.POINTER(Counters)]
    lib.orc_intersects.restype = C.c_int32
    lib.orc_intersects.argtypes = [P, fp, fp, C.POINTER(Params), ip, fp, fp, fp]
    lib.orc_screen_ray.argtypes = [C.POINTER(Camera), C.c_float, C.c_float, fp]
    lib.orc_screen_points.argtypes = [C.POINTER(Camera), C.c_int32, C.c_int32, C.c_int32, fp]
    lib.orc_write_ppm.restype = C.c_int32
    lib.orc_write_ppm.argtypes = [C.c_char_p, fp, C.c_int32, C.c_int32]
    _libs[name] = lib
    return lib


def _fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _iptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _arr(ptr, shape, dtype):
    n = int(np.prod(shape))
    if n == 0:
        return np.zeros(shape, dtype=dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True).reshape(shape)


DEFAULT_LIGHTS = ((-1.0, 1.0, 1.0, 1.0, 1.0, 1.0),)  # flyscene.cpp:9


class OracleScene:
    """One loaded scene: load -> precompute -> lights -> BVH, in the order of Flyscene::initialize
    (flyscene.cpp:18-134)."""

    def __init__(self, obj_path=None, arrays=None, lights=DEFAULT_LIGHTS, hard_shadow=True, soft_points=12,
                 soft_radius=0.3, spheres=(), normalize_model=True, fast=False, textures=None):
        self.lib = load(fast)
        if obj_path is not None:
            self.h = self.lib.orc_scene_load_obj(os.fsencode(obj_path))
            if not self.h:
                raise FileNotFoundError(obj_path)
        else:
            v, n, f, fm, mats = arrays
            v = np.ascontiguousarray(v, np.float32)
            f = np.ascontiguousarray(f, np.int32)
            fm = np.ascontiguousarray(fm, np.int32)
            nptr = None
            if n is not None:
                n = np.ascontiguousarray(n, np.float32)
                nptr = _fptr(n)
            marr = (Material * max(1, len(mats)))()
            for i, m in enumerate(mats):
                marr[i].ka[:] = m["ka"]
                marr[i].kd[:] = m["kd"]
                marr[i].ks[:] = m["ks"]
                marr[i].shininess = m["shininess"]
            self.h = self.lib.orc_scene_from_arrays(_fptr(v), nptr, len(v), _iptr(f), _iptr(fm), len(f), marr, len(mats))
            if textures is not None:
                tc = np.ascontiguousarray(textures["face_texcoords"], np.float32).reshape(len(f), 6)
                self.lib.orc_scene_set_texcoords(self.h, _fptr(tc))
                ids = []
                for im in textures["images"]:
                    im = np.ascontiguousarray(im, np.float32)
                    ids.append(self.lib.orc_scene_add_texture(self.h, im.shape[1], im.shape[0], _fptr(im)))
                for m, t in enumerate(np.asarray(textures["material_textures"], np.int32).reshape(-1)):
                    self.lib.orc_scene_set_material_texture(self.h, m, ids[t] if t >= 0 else -1)
        self.lib.orc_scene_precompute(self.h, 1 if normalize_model else 0)
        self.set_lights(lights, hard_shadow, soft_points, soft_radius)
        sp = np.ascontiguousarray(np.array(spheres, np.float32).reshape(-1, 4))
        self.lib.orc_scene_set_spheres(self.h, _fptr(sp), len(sp))
        self.lib.orc_scene_build_bvh(self.h)

    def set_lights(self, lights, hard_shadow=True, soft_points=12, soft_radius=0.3):
        pc = np.ascontiguousarray(np.array(lights, np.float32).reshape(-1, 6))
        self.lib.orc_scene_set_lights(self.h, _fptr(pc), len(pc), int(hard_shadow), int(soft_points), float(soft_radius))

    def add_light(self, position, color=(1.0, 1.0, 1.0)):
        """Flyscene::addLight (flyscene.hpp:76): a light without shadow sample points."""
        pc = np.ascontiguousarray(np.array([*position, *color], np.float32))
        self.lib.orc_scene_add_light(self.h, _fptr(pc))

    def close(self):
        if getattr(self, "h", None):
            self.lib.orc_scene_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- flattened arrays (PrecomputedData, defs.h:7-20) ----
    @property
    def nv(self):
        return self.lib.orc_num_vertices(self.h)

    @property
    def nf(self):
        return self.lib.orc_num_faces(self.h)

    def arrays(self):
        L = self.lib
        nv, nf, nm = self.nv, self.nf, L.orc_num_materials(self.h)
        nl, spl = L.orc_num_lights(self.h), L.orc_samples_per_light(self.h)
        mats = L.orc_materials(self.h)
        return {
            "raw_vertices": _arr(L.orc_raw_vertices(self.h), (nv, 3), np.float32),
            "raw_normals": _arr(L.orc_raw_normals(self.h), (nv, 3), np.float32),
            "vertices": _arr(L.orc_vertices(self.h), (nv, 3), np.float32),
            "normals": _arr(L.orc_normals(self.h), (nv, 3), np.float32),
            "face_vertex_ids": _arr(L.orc_face_vertex_ids(self.h), (nf, 3), np.int32),
            "face_normals": _arr(L.orc_face_normals(self.h), (nf, 3), np.float32),
            "face_origin_distance": _arr(L.orc_face_origin_distance(self.h), (nf,), np.float32),
            "face_material_ids": _arr(L.orc_face_material_ids(self.h), (nf,), np.int32),
            "materials": np.array([[*mats[i].ka, *mats[i].kd, *mats[i].ks, mats[i].shininess] for i in range(nm)],
                                  np.float32).reshape(nm, 10),
            "light_samples": _arr(L.orc_light_samples(self.h), (nl, spl, 3), np.float32),
            "model_matrix": _arr(L.orc_model_matrix(self.h), (3, 4), np.float32),
        }

    def texture_arrays(self):
        """Face::texcoord, material -> texture, texturedatas (flyscene.cpp:52-71,476-481,502)."""
        L = self.lib
        nf, nm = self.nf, L.orc_num_materials(self.h)
        tc = L.orc_face_texcoords(self.h)
        mt = L.orc_material_textures(self.h)
        out = {"face_texcoords": _arr(tc, (nf, 3, 2), np.float32) if tc else np.zeros((0, 3, 2), np.float32),
               "material_textures": _arr(mt, (nm,), np.int32) if mt else np.full(nm, -1, np.int32), "images": []}
        for k in range(L.orc_num_textures(self.h)):
            w, h = C.c_int32(), C.c_int32()
            p = L.orc_texture_data(self.h, k, C.byref(w), C.byref(h))
            out["images"].append(_arr(p, (h.value, w.value, 3), np.float32))
        return out

    def bvh_stats(self):
        L = self.lib
        return {"nodes": L.orc_bvh_node_count(self.h), "leaves": L.orc_bvh_leaf_count(self.h),
                "depth": L.orc_bvh_depth(self.h)}

    def bvh_face_order(self):
        out = np.zeros(self.nf, np.int32)
        self.lib.orc_bvh_face_order(self.h, _iptr(out))
        return out

    def bvh_dump(self):
        n = self.lib.orc_bvh_node_count(self.h)
        boxes = np.zeros((n, 6), np.float32)
        info = np.zeros((n, 2), np.int32)
        self.lib.orc_bvh_dump(self.h, _fptr(boxes), _iptr(info))
        return boxes, info

    # ---- rendering ----
    def default_camera(self, w, h):
        cam = Camera()
        self.lib.orc_camera_default(C.byref(cam), w, h)
        return cam

    def default_params(self, w, h, **kw):
        p = Params()
        self.lib.orc_params_default(C.byref(p), w, h)
        for k, v in kw.items():
            if isinstance(v, (tuple, list)):
                getattr(p, k)[:] = v
            else:
                setattr(p, k, v)
        return p

    def render(self, w, h, camera=None, planes=True, **kw):
        cam = camera if camera is not None else self.default_camera(w, h)
        p = self.default_params(w, h, **kw)
        rgb = np.zeros((h, w, 3), np.float32)
        fid = np.full((h, w), -3, np.int32)
        t = np.zeros((h, w), np.float32)
        li = np.zeros((h, w), np.float32)
        nan = np.zeros((h, w), np.int32)
        cnt = Counters()
        rc = self.lib.orc_render(self.h, C.byref(cam), C.byref(p), _fptr(rgb), _iptr(fid) if planes else None,
                                 _fptr(t) if planes else None, _fptr(li) if planes else None, _iptr(nan) if planes else None, C.byref(cnt))
        if rc != 0:
            raise RuntimeError("orc_render failed: %d" % rc)
        return {"rgb": rgb, "face_id": fid, "t": t, "light": li, "nan_mask": nan, "counters": cnt}

    def intersects(self, origin, direction, **kw):
        p = self.default_params(1, 1, **kw)
        o = np.ascontiguousarray(origin, np.float32)
        d = np.ascontiguousarray(direction, np.float32)
        fid = C.c_int32(-2)
        t = C.c_float(0)
        hp = np.zeros(3, np.float32)
        rf = np.zeros(3, np.float32)
        hit = self.lib.orc_intersects(self.h, _fptr(o), _fptr(d), C.byref(p), C.byref(fid), C.byref(t), _fptr(hp), _fptr(rf))
        return bool(hit), fid.value, t.value, hp, rf


def screen_ray(camera, px, py, fast=False):
    d = np.zeros(3, np.float32)
    load(fast).orc_screen_ray(C.byref(camera), float(px), float(py), _fptr(d))
    return d


def screen_points(camera, width, height, ystep=1):
    """Camera::screenToWorld of the pixel corners of every ystep-th row: [rows, width, 3]."""
    rows = (height + ystep - 1) // ystep
    out = np.zeros((rows, width, 3), np.float32)
    load().orc_screen_points(C.byref(camera), width, height, ystep, _fptr(out))
    return out


def write_ppm(path, rgb, fast=False):
    rgb = np.ascontiguousarray(rgb, np.float32)
    h, w, _ = rgb.shape
    return load(fast).orc_write_ppm(os.fsencode(path), _fptr(rgb), w, h)
