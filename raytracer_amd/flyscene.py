"""Python face of the GL-free Flyscene host (raytracer_amd/csrc/flyscene_host.hpp).

Mirrors the reference's entry points (src/flyscene.hpp:45-190): ``initialize(width, height)``,
``getCamera()``, ``addLight()``, ``raytraceScene()``.  All work happens in librtamd.so; this module
only moves pointers.  Device buffers for the asynchronous entry point are torch tensors.
"""
import ctypes as C
import os

import numpy as np

from . import _lib as L

DEFAULT_LIGHT = (-1.0, 1.0, 1.0)  # src/flyscene.cpp:9

_FLOAT_ARRAYS = {
    "raw_vertices": (-1, 3), "raw_normals": (-1, 3), "vertices": (-1, 3), "normals": (-1, 3), "face_normals": (-1, 3),
    "face_origin_distance": (-1,), "light_samples": (-1, 3), "materials": (-1, 10), "model_matrix": (3, 4),
}
_INT_ARRAYS = {"face_vertex_ids": (-1, 3), "face_material_ids": (-1,)}


def _fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _iptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


class Flyscene:
    def __init__(self, model=None, arrays=None, lights=None, hard_shadow=True, soft_points=12, soft_radius=0.3,
                 spheres=(), normalize_model=True, output="result.ppm", max_leaf_faces=0, sah_bins=0, textures=None,
                 shadow_rays=0, primary_rays=0, builder=0):
        """model: OBJ path (default: the reference's resources/models/bunny.obj, src/flyscene.cpp:39).
        arrays: (vertices[nv,3], normals[nv,3] or None, faces[nf,3], face_material[nf], materials) with
        materials a list of dicts ka/kd/ks/shininess.  lights: list of (x, y, z) or (x, y, z, r, g, b);
        None keeps the reference's single light at (-1, 1, 1).  textures (with arrays only; an OBJ brings its
        own through vt / map_kd): dict face_texcoords[nf,3,2], material_textures[nm] (-1 = none), images = list of
        [h,w,3] float arrays in the row order of the reference's texturedatas (file rows flipped)."""
        self.lib = L.lib()
        h = C.c_void_p()
        L.check(self.lib.rtfly_create(C.byref(h)), "rtfly_create")
        self.h = h
        if arrays is not None:
            v, n, f, fm, mats = arrays
            v = np.ascontiguousarray(v, np.float32)
            f = np.ascontiguousarray(f, np.int32)
            fm = np.ascontiguousarray(fm, np.int32)
            nptr = None
            if n is not None:
                n = np.ascontiguousarray(n, np.float32)
                nptr = _fptr(n)
            marr = (L.Material * max(1, len(mats)))()
            for i, m in enumerate(mats):
                marr[i].ka[:] = m["ka"]
                marr[i].kd[:] = m["kd"]
                marr[i].ks[:] = m["ks"]
                marr[i].shininess = m["shininess"]
            L.check(self.lib.rtfly_set_model_arrays(self.h, _fptr(v), nptr, len(v), _iptr(f), _iptr(fm), len(f), marr, len(mats)),
                    "rtfly_set_model_arrays")
            if textures is not None:
                tc = np.ascontiguousarray(textures["face_texcoords"], np.float32).reshape(len(f), 6)
                mt = np.ascontiguousarray(textures["material_textures"], np.int32).reshape(len(mats))
                imgs = [np.ascontiguousarray(im, np.float32) for im in textures["images"]]
                tarr = (L.Texture * max(1, len(imgs)))()
                for i, im in enumerate(imgs):
                    tarr[i].height, tarr[i].width = im.shape[0], im.shape[1]
                    tarr[i].rgb = _fptr(im)
                L.check(self.lib.rtfly_set_texture_arrays(self.h, _fptr(tc), _iptr(mt), tarr, len(imgs)), "rtfly_set_texture_arrays")
        elif model is not None:
            L.check(self.lib.rtfly_set_model(self.h, os.fsencode(model)), "rtfly_set_model")
        if lights is not None:
            L.check(self.lib.rtfly_clear_lights(self.h), "rtfly_clear_lights")
            for l in lights:
                pos = (C.c_float * 3)(*l[:3])
                col = (C.c_float * 3)(*(l[3:6] if len(l) >= 6 else (1.0, 1.0, 1.0)))
                L.check(self.lib.rtfly_add_light_at(self.h, pos, col), "rtfly_add_light_at")
        L.check(self.lib.rtfly_set_shadow_sampling(self.h, int(hard_shadow), int(soft_points), float(soft_radius)),
                "rtfly_set_shadow_sampling")
        for s in spheres:
            L.check(self.lib.rtfly_add_sphere(self.h, (C.c_float * 3)(*s[:3]), float(s[3])), "rtfly_add_sphere")
        L.check(self.lib.rtfly_set_normalize_model(self.h, int(normalize_model)), "rtfly_set_normalize_model")
        L.check(self.lib.rtfly_set_output(self.h, os.fsencode(output)), "rtfly_set_output")
        if max_leaf_faces or sah_bins or shadow_rays or primary_rays or builder:
            b = L.BuildParams()
            b.max_leaf_faces, b.sah_bins, b.shadow_rays = int(max_leaf_faces), int(sah_bins), int(shadow_rays)
            b.primary_rays, b.builder = int(primary_rays), int(builder)
            L.check(self.lib.rtfly_set_build_params(self.h, C.byref(b)), "rtfly_set_build_params")
        self.width = self.height = 0

    def close(self):
        if getattr(self, "h", None):
            self.lib.rtfly_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ reference entry points
    def initialize(self, width, height, upload=True):
        """Flyscene::initialize (src/flyscene.cpp:18-163).  upload=False keeps everything on the host
        (no GPU needed): only the flattened arrays are produced."""
        L.check(self.lib.rtfly_initialize(self.h, int(width), int(height), int(bool(upload))), "rtfly_initialize")
        self.width, self.height = int(width), int(height)
        return self

    def getCamera(self):
        cam = L.Camera()
        L.check(self.lib.rtfly_get_camera(self.h, C.byref(cam)), "rtfly_get_camera")
        return cam

    def addLight(self):
        L.check(self.lib.rtfly_add_light(self.h), "rtfly_add_light")

    def raytraceScene(self):
        """Flyscene::raytraceScene (src/flyscene.cpp:381-459): render and write the PPM."""
        L.check(self.lib.rtfly_raytrace_scene(self.h), "rtfly_raytrace_scene")

    # ------------------------------------------------------------------ camera
    def camera_reset(self):
        L.check(self.lib.rtfly_camera_reset(self.h), "rtfly_camera_reset")

    def camera_translate(self, dx, dy, dz):
        L.check(self.lib.rtfly_camera_translate(self.h, dx, dy, dz), "rtfly_camera_translate")
        L.check(self.lib.rtfly_camera_update_view(self.h), "rtfly_camera_update_view")

    def camera_rotate(self, angle_x, angle_y):
        L.check(self.lib.rtfly_camera_rotate(self.h, angle_x, angle_y), "rtfly_camera_rotate")
        L.check(self.lib.rtfly_camera_update_view(self.h), "rtfly_camera_update_view")

    # ------------------------------------------------------------------ parameters
    @property
    def params(self):
        """Live rtamd_render_params of the scene object (edit fields in place)."""
        return self.lib.rtfly_params(self.h).contents

    def set_params(self, **kw):
        p = self.params
        for k, v in kw.items():
            if isinstance(v, (tuple, list)):
                getattr(p, k)[:] = v
            else:
                setattr(p, k, v)
        return p

    def params_copy(self, **kw):
        p = L.RenderParams()
        C.memmove(C.byref(p), C.byref(self.params), C.sizeof(p))
        for k, v in kw.items():
            if isinstance(v, (tuple, list)):
                getattr(p, k)[:] = v
            else:
                setattr(p, k, v)
        return p

    # ------------------------------------------------------------------ rendering
    def render(self, face_ids=True):
        """Blocking render of the current view into numpy arrays (host-buffer entry point)."""
        p = self.params
        rows = self.lib.rtamd_local_rows(C.byref(p))
        rgb = np.zeros((rows, p.width, 3), np.float32)
        fid = np.full((rows, p.width), -3, np.int32) if face_ids else None
        st = L.Stats()
        L.check(self.lib.rtfly_raytrace_to(self.h, _fptr(rgb), _iptr(fid) if face_ids else None, C.byref(st)), "rtfly_raytrace_to")
        return {"rgb": rgb, "face_id": fid, "stats": st}

    @property
    def scene(self):
        s = self.lib.rtfly_scene(self.h)
        if not s:
            raise RuntimeError("scene not uploaded: call initialize(width, height) on a machine with a GPU")
        return C.c_void_p(s)

    def local_rows(self, params=None):
        p = params if params is not None else self.params
        return self.lib.rtamd_local_rows(C.byref(p))

    def render_device(self, rgb, face_id=None, t_hit=None, params=None, camera=None, stream=None, noalloc=False):
        """Enqueue one frame on `stream` (a torch.cuda.Stream, default: torch's current stream) writing
        into torch CUDA tensors.  No host synchronisation.  noalloc=True goes through
        rtamd_render_device_noalloc (work buffers in the library's static device array, nothing allocated)."""
        import torch

        p = params if params is not None else self.params
        cam = camera if camera is not None else self.getCamera()
        rows = self.lib.rtamd_local_rows(C.byref(p))
        need = rows * p.width
        if rgb.dtype != torch.float32 or not rgb.is_cuda or not rgb.is_contiguous() or rgb.numel() < need * 3:
            raise ValueError("rgb must be a contiguous float32 CUDA tensor with at least %d elements" % (need * 3))
        if face_id is not None and (face_id.dtype != torch.int32 or not face_id.is_cuda or not face_id.is_contiguous() or face_id.numel() < need):
            raise ValueError("face_id must be a contiguous int32 CUDA tensor with at least %d elements" % need)
        if t_hit is not None and (t_hit.dtype != torch.float32 or not t_hit.is_cuda or not t_hit.is_contiguous() or t_hit.numel() < need):
            raise ValueError("t_hit must be a contiguous float32 CUDA tensor with at least %d elements" % need)
        s = stream if stream is not None else torch.cuda.current_stream(rgb.device)
        entry = self.lib.rtamd_render_device_noalloc if noalloc else self.lib.rtamd_render_device
        L.check(entry(self.scene, C.byref(cam), C.byref(p), C.c_void_p(rgb.data_ptr()),
                      C.c_void_p(face_id.data_ptr()) if face_id is not None else None,
                      C.c_void_p(t_hit.data_ptr()) if t_hit is not None else None,
                      C.c_void_p(s.cuda_stream)), "rtamd_render_device_noalloc" if noalloc else "rtamd_render_device")

    def stats(self):
        st = L.Stats()
        L.check(self.lib.rtamd_render_stats(self.scene, C.byref(st)), "rtamd_render_stats")
        return st

    def workspace_bytes(self):
        n = C.c_uint64()
        L.check(self.lib.rtamd_scene_workspace_bytes(self.scene, C.byref(n)), "rtamd_scene_workspace_bytes")
        return n.value

    def set_workspace_limit(self, nbytes):
        """Frames whose work buffers would exceed `nbytes` are rendered in bands of rows (0: never)."""
        L.check(self.lib.rtamd_scene_set_workspace_limit(self.scene, C.c_uint64(int(nbytes))), "rtamd_scene_set_workspace_limit")

    def info(self):
        i = L.SceneInfo()
        L.check(self.lib.rtamd_scene_get_info(self.scene, C.byref(i)), "rtamd_scene_get_info")
        return i

    # ------------------------------------------------------------------ flattened scene
    def arrays(self):
        """struct PrecomputedData (src/defs.h:7-20) and friends as numpy copies."""
        out = {}
        for name, shape in list(_FLOAT_ARRAYS.items()) + list(_INT_ARRAYS.items()):
            ptr = C.c_void_p()
            n = C.c_int64()
            L.check(self.lib.rtfly_get_array(self.h, name.encode(), C.byref(ptr), C.byref(n)), "rtfly_get_array")
            dt = np.float32 if name in _FLOAT_ARRAYS else np.int32
            if n.value == 0 or not ptr.value:
                a = np.zeros(0, dt)
            else:
                ct = C.c_float if dt == np.float32 else C.c_int32
                a = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ct)), shape=(n.value,)).astype(dt, copy=True)
            out[name] = a.reshape(shape)
        return out


def _get_array(fly, name, dt):
    ptr = C.c_void_p()
    n = C.c_int64()
    L.check(fly.lib.rtfly_get_array(fly.h, name.encode(), C.byref(ptr), C.byref(n)), "rtfly_get_array")
    if n.value == 0 or not ptr.value:
        return np.zeros(0, dt)
    ct = C.c_float if dt == np.float32 else C.c_int32
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ct)), shape=(n.value,)).astype(dt, copy=True)


def texture_arrays(fly):
    """What the texture branch of shadeOffFace reads (src/flyscene.cpp:476-481,502,519-533), as numpy copies:
    face_texcoords[nf,3,2] (empty without `vt`), material_textures[nm], images (texturedatas)."""
    out = {"face_texcoords": _get_array(fly, "face_texcoords", np.float32).reshape(-1, 3, 2),
           "material_textures": _get_array(fly, "material_textures", np.int32), "images": []}
    k = 0
    while True:
        ptr = C.c_void_p()
        n = C.c_int64()
        if fly.lib.rtfly_get_array(fly.h, ("texture_size:%d" % k).encode(), C.byref(ptr), C.byref(n)) != L.OK:
            break
        w, h = _get_array(fly, "texture_size:%d" % k, np.int32)
        out["images"].append(_get_array(fly, "texture:%d" % k, np.float32).reshape(h, w, 3))
        k += 1
    return out


def write_ppm(path, rgb, binary=False):
    """ImageImporter::writePPMImage (utils/ppmIO.hpp:188-209); binary=True writes P6 with the same quantisation."""
    rgb = np.ascontiguousarray(rgb, np.float32)
    h, w, _ = rgb.shape
    if binary:
        L.check(L.lib().rtfly_write_ppm_binary(os.fsencode(path), _fptr(rgb), w, h), "rtfly_write_ppm_binary")
    else:
        L.check(L.lib().rtfly_write_ppm(os.fsencode(path), _fptr(rgb), w, h), "rtfly_write_ppm")


def default_camera(width, height):
    cam = L.Camera()
    L.lib().rtamd_camera_default(C.byref(cam), width, height)
    return cam


def default_params(width, height, **kw):
    p = L.RenderParams()
    L.lib().rtamd_render_params_default(C.byref(p), width, height)
    for k, v in kw.items():
        if isinstance(v, (tuple, list)):
            getattr(p, k)[:] = v
        else:
            setattr(p, k, v)
    return p
