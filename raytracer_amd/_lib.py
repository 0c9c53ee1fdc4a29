"""ctypes view of raytracer_amd/lib/librtamd.so (the C ABI declared in include/rtamd.h and
include/rtamd_flyscene.h).  There is no fallback: if the library is missing or does not export a
declared symbol, importing this module raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RTAMD_LIB") or os.path.join(_HERE, "lib", "librtamd.so")  # RTAMD_LIB: experiment builds

OK, ERR_INVALID, ERR_HIP, ERR_LIMIT, ERR_NO_DEVICE = 0, 1, 2, 3, 4
FACE_SPHERE, FACE_MISS = -1, -2
FLAG_COUNT_WORK, FLAG_TIME_KERNELS, FLAG_SINGLE_STREAM = 1, 2, 4
NOALLOC_WORKSPACE_BYTES = 64 << 20
SHADOW_RAYS_AUTO, SHADOW_RAYS_TREE = 0, 1
PRIMARY_RAYS_AUTO, PRIMARY_RAYS_TREE, PRIMARY_RAYS_BINS = 0, 1, 2
BUILDER_HOST_SAH, BUILDER_DEVICE_LBVH = 0, 1


class Material(C.Structure):
    _fields_ = [("ka", C.c_float * 3), ("kd", C.c_float * 3), ("ks", C.c_float * 3), ("shininess", C.c_float)]


class Sphere(C.Structure):
    _fields_ = [("center", C.c_float * 3), ("radius", C.c_float)]


_fp, _ip = C.POINTER(C.c_float), C.POINTER(C.c_int32)


class Texture(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("rgb", _fp)]


class SceneDesc(C.Structure):
    _fields_ = [
        ("vertices", _fp), ("normals", _fp), ("num_vertices", C.c_int32),
        ("face_vertex_ids", _ip), ("face_normals", _fp), ("face_origin_distance", _fp), ("face_material_ids", _ip),
        ("num_faces", C.c_int32),
        ("materials", C.POINTER(Material)), ("num_materials", C.c_int32),
        ("light_positions", _fp), ("light_colors", _fp), ("num_lights", C.c_int32),
        ("light_samples", _fp), ("samples_per_light", C.c_int32),
        ("spheres", C.POINTER(Sphere)), ("num_spheres", C.c_int32),
        ("face_texcoords", _fp), ("material_textures", _ip), ("textures", C.POINTER(Texture)), ("num_textures", C.c_int32),
        ("num_unsampled_lights", C.c_int32),
    ]


class BuildParams(C.Structure):
    _fields_ = [("max_leaf_faces", C.c_int32), ("sah_bins", C.c_int32), ("shadow_rays", C.c_int32), ("primary_rays", C.c_int32),
                ("builder", C.c_int32), ("reserved", C.c_int32 * 3)]


class Camera(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("inv_view", C.c_float * 12), ("viewport", C.c_float * 4),
                ("aspect_ratio", C.c_float), ("fovy", C.c_float)]


class RenderParams(C.Structure):
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32), ("ssaa_level", C.c_int32), ("max_reflections", C.c_int32),
        ("reflect_threshold", C.c_float), ("bounce_offset", C.c_float), ("t_epsilon", C.c_float),
        ("background", C.c_float * 3), ("background_deep", C.c_float * 3),
        ("row_tile", C.c_int32), ("num_parts", C.c_int32), ("part", C.c_int32),
        ("flags", C.c_uint32), ("reserved", C.c_int32 * 4),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("rays_primary", C.c_uint64), ("rays_shadow", C.c_uint64), ("rays_bounce", C.c_uint64), ("hits", C.c_uint64),
        ("box_tests", C.c_uint64 * 3), ("tri_plane", C.c_uint64 * 3), ("tri_edge", C.c_uint64 * 3),
        ("ms_trace", C.c_float), ("ms_shadow", C.c_float), ("ms_compact", C.c_float), ("ms_shade", C.c_float),
        ("ms_frames", C.c_float), ("launches_trace", C.c_int32), ("launches_shadow", C.c_int32), ("frames", C.c_int32),
        ("ms_raster", C.c_float), ("launches_raster", C.c_int32),
        ("bin_entries", C.c_uint32), ("bins_nonempty", C.c_uint32), ("primary_path", C.c_int32), ("bin_overflow", C.c_int32),
    ]

    @property
    def rays(self):
        return self.rays_primary + self.rays_shadow + self.rays_bounce

    def as_dict(self):
        out = {}
        for k, _ in self._fields_:
            if k == "reserved":
                continue
            v = getattr(self, k)
            out[k] = list(v) if hasattr(v, "__len__") else v
        return out


class SceneInfo(C.Structure):
    _fields_ = [
        ("num_faces", C.c_int32), ("num_device_faces", C.c_int32), ("num_nodes", C.c_int32), ("num_leaves", C.c_int32),
        ("depth", C.c_int32), ("device", C.c_int32), ("device_bytes", C.c_uint64), ("build_ms", C.c_float),
        ("upload_ms", C.c_float), ("shadow_grid_entries", C.c_uint64), ("builder", C.c_int32), ("hierarchy_ms", C.c_float),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


_P = C.c_void_p
# name -> (restype, argtypes); every symbol declared in include/*.h is listed here.
SYMBOLS = {
    "rtamd_last_error": (C.c_char_p, []),
    "rtamd_camera_default": (None, [C.POINTER(Camera), C.c_int32, C.c_int32]),
    "rtamd_render_params_default": (None, [C.POINTER(RenderParams), C.c_int32, C.c_int32]),
    "rtamd_scene_create": (C.c_int, [C.POINTER(SceneDesc), C.POINTER(BuildParams), C.POINTER(_P)]),
    "rtamd_scene_destroy": (C.c_int, [_P]),
    "rtamd_scene_set_lights": (C.c_int, [_P, _fp, _fp, C.c_int32]),
    "rtamd_scene_get_info": (C.c_int, [_P, C.POINTER(SceneInfo)]),
    "rtamd_scene_workspace_bytes": (C.c_int, [_P, C.POINTER(C.c_uint64)]),
    "rtamd_scene_set_workspace_limit": (C.c_int, [_P, C.c_uint64]),
    "rtamd_reference_face_rank": (C.c_int, [C.POINTER(SceneDesc), _ip]),
    "rtamd_local_rows": (C.c_int32, [C.POINTER(RenderParams)]),
    "rtamd_global_row": (C.c_int32, [C.POINTER(RenderParams), C.c_int32]),
    "rtamd_render": (C.c_int, [_P, C.POINTER(Camera), C.POINTER(RenderParams), _fp, _ip, C.POINTER(Stats)]),
    "rtamd_render_ppm": (C.c_int, [_P, C.POINTER(Camera), C.POINTER(RenderParams), _P, C.POINTER(C.c_int32), C.POINTER(Stats)]),
    "rtamd_host_alloc": (C.c_int, [C.POINTER(_P), C.c_uint64]),
    "rtamd_host_free": (C.c_int, [_P]),
    "rtamd_render_device": (C.c_int, [_P, C.POINTER(Camera), C.POINTER(RenderParams), _P, _P, _P, _P]),
    "rtamd_render_stats": (C.c_int, [_P, C.POINTER(Stats)]),
    "rtamd_render_device_noalloc": (C.c_int, [_P, C.POINTER(Camera), C.POINTER(RenderParams), _P, _P, _P, _P]),
    "rtamd_quantize_ppm": (C.c_int, [_P, _P, C.c_uint64, _P]),
    "rtamd_assemble_rows": (C.c_int, [C.POINTER(RenderParams), _P, _P, C.c_int32, _P]),
    "rtfly_create": (C.c_int, [C.POINTER(_P)]),
    "rtfly_destroy": (C.c_int, [_P]),
    "rtfly_set_model": (C.c_int, [_P, C.c_char_p]),
    "rtfly_set_model_arrays": (C.c_int, [_P, _fp, _fp, C.c_int32, _ip, _ip, C.c_int32, C.POINTER(Material), C.c_int32]),
    "rtfly_set_texture_arrays": (C.c_int, [_P, _fp, _ip, C.POINTER(Texture), C.c_int32]),
    "rtfly_clear_lights": (C.c_int, [_P]),
    "rtfly_add_light_at": (C.c_int, [_P, _fp, _fp]),
    "rtfly_set_shadow_sampling": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_float]),
    "rtfly_add_sphere": (C.c_int, [_P, _fp, C.c_float]),
    "rtfly_set_normalize_model": (C.c_int, [_P, C.c_int32]),
    "rtfly_set_output": (C.c_int, [_P, C.c_char_p]),
    "rtfly_set_build_params": (C.c_int, [_P, C.POINTER(BuildParams)]),
    "rtfly_initialize": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32]),
    "rtfly_add_light": (C.c_int, [_P]),
    "rtfly_camera_reset": (C.c_int, [_P]),
    "rtfly_camera_translate": (C.c_int, [_P, C.c_float, C.c_float, C.c_float]),
    "rtfly_camera_rotate": (C.c_int, [_P, C.c_float, C.c_float]),
    "rtfly_camera_update_view": (C.c_int, [_P]),
    "rtfly_get_camera": (C.c_int, [_P, C.POINTER(Camera)]),
    "rtfly_params": (C.POINTER(RenderParams), [_P]),
    "rtfly_raytrace_scene": (C.c_int, [_P]),
    "rtfly_raytrace_to": (C.c_int, [_P, _fp, _ip, C.POINTER(Stats)]),
    "rtfly_scene": (_P, [_P]),
    "rtfly_get_array": (C.c_int, [_P, C.c_char_p, C.POINTER(_P), C.POINTER(C.c_int64)]),
    "rtfly_write_ppm": (C.c_int, [C.c_char_p, _fp, C.c_int32, C.c_int32]),
    "rtfly_write_ppm_binary": (C.c_int, [C.c_char_p, _fp, C.c_int32, C.c_int32]),
}

_lib = None
HIP_RUNTIME = None


def _load_hip_runtime():
    """librtamd.so carries no DT_NEEDED entry for the HIP runtime: the process must hold exactly one copy of
    libamdhip64 and it has to be the one PyTorch uses, because device buffers, streams and RCCL
    communicators come from torch.  PyTorch's ROCm wheels bundle their own copy (loaded RTLD_LOCAL), so it
    is re-opened here with RTLD_GLOBAL, which makes its hip* symbols resolve for the library loaded next.
    Without torch the system runtime under /opt/rocm is used."""
    global HIP_RUNTIME
    candidates = []
    try:
        import torch

        candidates.append(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    except ImportError:
        pass
    candidates += [os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", "libamdhip64.so"), "libamdhip64.so"]
    for path in candidates:
        if os.path.isabs(path) and not os.path.exists(path):
            continue
        try:
            C.CDLL(path, mode=C.RTLD_GLOBAL)
            HIP_RUNTIME = path
            return
        except OSError:
            continue
    raise ImportError("no HIP runtime (libamdhip64.so) could be loaded; tried %s" % ", ".join(candidates))


def lib():
    """Load the shared library once and attach prototypes.  Raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s not found: build it with `make -C raytracer_amd/csrc` (there is no CPU fallback)" % LIB_PATH)
    _load_hip_runtime()
    handle = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(handle, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = handle
    return _lib


class RtamdError(RuntimeError):
    def __init__(self, code, what):
        msg = lib().rtamd_last_error()
        super().__init__("%s failed with code %d: %s" % (what, code, msg.decode() if msg else ""))
        self.code = code


def check(code, what):
    if code != OK:
        raise RtamdError(code, what)
