"""Shared helpers of the parity tests: run the same scene through the HIP path and the CPU oracle and
compare the planes.  Only tests (and the smoke test / cpu_baseline leg) import the oracle."""
import os

import numpy as np

from oracle.oracle_py import OracleScene

import raytracer_amd as R

SCENES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "scenes")

# Tolerances of the parity gate.
FACE_ID_MIN_MATCH = 0.999   # share of primary rays with the same face id (acceptance bar)
PIXEL_TOL = 1e-3            # per-channel colour error on pixels whose face ids agree (acceptance bar)


def scene_path(name):
    return os.path.join(SCENES, name + ".obj")


def make_pair(name=None, arrays=None, lights=None, hard_shadow=True, soft_points=12, soft_radius=0.3, spheres=(),
              width=64, height=64, **build):
    """(Flyscene on the GPU, OracleScene on the CPU) for the same inputs."""
    kw = dict(lights=lights, hard_shadow=hard_shadow, soft_points=soft_points, soft_radius=soft_radius, spheres=spheres)
    if "normalize_model" in build:  # (an argument of both)
        kw["normalize_model"] = build.pop("normalize_model")
    if arrays is not None:
        fly = R.Flyscene(arrays=arrays, output="", **kw, **build)
        okw = dict(kw)
        okw["lights"] = [tuple(l) + (1.0, 1.0, 1.0) if len(l) == 3 else tuple(l) for l in (lights or [(-1.0, 1.0, 1.0)])]
        orc = OracleScene(arrays=arrays, **okw)
    else:
        fly = R.Flyscene(model=scene_path(name), output="", **kw, **build)
        okw = dict(kw)
        okw["lights"] = [tuple(l) + (1.0, 1.0, 1.0) if len(l) == 3 else tuple(l) for l in (lights or [(-1.0, 1.0, 1.0)])]
        orc = OracleScene(scene_path(name), **okw)
    fly.initialize(width, height)
    return fly, orc


def oracle_camera(orc, cam):
    """Copy an rtamd camera into the oracle's camera struct (same fields)."""
    from oracle.oracle_py import Camera

    c = Camera()
    c.origin[:] = list(cam.origin)
    c.inv_view[:] = list(cam.inv_view)
    c.viewport[:] = list(cam.viewport)
    c.aspect_ratio = cam.aspect_ratio
    c.fovy = cam.fovy
    return c


def compare(fly, orc, width, height, all_faces=False, **params):
    """Render with both and return a dict of agreement figures.

    all_faces: a third oracle run in which no box turns a ray away (every face is tried, in the reference's order; NaN
    decisions rejected): the HIP path's definition of a frame, to be matched on EVERY pixel.  The reference's own box test
    rounds and may turn a ray away from a face its face test accepts (a ray through a corner of a leaf's box); scenes built
    to put rays there are compared with this run, and with the faithful run wherever the two oracle runs agree.

    The oracle is run twice: as the reference behaves (NaN plane distances accepted) and with NaN
    distances rejected, which is what the HIP path defines.  Agreement is demanded against the
    faithful run on every pixel the oracle did not flag as NaN-dependent, and against the NaN-rejecting
    run on every pixel."""
    fly.set_params(**params)
    got = fly.render()
    cam = oracle_camera(orc, fly.getCamera())
    okw = {k: v for k, v in params.items() if k in ("ssaa_level", "max_reflections", "reflect_threshold", "bounce_offset",
                                                   "t_epsilon", "background", "background_deep")}
    ref = orc.render(width, height, camera=cam, **okw)
    ref2 = orc.render(width, height, camera=cam, reject_nan=1, **okw)
    clean = ref["nan_mask"] == 0
    same = got["face_id"] == ref["face_id"]
    same2 = got["face_id"] == ref2["face_id"]
    err = np.abs(got["rgb"] - ref["rgb"]).max(axis=-1)
    err2 = np.abs(got["rgb"] - ref2["rgb"]).max(axis=-1)
    st, cnt = got["stats"], ref2["counters"]
    extra = {}
    if all_faces:
        ref3 = orc.render(width, height, camera=cam, reject_nan=1, all_faces=1, **okw)
        same3 = got["face_id"] == ref3["face_id"]
        err3 = np.abs(got["rgb"] - ref3["rgb"]).max(axis=-1)
        boxes_agree = (ref3["face_id"] == ref2["face_id"]) & (np.abs(ref3["rgb"] - ref2["rgb"]).max(axis=-1) == 0)
        extra = {
            "face_mismatch_pixels_all_faces": int((~same3).sum()),
            "bad_pixels_all_faces": int((err3 > PIXEL_TOL).sum()),
            "rays_oracle_all_faces": (ref3["counters"].rays_primary, ref3["counters"].rays_shadow, ref3["counters"].rays_bounce),
            "box_dependent_pixels": int((~boxes_agree).sum()),  # pixels the reference's boxes decide
            "face_mismatch_pixels_where_boxes_agree": int((~same2 & boxes_agree).sum()),
            "bad_pixels_where_boxes_agree": int(((err2 > PIXEL_TOL) & boxes_agree).sum()),
            "ref_all_faces": ref3,
        }
    return {
        **extra,
        "pixels": width * height,
        "nan_pixels": int((~clean).sum()),
        "face_match": float(same[clean].mean()) if clean.any() else 1.0,
        "face_mismatch_pixels": int((~same & clean).sum()),
        "face_match_nanfree": float(same2.mean()),
        "face_mismatch_pixels_nanfree": int((~same2).sum()),
        "max_err_matching": float(err[same & clean].max()) if (same & clean).any() else 0.0,
        "max_err_nanfree_matching": float(err2[same2].max()) if same2.any() else 0.0,
        "bad_pixels_nanfree": int((err2 > PIXEL_TOL).sum()),
        "rays_gpu": (st.rays_primary, st.rays_shadow, st.rays_bounce),
        "rays_oracle": (cnt.rays_primary, cnt.rays_shadow, cnt.rays_bounce),
        "got": got, "ref": ref, "ref_nanfree": ref2,
    }
