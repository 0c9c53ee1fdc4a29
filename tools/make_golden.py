#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the CPU oracle (oracle/liboracle.so, canonical arithmetic).

The reference ships no golden images and cannot be built or run here, so these vectors pin the
ORACLE, not the reference: they guard the restatement against accidental change and give the GPU tests
inputs/outputs that travel to the GPU box.  Run from the repository root:

    make -C oracle && python tools/make_golden.py
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tempfile  # noqa: E402

from oracle.oracle_py import OracleScene, write_ppm  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
SMALL = ["reflect_test", "reflect_test_2", "shadowmodel", "NthTest", "glass"]
TEXTURED = ["textured_quad"]  # synthetic scene of this repository: vt + map_kd + ASCII PPM textures
LARGE = ["toy", "toy_on_plane", "dodgeColorTest", "bunny"]


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def one(name, size, full):
    s = OracleScene(os.path.join(ROOT, "tests", "scenes", name + ".obj"))
    r = s.render(size, size)
    r2 = s.render(size, size, reject_nan=1)
    a = s.arrays()
    c = r["counters"]
    out = {
        "size": np.int32(size),
        "face_id": r["face_id"], "rgb": r["rgb"], "nan_mask": r["nan_mask"].astype(np.uint8),
        "face_id_nanfree": r2["face_id"], "rgb_nanfree": r2["rgb"],
        "rays": np.array([c.rays_primary, c.rays_shadow, c.rays_bounce], np.int64),
        "rays_nanfree": np.array([r2["counters"].rays_primary, r2["counters"].rays_shadow, r2["counters"].rays_bounce], np.int64),
        "bvh_stats": np.array([s.bvh_stats()[k] for k in ("nodes", "leaves", "depth")], np.int32),
        "face_order_sha256": np.array(digest(s.bvh_face_order())),
    }
    for k in ("vertices", "normals", "face_normals", "face_origin_distance", "face_vertex_ids", "face_material_ids", "light_samples",
              "materials"):
        out["sha256_" + k] = np.array(digest(a[k]))
    t = s.texture_arrays()
    out["sha256_face_texcoords"] = np.array(digest(t["face_texcoords"]))
    out["sha256_textures"] = np.array(digest(np.concatenate([im.reshape(-1) for im in t["images"]]) if t["images"] else np.zeros(0, np.float32)))
    with tempfile.TemporaryDirectory() as tmp:  # the bytes of result.ppm (writePPMImage, utils/ppmIO.hpp:188-209)
        write_ppm(os.path.join(tmp, "result.ppm"), r["rgb"])
        out["ppm_sha256"] = np.array(hashlib.sha256(open(os.path.join(tmp, "result.ppm"), "rb").read()).hexdigest())
    if full:
        boxes, info = s.bvh_dump()
        out.update({"t": r["t"], "light": r["light"], "bvh_boxes": boxes, "bvh_info": info, "face_order": s.bvh_face_order(),
                    "vertices": a["vertices"], "face_origin_distance": a["face_origin_distance"], "light_samples": a["light_samples"]})
    np.savez_compressed(os.path.join(OUT, "%s_%d.npz" % (name, size)), **out)
    print(name, size, "rays", out["rays"].tolist(), "nan pixels", int(r["nan_mask"].sum()))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    for n in SMALL:
        one(n, 64, True)
        one(n, 256, False)
    for n in LARGE:
        one(n, 256, False)
    for n in TEXTURED:
        one(n, 64, True)
        one(n, 256, False)
