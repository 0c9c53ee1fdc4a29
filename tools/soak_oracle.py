"""Soak against the oracle: random soups with corners snapped onto the rays of pixel samples, onto the coordinate planes and
onto each other (shared corners and edges, coincident faces), small images, every GPU frame against the oracle run that accepts
faces by the face test alone (all_faces, NaN decisions rejected: the product's definition of a frame) on every pixel, and
against the faithful run wherever the two oracle runs agree:  python tools/soak_oracle.py [FIRST_SEED [COUNT]]"""
import sys

import numpy as np

_ROOT = __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))
sys.path.insert(0, _ROOT)
sys.path.insert(0, _ROOT + "/tests")
from parity_util import PIXEL_TOL, compare, make_pair  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = box_pixels = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(10_000 + seed)
    W, H = [(64, 48), (81, 61), (47, 95), (120, 33)][seed % 4]
    ssaa = 1 + (seed // 4) % 2
    aspect, tan_half = W / H, np.tan(np.radians(30.0))
    n = int(rng.choice([20, 120, 600]))
    # corners: on rays of pixel samples (a few lattice points, so that faces share corners and edges), at random depths
    pts = int(rng.integers(6, 40))
    sub = rng.integers(0, ssaa, pts) / ssaa
    px, py = rng.integers(0, W, pts) + sub, rng.integers(0, H, pts) + sub
    depth = rng.uniform(1.0, 3.0, pts)
    nx, ny = (2.0 * px / W - 1.0) * aspect * tan_half, (1.0 - 2.0 * py / H) * tan_half
    corners = np.stack([nx * depth, ny * depth, 2.0 - depth], 1)
    tri = corners[rng.integers(0, pts, (n, 3))] + (rng.random((n, 3, 1)) < 0.3) * rng.uniform(-0.3, 0.3, (n, 3, 3))
    snap = rng.random((n, 3, 3)) < 0.05
    tri = np.where(snap, 0.0, tri).astype(np.float32)  # some coordinates exactly on the coordinate planes
    verts = tri.reshape(-1, 3)
    faces = np.arange(3 * n, dtype=np.int32).reshape(n, 3)
    mats = [dict(ka=rng.uniform(0, 0.2, 3).tolist(), kd=rng.uniform(0.1, 1.0, 3).tolist(), ks=[float(rng.choice([0.0, 0.3, 0.9]))] * 3,
                 shininess=float(rng.choice([1.0, 8.0, 64.0]))) for _ in range(3)]
    arrays = (verts, None, faces, rng.integers(0, 3, n).astype(np.int32), mats)
    lights = [tuple(rng.uniform(-2.0, 2.0, 3).tolist()) for _ in range(1 + seed % 2)]
    if seed % 5 == 0:
        lights.append((0.0, 1.5, 0.0))  # a light on an axis
    spheres = [tuple(rng.uniform(-0.6, 0.6, 3).tolist()) + (float(rng.uniform(0.05, 0.3)),) for _ in range(int(rng.integers(1, 4)))] if seed % 7 == 3 else ()
    fly, orc = make_pair(arrays=arrays, lights=lights, soft_points=seed % 3, normalize_model=False, width=W, height=H, spheres=spheres)
    r = compare(fly, orc, W, H, all_faces=True, max_reflections=2, ssaa_level=ssaa)
    ok = (r["face_mismatch_pixels_all_faces"] == 0 and r["bad_pixels_all_faces"] == 0 and r["rays_gpu"] == r["rays_oracle_all_faces"]
          and r["face_mismatch_pixels_where_boxes_agree"] == 0 and r["bad_pixels_where_boxes_agree"] == 0)
    box_pixels += r["box_dependent_pixels"]
    if not ok:
        bad += 1
        print("MISMATCH seed %d %dx%d ssaa %d, %d faces: %d face ids, %d colours against all_faces; rays %s vs %s; where the boxes agree: %d, %d" % (
            seed, W, H, ssaa, n, r["face_mismatch_pixels_all_faces"], r["bad_pixels_all_faces"], r["rays_gpu"], r["rays_oracle_all_faces"],
            r["face_mismatch_pixels_where_boxes_agree"], r["bad_pixels_where_boxes_agree"]), flush=True)
    fly.close()
    if (seed - first) % 10 == 9:
        print("seeds %d..%d done, %d mismatches, %d pixels decided by the reference's boxes so far" % (first, seed, bad, box_pixels), flush=True)
print("soak against the oracle: %d seeds, %d mismatches, %d pixels decided by the reference's boxes (tolerance %g)" % (count, bad, box_pixels, PIXEL_TOL))
sys.exit(1 if bad else 0)
