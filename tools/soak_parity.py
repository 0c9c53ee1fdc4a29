"""Soak: random triangle soups (the generator of tests/test_gpu_parity.py) and dust clouds, the default path (primary rays
face by face in screen space, shadow grids), the screen bins and the device builders against the tree walk for every ray,
bit for bit, whole and in as many bands of rows as the library will cut (each band a partitioned frame), at sizes and
seeds the test suite does not run:  python tools/soak_parity.py [FIRST_SEED [COUNT]]"""
import sys

import numpy as np

_ROOT = __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))
sys.path.insert(0, _ROOT)
sys.path.insert(0, _ROOT + "/tests")
import raytracer_amd as R  # noqa: E402
from raytracer_amd import _lib as L  # noqa: E402
from test_gpu_parity import _random_soup  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 16
count = int(sys.argv[2]) if len(sys.argv) > 2 else 48
bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    arrays, lights = _random_soup(seed)
    if seed % 3 == 0:  # shrink the soup: faces of a few pixels and less
        verts = arrays[0] * np.float32(1.0)
        n = len(arrays[2])
        tri = verts.reshape(n, 3, 3)
        tri = (tri.mean(axis=1, keepdims=True) + (tri - tri.mean(axis=1, keepdims=True)) * rng.choice([0.01, 0.05, 0.2])).astype(np.float32)
        arrays = (tri.reshape(-1, 3),) + tuple(arrays[1:])
    W, H = [(320, 240), (511, 257), (1280, 720), (97, 1031)][seed % 4]
    kw = dict(arrays=arrays, lights=lights, soft_points=seed % 3, output="")
    ref = R.Flyscene(primary_rays=L.PRIMARY_RAYS_TREE, shadow_rays=L.SHADOW_RAYS_TREE, **kw).initialize(W, H)
    ref.set_params(max_reflections=2)
    want = ref.render()
    for build in (dict(), dict(builder=L.BUILDER_DEVICE_LBVH), dict(primary_rays=L.PRIMARY_RAYS_BINS),
                  dict(primary_rays=L.PRIMARY_RAYS_BINS, builder=L.BUILDER_DEVICE_LBVH)):
        for banded in (False, True):
            f = R.Flyscene(**kw, **build).initialize(W, H)
            f.set_params(max_reflections=2)
            if banded:
                f.set_workspace_limit(1)  # as many bands as the stage counters allow
            got = f.render()
            same = np.array_equal(got["face_id"], want["face_id"]) and np.array_equal(got["rgb"].view(np.uint32), want["rgb"].view(np.uint32))
            same = same and got["stats"].rays_shadow == want["stats"].rays_shadow and got["stats"].rays_bounce == want["stats"].rays_bounce
            if not same:
                bad += 1
                print("MISMATCH seed %d %dx%d %s banded=%s: %d face ids, %d colour words differ" % (
                    seed, W, H, build, banded, int((got["face_id"] != want["face_id"]).sum()),
                    int((got["rgb"].view(np.uint32) != want["rgb"].view(np.uint32)).sum())), flush=True)
            f.close()
    if (seed - first) % 8 == 7:
        print("seeds %d..%d done, %d mismatches so far" % (first, seed, bad), flush=True)
print("soak: %d seeds, %d mismatches" % (count, bad))
sys.exit(1 if bad else 0)
