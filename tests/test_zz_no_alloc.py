"""rtamd_render_device_noalloc (work buffers in the library's static device array, nothing allocated) against
rtamd_render_device on the same inputs, through the C ABI: every output plane bit for bit, on two calls in a row, the
second with a smaller image, so that whatever the first call left in the shared array must not leak into the second.
A frame too large for the array goes through it in bands of rows; one too large even so is refused.
Collected last on purpose (the array is shared by every scene on the GPU)."""
import numpy as np
import pytest
import torch

from conftest import scene_path

import raytracer_amd as R
from raytracer_amd import _lib as L

pytestmark = pytest.mark.gpu


def planes(fly, width, height, noalloc, **params):
    dev = torch.device("cuda", 0)
    p = fly.params_copy(width=width, height=height, **params)
    cam = R.flyscene.default_camera(width, height)
    rows = fly.local_rows(p)
    rgb = torch.full((rows, width, 3), -7.0, dtype=torch.float32, device=dev)
    fid = torch.full((rows, width), -9, dtype=torch.int32, device=dev)
    t = torch.full((rows, width), -7.0, dtype=torch.float32, device=dev)
    fly.render_device(rgb, fid, t, params=p, camera=cam, noalloc=noalloc)
    torch.cuda.synchronize(dev)
    st = fly.stats()
    return rgb.cpu().numpy(), fid.cpu().numpy(), t.cpu().numpy(), (st.rays_primary, st.rays_shadow, st.rays_bounce, st.hits)


def same(a, b):
    assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32))
    assert np.array_equal(a[1], b[1])
    assert np.array_equal(a[2].view(np.uint32), b[2].view(np.uint32))
    assert a[3] == b[3] and a[3][1] > 0


@pytest.mark.parametrize("name", ["toy", "reflect_test"])
def test_noalloc_entry_point_matches_render_device_twice_in_a_row(name):
    fly = R.Flyscene(model=scene_path(name), output="").initialize(320, 240)
    big = planes(fly, 320, 240, noalloc=True)
    small = planes(fly, 136, 100, noalloc=True)  # smaller, and not a multiple of the 8x8 tiles
    same(big, planes(fly, 320, 240, noalloc=False))
    same(small, planes(fly, 136, 100, noalloc=False))
    if name == "reflect_test":
        assert big[3][2] > 0  # bounce rays went through the static queues too
    # a row partition and SSAA through the same array
    part = dict(row_tile=8, num_parts=3, part=1, ssaa_level=2)
    same(planes(fly, 136, 100, noalloc=True, **part), planes(fly, 136, 100, noalloc=False, **part))


def test_noalloc_entry_point_renders_a_frame_larger_than_its_array_in_bands():
    """A 1080p frame needs more work buffers than the static array holds; the entry point cuts it into bands of rows that
    fit, one after the other through the array, and the frame is the same bit for bit."""
    fly = R.Flyscene(model=scene_path("reflect_test"), output="").initialize(64, 64)
    big = planes(fly, 1920, 1080, noalloc=True, max_reflections=1)
    same(big, planes(fly, 1920, 1080, noalloc=False, max_reflections=1))
    assert big[3][2] > 0
    same(planes(fly, 200, 120, noalloc=True), planes(fly, 200, 120, noalloc=False))  # and a small one right behind it


def test_noalloc_entry_point_refuses_frames_beyond_its_array():
    """Every band takes a frame's worth of stage counters, so a frame of many passes can be cut into few bands only: 24 passes
    (2 x 2 samples, five bounces) leave two bands, and half a 1080p frame does not fit the array."""
    fly = R.Flyscene(model=scene_path("reflect_test"), output="").initialize(64, 64)
    with pytest.raises(L.RtamdError) as e:
        planes(fly, 1920, 1080, noalloc=True, ssaa_level=2, max_reflections=5)
    assert e.value.code == L.ERR_LIMIT and "rtamd_render_device_noalloc holds" in str(e.value)
    # the scene is still usable afterwards
    same(planes(fly, 64, 64, noalloc=True), planes(fly, 64, 64, noalloc=False))
