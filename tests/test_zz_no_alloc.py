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


def test_noalloc_entry_point_allocates_nothing_and_can_be_captured_in_a_graph():
    """The property the entry point exists for.  On a warmed scene the free device memory is the same before and after a
    run of calls, a scene rendered through this entry point alone owns no work buffers, and a call can be enqueued under
    stream capture -- where an allocation, an event or a synchronisation would abort the capture -- and replayed to the
    same bits."""
    dev = torch.device("cuda", 0)
    W, H = 200, 120
    fly = R.Flyscene(model=scene_path("reflect_test"), output="").initialize(W, H)
    ref = R.Flyscene(model=scene_path("reflect_test"), output="").initialize(W, H)
    want = planes(ref, W, H, noalloc=False)
    p = fly.params_copy(width=W, height=H)
    cam = R.flyscene.default_camera(W, H)
    rgb = torch.full((H, W, 3), -7.0, dtype=torch.float32, device=dev)
    fid = torch.full((H, W), -9, dtype=torch.int32, device=dev)
    t = torch.full((H, W), -7.0, dtype=torch.float32, device=dev)
    fly.render_device(rgb, fid, t, params=p, camera=cam, noalloc=True)  # warm-up: code objects, the per-device array
    torch.cuda.synchronize(dev)
    free_before = torch.cuda.mem_get_info(dev)[0]
    for _ in range(5):
        fly.render_device(rgb, fid, t, params=p, camera=cam, noalloc=True)
    torch.cuda.synchronize(dev)
    assert torch.cuda.mem_get_info(dev)[0] == free_before
    assert fly.workspace_bytes() == 0
    assert np.array_equal(rgb.cpu().numpy().view(np.uint32), want[0].view(np.uint32))

    side = torch.cuda.Stream(dev)
    graph = torch.cuda.CUDAGraph()
    rgb.fill_(-7.0), fid.fill_(-9), t.fill_(-7.0)
    torch.cuda.synchronize(dev)
    with torch.cuda.graph(graph, stream=side):
        fly.render_device(rgb, fid, t, params=p, camera=cam, stream=torch.cuda.current_stream(dev), noalloc=True)
    torch.cuda.synchronize(dev)
    assert float(rgb.max()) == -7.0  # captured, not run
    for _ in range(2):
        rgb.fill_(-7.0), fid.fill_(-9), t.fill_(-7.0)
        graph.replay()
        torch.cuda.synchronize(dev)
        assert np.array_equal(rgb.cpu().numpy().view(np.uint32), want[0].view(np.uint32))
        assert np.array_equal(fid.cpu().numpy(), want[1])
        assert np.array_equal(t.cpu().numpy().view(np.uint32), want[2].view(np.uint32))


def test_two_scenes_take_turns_in_the_shared_array():
    """Every scene on a GPU lays its work buffers out in the same static array: frames of different scenes, one after the
    other, must not see each other's leftovers (keys of the screen-space pass, counters, queues)."""
    a = R.Flyscene(model=scene_path("toy"), output="").initialize(160, 120)
    b = R.Flyscene(model=scene_path("reflect_test"), output="").initialize(160, 120)
    want_a, want_b = planes(a, 160, 120, noalloc=False), planes(b, 144, 81, noalloc=False)
    for _ in range(2):
        same(planes(a, 160, 120, noalloc=True), want_a)
        same(planes(b, 144, 81, noalloc=True), want_b)
