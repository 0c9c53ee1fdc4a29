"""MI355X (gfx950) render pass for the Flyscene ray tracer.

Layout: csrc/ holds the HIP kernels, the C ABI (include/rtamd.h) and the GL-free C++ Flyscene host;
flyscene.py is the ctypes face of that host; tiles.py splits a frame into row tiles across GPUs and
gathers them with torch.distributed (RCCL).  The shared library is required: nothing here computes on
the CPU."""
from . import _lib
from .flyscene import Flyscene, default_camera, default_params, write_ppm

__all__ = ["Flyscene", "default_camera", "default_params", "write_ppm", "_lib"]
