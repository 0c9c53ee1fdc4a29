"""Scene-build timing: python tools/time_build.py WORKLOAD [BUILDER]; set RTAMD_VERBOSE_BUILD=1 for the phases."""
import sys
import time

sys.path.insert(0, ".")
from raytracer_amd import workloads  # noqa: E402

w = workloads.get(sys.argv[1])
builder = int(sys.argv[2]) if len(sys.argv) > 2 else 0
for attempt in range(2):  # the second build shows the cost without first-use overheads of the process
    t = time.time()
    fly, params, cam = workloads.make_flyscene(w, builder=builder)
    i = fly.info()
    print("%s builder=%d attempt %d: total %.0f ms (mesh generation included), build_ms %.1f hierarchy_ms %.1f upload_ms %.1f nodes %d" %
          (sys.argv[1], i.builder, attempt, 1e3 * (time.time() - t), i.build_ms, i.hierarchy_ms, i.upload_ms, i.num_nodes), flush=True)
    fly.close()
