"""Timeline of ONE frame rendered alone (one stream, an idle GPU), from a rocprofv3 kernel trace.
  on the GPU box:  cd /tmp && rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 ROOT/tools/lone_trace.py run WORKLOAD [WORLD]
                   (WORLD > 1: rank 0's share of the frame under row tiles of 16 dealt to WORLD ranks)
  then:            python3 tools/lone_trace.py show OUT
prints every kernel of the last frame: start relative to the frame's first kernel, duration, and the gap since the kernel before."""
import csv
import glob
import os
import re
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)

if sys.argv[1] == "run":
    import torch

    from raytracer_amd import tiles, workloads

    w = workloads.get(sys.argv[2] if len(sys.argv) > 2 else "bunny_1080p")
    world = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    fly, params, cam = workloads.make_flyscene(w)
    p = tiles.part_params(fly.params_copy(row_tile=16), world, 0)
    buf = torch.zeros((max(fly.local_rows(p), 1), w["width"], 3), dtype=torch.float32, device="cuda")
    for _ in range(12):
        fly.render_device(buf, params=p, camera=cam)
        torch.cuda.synchronize()
else:
    rows = []
    for path in glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            m = re.search(r"(\w+_kernel(?:<[^>]*>)?)\(", r["Kernel_Name"])
            if m:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1)))
    rows.sort()
    # frames start with prep_kernel
    starts = [i for i, r in enumerate(rows) if r[2].startswith("prep_kernel")]
    last = rows[starts[-1]:]
    t0, prev_end = last[0][0], last[0][0]
    for s, e, name in last:
        print("%-34s start %7.1f us  lasts %6.1f us  gap %5.1f us" % (name, (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3))
        prev_end = e
    print("frame: %.1f us first start to last end" % ((last[-1][1] - t0) / 1e3))
