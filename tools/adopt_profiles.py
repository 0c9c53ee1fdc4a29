"""Copy what tools/collect_profiles.sh left under gpurun_out/profiles_TAG/ into profiles/ (tracked) and print the figures the
documents quote:  python tools/adopt_profiles.py [TAG]"""
import csv
import glob
import json
import os
import re
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r3"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "profiles_" + tag)
dst = os.path.join(root, "profiles")
for f in glob.glob(os.path.join(src, "bench_*.json")):
    shutil.copy(f, os.path.join(dst, "%s_%s" % (tag, os.path.basename(f))))
for f in glob.glob(os.path.join(src, "pmc_*.txt")) + glob.glob(os.path.join(src, "kernel_stats_*.csv")):
    shutil.copy(f, os.path.join(dst, "%s_%s" % (tag, os.path.basename(f))))


def line(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


for f in sorted(glob.glob(os.path.join(dst, tag + "_bench_*.json"))):
    if "under" in f:
        continue
    d = line(f)
    k = d.get("kernels", {})
    print(os.path.basename(f)[len(tag) + 7:-5], d["steps"], d["ms_per_step"], d["ms_per_step_min"], d["ms_per_step_max"], round(d["value"]),
          d.get("frame_ms_single"), k.get("raster_ms_per_frame"), k.get("trace_ms_per_frame"), k.get("shadow_ms_per_frame"),
          k.get("compact_ms_per_frame"), k.get("shade_ms_per_frame"), d["config"].get("frames_in_flight"), d["config"].get("workspace_MB"),
          d.get("gpu_over_cpu"))
d = line(os.path.join(dst, tag + "_bench_bunny_1080p.json"))
r = d["roofline"]
print({k: r[k] for k in ("kernel", "bound", "achieved", "frac", "traffic", "algorithmic_bytes_per_launch", "avg_launch_ms", "launches_per_frame", "share_of_frame")})
print(d["roofline_valu"])
print("cpu", d["cpu_baseline"]["value"], "host boundary", d["host_boundary"],
      "single", d["single_frame"]["wall_ms"], d["single_frame"]["events_ms"])
tv = tb = 0.0
for name, v in d["kernels_profiled"]["per_launch"].items():
    print(name, v)
for mode in ("serial", "default"):
    print(mode)
    for row in list(csv.DictReader(open(os.path.join(dst, "%s_kernel_stats_bunny_1080p_%s.csv" % (tag, mode)))))[:10]:
        m = re.search(r"(\w+_kernel(<[^>]*>)?)", row["Name"])
        print("  %-34s %5s calls %8.1f us avg %8.1f min %8.1f max" % (m.group(1) if m else row["Name"][:34], row["Calls"], float(row["AverageNs"]) / 1e3,
                                                                       float(row["MinNs"]) / 1e3, float(row["MaxNs"]) / 1e3))
print("under rocprof (default):", line(os.path.join(dst, tag + "_bench_under_rocprof_default_bunny_1080p.json"))["ms_per_step"])
