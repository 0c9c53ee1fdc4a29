#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: mean counter value per dispatch, grouped by kernel.
usage: tools/pmc_summary.py DIR [DIR...]   (directories written by tools/pmc_passes.sh)"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    for key in ("shadow_walk_kernel", "shade_compact_kernel", "walk_steal_kernel", "walk_kernel", "shadow_grid_kernel",
                "face_raster_kernel", "raster_kernel", "bin_count_kernel", "bin_fill_kernel", "bin_offsets_kernel", "compact_kernel", "shade_kernel",
                "prep_kernel"):
        if key in name:
            args = name.split(key)[1].split("(")[0]
            return key + args
    return None


def main():
    acc = defaultdict(lambda: defaultdict(list))
    for d in sys.argv[1:]:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            per_dispatch = defaultdict(float)
            names = {}
            with open(path) as f:
                for row in csv.DictReader(f):
                    k = short(row["Kernel_Name"])
                    if k is None:
                        continue
                    key = (row["Dispatch_Id"], row["Counter_Name"])
                    per_dispatch[key] += float(row["Counter_Value"])
                    names[row["Dispatch_Id"]] = k
            for (disp, cname), v in per_dispatch.items():
                acc[names[disp]][cname].append(v)
    for kern in sorted(acc):
        print(kern)
        for cname in sorted(acc[kern]):
            vals = acc[kern][cname]
            print("    %-40s mean %16.1f   (n=%d)" % (cname, sum(vals) / len(vals), len(vals)))


if __name__ == "__main__":
    main()
