#!/usr/bin/env python3
"""Overlap analysis of a rocprofv3 --kernel-trace CSV: how many kernels run at a time, how long each kernel takes when
frames overlap, and how long a stream sits idle between two of its kernels.
usage: tools/trace_overlap.py DIR_WITH_kernel_trace.csv [--last-ms 30]"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    last_ms = float(sys.argv[sys.argv.index("--last-ms") + 1]) if "--last-ms" in sys.argv else 30.0
    paths = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for path in paths:
        with open(path) as f:
            for r in csv.DictReader(f):
                m = re.search(r"(\w+_kernel(?:<[^>]*>)?)\(", r["Kernel_Name"])
                name = m.group(1) if m else r["Kernel_Name"][:40]
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Queue_Id", "?"), r.get("Stream_Id", r.get("Queue_Id", "?"))))
    rows.sort()
    if not rows:
        sys.exit("no kernel_trace.csv under %s" % root)
    t_end = max(r[1] for r in rows)
    # the window with the library's kernels in steady state: skip the tail (event pass, host boundary) by taking the busiest stretch
    lib = [r for r in rows if "_kernel" in r[2]]
    # pick the window of `last_ms` with the most kernel starts
    starts = [r[0] for r in lib]
    best, lo = 0, 0
    j = 0
    for i in range(len(starts)):
        while starts[i] - starts[j] > last_ms * 1e6:
            j += 1
        if i - j > best:
            best, lo = i - j, j
    w0, w1 = starts[lo], starts[lo] + int(last_ms * 1e6)
    win = [r for r in lib if r[0] >= w0 and r[1] <= w1]
    print("window %.1f ms, %d kernels, %d queues" % ((w1 - w0) / 1e6, len(win), len(set(r[3] for r in win))))
    # concurrency histogram (time-weighted)
    ev = []
    for s, e, *_ in win:
        ev.append((s, 1))
        ev.append((e, -1))
    ev.sort()
    cur, last, hist = 0, w0, defaultdict(int)
    for t, d in ev:
        hist[cur] += t - last
        last = t
        cur += d
    total = sum(hist.values())
    print("kernels running at a time (share of the window):", " ".join("%d:%.0f%%" % (k, 100.0 * v / total) for k, v in sorted(hist.items()) if v * 200 > total))
    print("mean concurrency %.2f" % (sum(k * v for k, v in hist.items()) / total))
    dur = defaultdict(list)
    for s, e, n, *_ in win:
        dur[n].append((e - s) / 1e3)
    print("%-28s %6s %9s %9s %9s %10s" % ("kernel", "n", "mean us", "min", "max", "sum ms"))
    for n, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        print("%-28s %6d %9.1f %9.1f %9.1f %10.2f" % (n, len(v), sum(v) / len(v), min(v), max(v), sum(v) / 1e3))
    print("sum of kernel durations / window = %.2f" % (sum(sum(v) for v in dur.values()) / 1e3 / ((w1 - w0) / 1e6)))
    # per queue: gaps between the end of one kernel and the start of the next
    byq = defaultdict(list)
    for r in win:
        byq[r[3]].append(r)
    gaps = []
    busy = []
    for q, v in byq.items():
        v.sort()
        b = 0
        for a, c in zip(v, v[1:]):
            gaps.append((c[0] - a[1]) / 1e3)
        for s, e, *_ in v:
            b += e - s
        busy.append(b / (w1 - w0))
    gaps.sort()
    if gaps:
        print("gap between consecutive kernels of a queue: median %.1f us, mean %.1f, p90 %.1f, max %.1f" % (
            gaps[len(gaps) // 2], sum(gaps) / len(gaps), gaps[int(len(gaps) * 0.9)], gaps[-1]))
    print("share of the window a queue has a kernel running: mean %.2f (min %.2f, max %.2f)" % (sum(busy) / len(busy), min(busy), max(busy)))


if __name__ == "__main__":
    main()
