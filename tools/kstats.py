#!/usr/bin/env python3
"""Print a rocprofv3 --kernel-trace --stats summary (kernel_stats.csv) as a short table."""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r['Name']
    m = re.search(r'(\w+_kernel(<[^>]*>)?)', n)
    print((m.group(1) if m else n[:40]).ljust(34), r['Calls'].rjust(5), '%9.1f us avg' % (float(r['AverageNs']) / 1e3), '%9.1f min' % (float(r['MinNs']) / 1e3), '%9.1f max' % (float(r['MaxNs']) / 1e3), '%6.2f%%' % float(r['Percentage']))
