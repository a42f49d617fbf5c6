#!/usr/bin/env python3
"""Print a rocprofv3 kernel_stats.csv compactly (name, calls, average us), optionally only names containing a pattern.
   python tools/kstats.py <dir> [pattern]"""
import csv, glob, re, sys
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        nm = re.sub(r"\(.*", "", r["Name"]).replace("void ", "")
        if pat in nm:
            print(f"{nm:40s} {int(r['Calls']):6d} calls  avg {float(r['AverageNs'])/1e3:9.1f} us  total {float(r['TotalDurationNs'])/1e6:9.1f} ms")
