#!/usr/bin/env python3
"""Diagnostic: the last kernels of a rocprofv3 --kernel-trace run as a timeline (start offset, duration, queue).
   python tools/timeline.py <dir with *kernel_trace.csv> [rows=150]"""
import csv, glob, re, sys
d = sys.argv[1]; rows_n = int(sys.argv[2]) if len(sys.argv) > 2 else 150
rows = [r for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True) for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-rows_n:]
t0 = int(rows[0]["Start_Timestamp"])
def short(n):
    n = re.sub(r"\(.*", "", n).replace("void ", "")
    return n
for r in rows:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{s/1000:10.1f} {(e-s)/1000:8.1f} us  q{r.get('Queue_Id','?'):>3}  grid {r.get('Grid_Size','?'):>8} wg {r.get('Workgroup_Size','?'):>4}  {short(r['Kernel_Name'])}")
