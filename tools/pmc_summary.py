#!/usr/bin/env python3
"""Average PMC counter values per dispatch for one kernel from rocprofv3 --pmc CSV output."""
import csv, collections, glob, sys
pat = sys.argv[2] if len(sys.argv) > 2 else "k_neighbours2<false>"
for f in sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)):
    rows = [r for r in csv.DictReader(open(f)) if pat in r["Kernel_Name"]]
    if not rows:
        continue
    agg = collections.defaultdict(float)
    disp = set()
    for r in rows:
        agg[r["Counter_Name"]] += float(r["Counter_Value"])
        disp.add(r["Dispatch_Id"])
    n = len(disp)
    print(f, "dispatches", n, "VGPR", rows[0]["VGPR_Count"], "SGPR", rows[0]["SGPR_Count"], "LDS", rows[0]["LDS_Block_Size"], "WG", rows[0]["Workgroup_Size"])
    for c, v in sorted(agg.items()):
        print(f"   {c:28s} {v / n:16.1f}")
