cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pb_c2 -- python3 $R/tools/build_bench.py c2 50 0 > $R/gpurun_out/pb_c2.log 2>&1 && cat $R/gpurun_out/pb_c2.log | tail -4 && python3 - <<PY
import csv,glob
for f in glob.glob("$R/gpurun_out/pb_c2/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Name"].startswith("pb_") or "k_snapshot" in r["Name"]: print(r["Name"][:40], r["Calls"], r["AverageNs"], r["MaxNs"])
PY
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pb_c3 -- python3 $R/tools/build_bench.py c3 10 0 > $R/gpurun_out/pb_c3.log 2>&1 ; cat $R/gpurun_out/pb_c3.log | tail -4 && python3 - <<PY
import csv,glob
for f in glob.glob("$R/gpurun_out/pb_c3/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Name"].startswith("pb_") or "k_snapshot" in r["Name"] or "k_apply" in r["Name"] or "k_neigh" in r["Name"]: print(r["Name"][:40], r["Calls"], r["AverageNs"], r["MaxNs"])
PY
