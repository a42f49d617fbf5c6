cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/ks_ev
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ks_ev -- python3 $R/tools/evolved_bench.py c2 6000 > $R/gpurun_out/ks_ev.log 2>&1
grep "^c2" $R/gpurun_out/ks_ev.log
python3 - <<PY
import csv,glob,collections
f=glob.glob("$R/gpurun_out/ks_ev/**/*kernel_trace.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
# last 2*450*~13 dispatches belong to the two measured phases; take per-kernel averages over the last 400 calls of each name
by=collections.defaultdict(list)
for r in rows:
    by[r["Kernel_Name"][:44]].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
for k,v in by.items():
    if "k_neighbours" in k or "k_apply" in k or "k_decide" in k:
        print(f"{k:46s} calls={len(v):6d} last-400 avg={sum(v[-400:])/len(v[-400:])/1000:8.1f} us  max={max(v[-400:])/1000:8.1f}")
PY
