cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CFG=${1:-c2}; WARM=${2:-300}; N=${3:-200}
rm -rf $R/gpurun_out/ks_$CFG
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ks_$CFG -- python3 $R/tools/run_steps.py $CFG $WARM $N > $R/gpurun_out/ks_$CFG.log 2>&1
python3 - <<PY
import csv,glob
for f in glob.glob("$R/gpurun_out/ks_$CFG/**/*kernel_stats.csv", recursive=True):
    rows=list(csv.DictReader(open(f)))
    rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
    for r in rows[:16]: print(f'{r["Name"][:60]:60s} calls={r["Calls"]:>6s} avg={float(r["AverageNs"])/1000:9.1f}us pct={r["Percentage"]}')
PY
