#!/bin/bash
# rocprofv3 --kernel-trace --stats over tools/run_state.py: per-kernel totals of a whole run (set-up phase included)
#   bash tools/kstats.sh c3 [prepare] [steps] > profiles/rNN_kernel_stats_c3.txt      (on the GPU box, from the repo root)
CFG=${1:-c3}; PREP=${2:--1}; STEPS=${3:-30}
R=$(cd "$(dirname "$0")/.." && pwd)
D=$R/gpurun_out/ks_$CFG
export MGL_NO_AUTOBUILD=1
cd /tmp && export TMPDIR=/tmp
rm -rf $D && mkdir -p $D
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $R/tools/run_state.py $CFG $PREP $STEPS > $D/run.log 2>&1 || { tail -5 $D/run.log; exit 1; }
tail -2 $D/run.log
python3 - <<PY
import csv, glob
for f in glob.glob("$D/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:45]:
        print(f"{r['Name'][:64]:64s} calls={int(r['Calls']):7d} total_ms={float(r['TotalDurationNs'])/1e6:9.1f} avg_us={float(r['AverageNs'])/1e3:9.1f} pct={float(r['Percentage']):6.2f}")
PY
find $D -type f ! -name '*.log' -delete
