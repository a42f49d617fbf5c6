#!/bin/bash
# Evidence for bench.py's roofline block, the way MI355X_MICROARCH.md prescribes: kernel trace + stats in one
# pass, FETCH_SIZE / WRITE_SIZE / SQ counters each in their own --pmc pass (kernel trace only, nothing else mixed in).
#   bash tools/collect_roofline.sh c3 [prepare] [steps]      (on the GPU box, from the repo root)
CFG=${1:-c3}; PREP=${2:-}; STEPS=${3:-30}
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${MGL_TAG:-r02}
D=$R/gpurun_out/${TAG}_roof_$CFG
export MGL_NO_AUTOBUILD=1
cd /tmp && export TMPDIR=/tmp
rm -rf $D && mkdir -p $D
ARGS="$CFG ${PREP:--1} $STEPS"
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $D/ks -- python3 $R/tools/run_state.py $ARGS > $D/ks.log 2>&1 || exit 1
timeout -k 10 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $D/fetch -- python3 $R/tools/run_state.py $ARGS > $D/fetch.log 2>&1 || exit 1
timeout -k 10 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $D/write -- python3 $R/tools/run_state.py $ARGS > $D/write.log 2>&1 || exit 1
timeout -k 10 900 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $D/sq -- python3 $R/tools/run_state.py $ARGS > $D/sq.log 2>&1 || exit 1
cd $R
python3 tools/pmc_to_json.py $D $CFG $STEPS gpurun_out/${TAG}_pmc_$CFG.json > $D/summary.txt 2>&1
# the per-kernel stats table of the same command, and the raw counter rows of the step's kernels (small)
for f in $(find $D/ks -name '*kernel_stats.csv'); do cp $f gpurun_out/${TAG}_kernel_stats_$CFG.csv; done
python3 - <<PY
import csv, glob
keep = ("k_neighbours2", "k_sim", "k_decide", "k_apply", "k_build_end", "k_snapshot", "k_neighbours(")
for sub in ("fetch", "write", "sq"):
    rows = [r for f in glob.glob("$D/" + sub + "/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f))]
    rows = [r for r in rows if any(k in r["Kernel_Name"] for k in keep)][-6000:]
    if rows:
        w = csv.DictWriter(open("gpurun_out/${TAG}_counters_${CFG}_" + sub + ".csv", "w", newline=""), fieldnames=list(rows[0].keys()))
        w.writeheader(); w.writerows(rows)
PY
find $D -type f ! -name '*.log' ! -name 'summary.txt' -delete
tail -5 $D/ks.log; head -c 1500 $D/summary.txt
