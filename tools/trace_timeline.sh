#!/bin/bash
# kernel timeline of the last steps on the state bench.py measures: tools/trace_timeline.sh c3 [steps] [rows] > profiles/rNN_timeline_c3.txt
CFG=${1:-c3}; STEPS=${2:-12}; ROWS=${3:-120}
R=$(cd "$(dirname "$0")/.." && pwd)
D=$R/gpurun_out/tl_$CFG
export MGL_NO_AUTOBUILD=1
cd /tmp && export TMPDIR=/tmp
rm -rf $D && mkdir -p $D
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $D -- python3 $R/tools/run_state.py $CFG -1 $STEPS > $D/run.log 2>&1 || { tail -5 $D/run.log; exit 1; }
python3 $R/tools/timeline.py $D $ROWS
find $D -type f ! -name '*.log' -delete
