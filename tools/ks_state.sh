#!/bin/bash
# kernel-trace summary of the last steps of tools/run_state.py:  bash tools/ks_state.sh c3 [prepare] [steps]
CFG=${1:-c3}; PREP=${2:--1}; STEPS=${3:-30}
R=${GRAFT_REPO_ROOT:-$(pwd)}
D=$R/gpurun_out/ks_state_$CFG
export MGL_NO_AUTOBUILD=1
cd /tmp && export TMPDIR=/tmp
rm -rf $D && mkdir -p $D
timeout -k 10 900 rocprofv3 --kernel-trace --output-format csv -d $D/ks -- python3 $R/tools/run_state.py $CFG $PREP $STEPS > $D/ks.log 2>&1
cd $R
python3 tools/pmc_to_json.py $D $CFG $STEPS $D/summary.json > /dev/null 2>&1
python3 - <<PY
import json
d = json.load(open("$D/summary.json"))
print("$CFG: span per step %.1f us" % d["span_us_per_step"])
for k, v in d["kernels"].items():
    print(f'{k[:50]:50s} x{v["launches_per_step"]:.2f} avg {v["avg_us"]:9.1f} us  per step {v["us_per_step"]:9.1f} us')
PY
tail -2 $D/ks.log
find $D/ks -type f -delete
