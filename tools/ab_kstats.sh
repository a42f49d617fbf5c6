#!/bin/bash
# Diagnostic: per-kernel averages (rocprofv3 --stats) of the same short run under several builds (megalania_amd/_build/ab/*.so).
#   bash tools/ab_kstats.sh c3 "300 4" pb_ckpt
CFG=${1:-c3}; ARGS=${2:-"300 4"}; PAT=${3:-pb_}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export MGL_NO_AUTOBUILD=1
cd /tmp && export TMPDIR=/tmp
for so in $R/megalania_amd/_build/ab/*.so; do
  rm -rf /tmp/kt_ab; export MGL_HIP_SO=$so
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_ab -- python3 $R/tools/run_state.py $CFG $ARGS > /tmp/kt_ab.log 2>&1
  echo "== $(basename $so)"; python3 $R/tools/kstats.py /tmp/kt_ab $PAT
done
