#!/bin/bash
# A/B of one environment switch on the state bench.py measures: tools/ab_env.sh c3 MGL_NO_CONT [steps]
# (prints the last stats line of tools/run_state.py with the switch unset and set)
CFG=${1:-c3}; VAR=${2:-MGL_NO_CONT}; STEPS=${3:-60}
R=$(cd "$(dirname "$0")/.." && pwd)
export MGL_NO_AUTOBUILD=1
for v in "" 1; do
  if [ -z "$v" ]; then unset $VAR; else export $VAR=1; fi
  echo "== $VAR=${v:-unset}"
  timeout -k 10 300 python3 $R/tools/run_state.py $CFG -1 $STEPS | tail -1
done
