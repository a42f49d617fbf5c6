#!/bin/bash
# Diagnostic: the same state and steps under several values of one environment knob of the library.
#   bash tools/ab_env.sh c3 MGL_PICK_WAVES "1 2 4 8" [steps]
CFG=${1:-c3}; VAR=$2; VALS=$3; STEPS=${4:-40}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export MGL_NO_AUTOBUILD=1
for v in $VALS; do
  env $VAR=$v timeout -k 10 300 python3 $R/tools/run_state.py $CFG -1 $STEPS | tail -1 | python3 -c "
import sys, ast
d = ast.literal_eval(sys.stdin.read())
print('$VAR=$v', 'ms/step', round(d['gpu_ms_total'] / d['steps'], 4), 'nbr', round(d['gpu_ms_neighbours'] / d['steps'], 4), 'cost', d['best_cost'], '2nd', d['second_pass_neighbours'])"
done
