#!/bin/bash
# Diagnostic: the same state and steps under several builds of the library (megalania_amd/_build/ab/*.so).
#   bash tools/ab_builds.sh c3 [steps] [reps]
CFG=${1:-c3}; STEPS=${2:-40}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export MGL_NO_AUTOBUILD=1
for so in $R/megalania_amd/_build/ab/*.so; do
  for rep in $(seq 1 ${3:-1}); do
    MGL_HIP_SO=$so timeout -k 10 300 python3 $R/tools/run_state.py $CFG -1 $STEPS | tail -1 | python3 -c "
import sys, ast
d = ast.literal_eval(sys.stdin.read())
print('$(basename $so)', 'ms/step', round(d['gpu_ms_total'] / d['steps'], 4), 'nbr', round(d['gpu_ms_neighbours'] / d['steps'], 4), 'cost', d['best_cost'], '2nd', d['second_pass_neighbours'])"
  done
done
