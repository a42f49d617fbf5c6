#!/usr/bin/env python3
"""Diagnostic: per bulk step, how many of the K evaluations improve on the base and how many of those the step takes
(the rest lose to an overlapping neighbour of smaller cost).  GPU only.
  python tools/step_yield.py c2 [steps=60]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from megalania_amd import binding, corpus
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
K = int(os.environ.get("MGL_K", {"c1": 1024, "c2": 4096, "c3": 16384, "c4": 16384, "c5": 4096}[cfg]))
data, _ = corpus.config_input(cfg)
sa = binding.SA(data, neighbours_per_step=K, iters_per_epoch=len(data), accept=os.environ.get("MGL_RUN_ACCEPT", "bulk"))
ev = tot_drop = tot_fail = 0
every = int(os.environ.get("MGL_EVERY", "1"))
for s in range(steps):
    st = sa.run(1)
    ev += st["evaluations"]
    tot_drop += st["dropped_neighbours"]; tot_fail += st["failed"]
    if every > 1 and s % every:
        continue
    print(f"step {s:4d} evals {ev:8d} failed {tot_fail:6d} dropped {tot_drop:5d} improving {st['improving_neighbours']:5d} taken {st['accepted']:5d} "
          f"yield {st['accepted'] / max(1, st['improving_neighbours']):.2f} est {18 + st['best_cost'] / 16384:9.1f}", flush=True)
