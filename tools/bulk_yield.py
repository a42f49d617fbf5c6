#!/usr/bin/env python3
"""Diagnostic: per bulk step, how many neighbours improved, how many were acceptable, how many were taken.
   python tools/bulk_yield.py c2 [steps=32]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from megalania_amd import binding, corpus
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 32
K = {"c1": 1024, "c2": 4096, "c3": 16384, "c4": 16384, "c5": 4096}[cfg]
data, _ = corpus.config_input(cfg)
sa = binding.SA(data, neighbours_per_step=K, seed=1673551, iters_per_epoch=len(data), accept="bulk")
ev = imp = acc = 0
for s in range(steps):
    st = sa.run(1)
    ev += st["evaluations"]; imp += st["improving_neighbours"]; acc += st["accepted"]
    print(f"step {s+1:3d} evals {st['evaluations']:5d} improving {st['improving_neighbours']:5d} taken {st['accepted']:5d}  ({100*st['accepted']/max(1,st['improving_neighbours']):.0f} %)  est bytes {18 + st['best_cost']/16384:.1f}  cum evals {ev} taken {acc} of improving {imp}")
