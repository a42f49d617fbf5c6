#!/usr/bin/env python3
"""Diagnostic: the split (pick + rest) neighbour launch against the single-kernel one, cost by cost."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from megalania_amd import binding, corpus
data, _ = corpus.config_input("c2")
K = 4096
a = binding.SA(data, neighbours_per_step=K)
os.environ["MGL_NO_SPLIT"] = "1"
b = binding.SA(data, neighbours_per_step=K)
del os.environ["MGL_NO_SPLIT"]
full = binding.SA(data, neighbours_per_step=K, fullwalk=True)
for s in range(70):
    ca, _, _ = a.neighbours(s, want_diffs=False)
    cb, _, _ = b.neighbours(s, want_diffs=False)
    bad = np.nonzero(ca != cb)[0]
    if len(bad):
        cur, _ = a.current()
        full.set_slab(cur)
        cf, _, _ = full.neighbours(s, want_diffs=False)
        print("step", s, "differ", len(bad), "first", bad[:5], "split", ca[bad[:5]], "single", cb[bad[:5]], "fullwalk", cf[bad[:5]])
    sta, stb = a.run(1), b.run(1)
    if sta["current_cost"] != stb["current_cost"]:
        print("trajectories part at step", s, sta["current_cost"], stb["current_cost"]); break
print("done", sta["current_cost"], stb["current_cost"])
