#!/usr/bin/env python3
"""Diagnostic: spread of the search result over seeds at fixed evaluation budgets (bulk mode).
   python tools/seed_spread.py c2 [seeds=6]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from megalania_amd import binding, corpus
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
nseeds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
mode = sys.argv[3] if len(sys.argv) > 3 else "bulk"
K = {"c1": 1024, "c2": 4096, "c3": 16384}[cfg]
data, _ = corpus.config_input(cfg)
marks = [25, 74, 245, 490]
res = {m: [] for m in marks}
for seed in range(1, nseeds + 1):
    sa = binding.SA(data, neighbours_per_step=K, seed=seed * 7919, iters_per_epoch=len(data), accept=mode)
    done = 0; rb = 0
    for m in marks:
        st = sa.run(m - done); done = m; rb += st["bulk_rollbacks"]
        res[m].append(18 + st["best_cost"] / 16384)
    sa.close()
    print("seed", seed, [round(res[m][-1], 1) for m in marks], "rollbacks", rb, flush=True)
for m in marks:
    a = np.array(res[m]); print(f"steps {m} (~{m*K*0.98/1e3:.0f} K evals): mean {a.mean():.1f} min {a.min():.1f} max {a.max():.1f}")
