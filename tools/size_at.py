#!/usr/bin/env python3
"""Estimated stream size of a fresh chain at given step counts, for a few seeds (the equal-evaluations comparison with
tests/golden/reference_curve_c2.json):   python tools/size_at.py c2 [seeds=3] [accept=auto]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from megalania_amd import binding, corpus
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
accept = sys.argv[3] if len(sys.argv) > 3 else "auto"
K0 = {"c1": 1024, "c2": 4096, "c3": 16384}[cfg]
K = int(os.environ.get("MGL_K", K0))  # (MGL_K: another step size at the same evaluation marks)
marks = [m * K0 // K for m in ([8, 25, 74, 245] if cfg == "c2" else [62, 256])]
data, _ = corpus.config_input(cfg)
rows = []
for sd in range(seeds):
    sa = binding.SA(data, neighbours_per_step=K, seed=1673551 + 7919 * sd, iters_per_epoch=len(data), accept=accept)
    done, ev, row = 0, 0, []
    for m in marks:
        st = sa.run(m - done); done = m; ev += st["evaluations"]
        row.append((ev, 18 + st["best_cost"] / 16384))
    rows.append(row)
    sa.close()
for i, m in enumerate(marks):
    v = [r[i][1] for r in rows]
    print(f"{cfg} K={K} {accept} steps {m:4d} evaluations {rows[0][i][0]:8d}: mean {sum(v)/len(v):9.1f}  min {min(v):9.1f} max {max(v):9.1f}")
