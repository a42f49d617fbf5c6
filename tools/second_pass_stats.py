#!/usr/bin/env python3
"""Who goes to the second pass on an evolved slab: repair picks vs list overflows; window sizes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from megalania_amd import binding, corpus
cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
prep = int(sys.argv[2]) if len(sys.argv) > 2 else 700
K = {"c1": 1024, "c2": 4096, "c3": 16384, "c4": 16384, "c5": 4096}[cfg]
data, _ = corpus.config_input(cfg)
sa = binding.SA(data, neighbours_per_step=K, iters_per_epoch=len(data), accept="bulk")
sa.run(prep)
sa.set_accept_mode("single")
tot = np.zeros(4, dtype=np.int64)
for s in range(20):
    sa.run(1)
    tot += sa.debug_dump(10, np.uint32)[:4].astype(np.int64)
print(cfg, "per step: second-pass list", tot[0] / 20, "last-resort list", tot[1] / 20, "spill slots", tot[2] / 20, "repair picks", tot[3] / 20)
costs, nd, _ = sa.neighbours(10**6, want_diffs=False)
win = sa.debug_dump(21, np.uint32).reshape(-1, 2)
ok = costs != binding.INVALID_COST
w = (win[ok, 1].astype(np.int64) - win[ok, 0].astype(np.int64))
print("window bytes: mean %.0f p50 %.0f p90 %.0f p99 %.0f max %d" % (w.mean(), np.percentile(w, 50), np.percentile(w, 90), np.percentile(w, 99), w.max()))
sa.close()
