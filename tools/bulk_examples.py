#!/usr/bin/env python3
"""Diagnostic: a few improving neighbours of an early step with their windows and journals."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from megalania_amd import binding, corpus
data, _ = corpus.config_input("c2")
sa = binding.SA(data, neighbours_per_step=4096, seed=1673551, iters_per_epoch=len(data), accept="bulk")
sa.run(1)
cur, curcost = sa.current()
costs, nd, diffs = sa.neighbours(1, want_diffs=True)
win = sa.debug_dump(21, np.uint32).reshape(-1, 2).astype(np.int64)
w2 = sa.debug_dump(22, np.uint32).astype(np.int64)
soft = w2 & 0x7FFFFFFF
ok = np.nonzero((costs != binding.INVALID_COST) & (costs < curcost))[0]
shown = 0
for j in ok:
    t, e = int(win[j, 0]), int(win[j, 1]); s = int(soft[j])
    if s - t < 100: continue
    d = diffs[j][:nd[j]] if hasattr(diffs, '__getitem__') else None
    print(f"j {j} target {t} soft +{s-t} end +{e-t} gain {(curcost-int(costs[j]))/16384:.2f} B ndiffs {nd[j]}")
    if d is not None:
        for x in d[:6]:
            print("     ", x)
    # base packets around
    seg = cur[t:t+12]
    print("      base at target:", [(int(a), int(b), int(c)) for a, b, c in zip(seg['type'], seg['dist'], seg['len'])][:8])
    shown += 1
    if shown >= 6: break
