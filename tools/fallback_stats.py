#!/usr/bin/env python3
"""Diagnostic: how many neighbours per step fall back to the full-walk kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from megalania_amd import binding, corpus
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
data, desc = corpus.config_input(cfg)
sa = binding.SA(data, neighbours_per_step=4096, timing=True)
fb = []
for s in range(steps):
    st = sa.run(1)
    cnt = sa.debug_dump(10, np.uint32)
    fb.append((int(cnt[0]), int(cnt[1]), int(cnt[2]), round(st["gpu_ms_neighbours"], 2), st["packets"], st["failed"]))
a = np.array(fb, dtype=np.float64)
print("columns: first-pass overflows, second-pass overflows, spill slots used, kernel ms, packets, failed")
for lo in range(0, steps, 20):
    blk = a[lo:lo + 20]
    print(lo, "mean", np.round(blk.mean(axis=0), 2), "max", blk.max(axis=0))
