#!/usr/bin/env python3
"""Diagnostic: how far an accepted move's change of the adaptive model reaches (Control::mod_lo/mod_hi) on an evolved slab.
   python tools/mod_range.py c3 [steps=200]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from megalania_amd import binding, corpus
cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
K = {"c1": 1024, "c2": 4096, "c3": 16384, "c4": 16384, "c5": 4096}[cfg]
data, _ = corpus.config_input(cfg)
n = len(data)
sa = binding.SA(data, neighbours_per_step=K, iters_per_epoch=n)
done = 0
while done < 6000:
    p = sa.run(128); done += p["steps"]
    if p["bulk_steps"] == 0: break
sa.set_accept_mode("single")
fr, inf, acc = [], 0, 0
for s in range(steps):
    st = sa.run(1)
    if not st["accepted"]: continue
    raw = sa.debug_dump(16, np.uint32)
    lo, hi = int(raw[-2]), int(raw[-1])
    if lo == 0xFFFFFFFF: continue
    acc += 1
    if hi == 0xFFFFFFFF: inf += 1; hi = n
    fr.append((min(hi, n) - lo) / n)
fr = np.array(fr)
print(f"{cfg}: {acc} accepted moves after {done} steps; model change reaches the end of the file in {inf}; changed share of the file: mean {fr.mean():.3f} p50 {np.percentile(fr,50):.3f} p90 {np.percentile(fr,90):.3f}")
print("share of later targets left untouched (finite cases): mean %.3f" % (1 - np.mean([f for f in fr])))
