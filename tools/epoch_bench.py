#!/usr/bin/env python3
"""Diagnostic: wall time of main.c:69-77's schedule (epochs of N iterations, each restarting from
the all-literal or the best slab) with and without device snapshots of the base.  GPU only."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from megalania_amd import binding, corpus

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 20
data, _ = corpus.config_input(cfg)
K = {"c1": 1024, "c2": 4096, "c3": 16384, "c5": 4096}[cfg]
steps = (len(data) + K - 1) // K
for snaps in (True, False):
    sa = binding.SA(data, neighbours_per_step=K, iters_per_epoch=len(data), snapshots=snaps)
    t0 = time.perf_counter()
    t_begin = 0.0
    for phase in range(2):
        for e in range(epochs):
            t1 = time.perf_counter()
            sa.begin_epoch(phase, from_best=phase != 0)
            t_begin += time.perf_counter() - t1
            st = sa.run(steps)
    el = time.perf_counter() - t0
    print(f"{cfg} snapshots={snaps}: {2 * epochs} epochs x {steps} steps in {el:.3f} s "
          f"(begin_epoch total {t_begin:.3f} s = {t_begin / (2 * epochs) * 1e3:.2f} ms each), best {st['best_cost']}", flush=True)
    sa.close()
