#!/usr/bin/env python3
"""Diagnostic: slowest-workgroup cycles per stage of k_apply_chains (mgl_debug_set key 0 = 50).  GPU only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from megalania_amd import binding, corpus
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 300
data, _ = corpus.config_input(cfg)
K = {"c1": 1024, "c2": 4096, "c3": 16384, "c5": 4096}[cfg]
sa = binding.SA(data, neighbours_per_step=K, timing=True)
sa.run(warm)
sa.L.mgl_debug_set(sa.h, 0, 50)
names = ["1 gather this context's events", "2a search + stage window", "2b serial re-simulation", "3 job emission", "4 checkpoint patches"]
acc = np.zeros(5)
walk = []
steps = 40
for _ in range(steps):
    st = sa.run(1)
    h = sa.debug_dump(14, np.uint32)
    acc += h[8:13]
    walk.append([int(h[13]), int(h[14]), int(h[15]) & 0xFFFFF, int(h[15]) >> 20])
    # the counters are maxima: clear them through a fresh step's header reset (k_apply_walk rewrites hdr[0..7] only)
print(f"{cfg}: touched contexts (last step) {int(h[2])}, inserted {int(h[0])}, removed {int(h[1])}, jobs B {int(h[4])}, jobs C {int(h[5])}, span {int(h[6])}, saved entries {int(h[7])}")
print("running maxima over the run (cycles):")
for n, v in zip(names, h[8:13]):
    print(f"  {n:34s} {int(v):9d}")
print("apply ms avg:", st["gpu_ms_rebuild"])
w = np.array(walk, dtype=np.float64)
print("k_apply_walk per accept (cycles): state lookup %.0f, walk %.0f (%.1f iterations, %.0f cycles each), context listing + end %.0f" % (w[:,0].mean(), w[:,1].mean(), w[:,3].mean(), w[:,1].mean() / max(1.0, w[:,3].mean()), w[:,2].mean()))
print("   medians: state %d, walk %d cycles, %d iterations; max iterations %d; per-iteration median %.0f cycles" % (np.median(w[:,0]), np.median(w[:,1]), np.median(w[:,3]), w[:,3].max(), np.median(w[:,1] / np.maximum(1, w[:,3]))))
