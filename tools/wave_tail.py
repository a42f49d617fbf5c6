#!/usr/bin/env python3
"""Diagnostic: per-neighbour wavefront lifetime in the neighbour kernels (MGL_F_PROFILE) on an evolved slab -- the
distribution, and what the slowest ones have in common (change events, packets walked).
   python tools/wave_tail.py c2 [steps_before]"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from megalania_amd import binding, corpus
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
K = {"c1": 1024, "c2": 4096, "c3": 16384, "c4": 16384, "c5": 4096}[cfg]
data, desc = corpus.config_input(cfg)
sa = binding.SA(data, neighbours_per_step=K, timing=True, iters_per_epoch=len(data), flags=binding.F_PROFILE)
done = 0
while done < 6000:
    p = sa.run(128); done += p["steps"]
    if p["bulk_steps"] == 0: break
sa.set_accept_mode("single")
sa.run(8)
rows = []
for rep in range(8):
    st = sa.run(1)
    raw = sa.debug_dump(9, np.uint64)[32:32 + K].copy()
    life = (raw & np.uint64(0xFFFFFFFFFF)).astype(np.float64)
    ev = ((raw >> np.uint64(40)) & np.uint64(0xFFF)).astype(np.int64)
    pk = ((raw >> np.uint64(52)) & np.uint64(0xFFF)).astype(np.int64)
    rows.append((life, ev, pk, st["gpu_ms_neighbours"]))
life = np.concatenate([r[0] for r in rows]); ev = np.concatenate([r[1] for r in rows]); pk = np.concatenate([r[2] for r in rows])
ok = life > 0
life, ev, pk = life[ok], ev[ok], pk[ok]
out = dict(config=cfg, steps_before=done, neighbours=int(ok.sum()), ms_per_step=float(np.mean([r[3] for r in rows])),
           lifetime_cycles={f"p{q}": float(np.percentile(life, q)) for q in (10, 50, 90, 99, 99.9, 100)}, mean=float(life.mean()))
edges = [0, 1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 4096]
out["by_events"] = []
for lo, hi in zip(edges, edges[1:]):
    m = (ev >= lo) & (ev < hi)
    if m.any():
        out["by_events"].append(dict(events=f"{lo}-{hi-1}", n=int(m.sum()), mean_cycles=float(life[m].mean()), max_cycles=float(life[m].max())))
out["by_packets_walked"] = []
for lo, hi in zip(edges, edges[1:]):
    m = (pk >= lo) & (pk < hi)
    if m.any():
        out["by_packets_walked"].append(dict(packets=f"{lo}-{hi-1}", n=int(m.sum()), mean_cycles=float(life[m].mean()), max_cycles=float(life[m].max())))
hist, bins = np.histogram(life, bins=20)
out["histogram"] = dict(counts=hist.tolist(), edges=[float(b) for b in bins])
print(json.dumps(out, indent=1))
