#!/usr/bin/env python3
"""Diagnostic: time to derive the base structures of a slab (mgl_sa_set_slab = upload + build)
with the block-parallel builder and with the one-wavefront builder.  GPU only."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from megalania_amd import binding, corpus

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
serial_too = (sys.argv[3] != "0") if len(sys.argv) > 3 else True
data, _ = corpus.config_input(cfg)
K = {"c1": 1024, "c2": 4096, "c3": 16384, "c5": 4096}[cfg]
props = dict(pb=2, max_bucket_scan=4096) if cfg == "c5" else {}
t0 = time.perf_counter()
sa = binding.SA(data, neighbours_per_step=K, **props)
print(f"{cfg}: create (parallel build) {time.perf_counter() - t0:.3f} s", flush=True)
lit, _ = sa.current()
sa.run(steps)
ev, cost = sa.current()
for name, slab in (("all-literal", lit), (f"after {steps} steps", ev)):
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        sa.set_slab(slab)
        ts.append(time.perf_counter() - t0)
    import numpy as np
    acc = sa.debug_dump(11, np.uint64)
    print(f"{cfg} {name}: parallel build set_slab {min(ts) * 1e3:.2f} ms (chain segments redone serially: {int(acc[7])})", flush=True)
assert sa.current()[1] == cost
sa.close()
if serial_too:
    t0 = time.perf_counter()
    sb = binding.SA(data, neighbours_per_step=8, serial_build=True, snapshots=False, **props)
    print(f"{cfg}: create (serial build) {time.perf_counter() - t0:.3f} s", flush=True)
    for name, slab in (("all-literal", lit), (f"after {steps} steps", ev)):
        t0 = time.perf_counter()
        sb.set_slab(slab)
        print(f"{cfg} {name}: serial build set_slab {(time.perf_counter() - t0) * 1e3:.2f} ms", flush=True)
    assert sb.current()[1] == cost
    sb.close()
