#!/usr/bin/env python3
"""Profiling target: bring one chain to the state bench.py measures (prepare steps in the library's default
accept mode, from the all-literal slab), then run `steps` more steps.
  rocprofv3 ... -- python3 tools/run_state.py c3 [prepare] [steps] [K]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from megalania_amd import binding, corpus

DEFAULT_K = {"c1": 1024, "c2": 4096, "c3": 16384, "c4": 16384, "c5": 4096}
DEFAULT_PREPARE = {"c1": 64, "c2": 400, "c3": 700, "c4": 700, "c5": 400}
cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
prepare = int(sys.argv[2]) if len(sys.argv) > 2 and int(sys.argv[2]) >= 0 else DEFAULT_PREPARE[cfg]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
K = int(sys.argv[4]) if len(sys.argv) > 4 else DEFAULT_K[cfg]
data, desc = corpus.config_input(cfg)
props = dict(pb=2, max_bucket_scan=4096) if cfg == "c5" else {}
sa = binding.SA(data, neighbours_per_step=K, seed=1673551, iters_per_epoch=len(data), timing=True, **props)
if prepare:
    p = sa.run(prepare)
    print("prepare", {k: p[k] for k in ("steps", "accepted", "bulk_steps", "best_cost", "packets")}, flush=True)
st = sa.run(steps)
print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in st.items()})
sa.close()
