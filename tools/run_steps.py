#!/usr/bin/env python3
"""Run `steps` SA steps after `warm` warm-up steps (profiling target: rocprofv3 -- python3 tools/run_steps.py c2 40 8)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from megalania_amd import binding, corpus
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 0
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
K = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
full = len(sys.argv) > 5 and sys.argv[5] == "fullwalk"
data, desc = corpus.config_input(cfg)
sa = binding.SA(data, neighbours_per_step=K, timing=True, fullwalk=full, **({"pb": 2, "max_bucket_scan": 4096} if cfg == "c5" else {}))
if warm:
    sa.run(warm)
st = sa.run(steps)
print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in st.items()})
sa.close()
