#!/usr/bin/env python3
"""Profiling target: bring one chain to the state bench.py measures (prepare steps in the library's default
accept mode, from the all-literal slab), then run `steps` more steps.
  rocprofv3 ... -- python3 tools/run_state.py c3 [prepare] [steps] [K]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from megalania_amd import binding, corpus

DEFAULT_K = {"c1": 1024, "c2": 4096, "c3": 16384, "c4": 16384, "c5": 4096}
PREPARE_CAP = {"c1": 400, "c2": 1500, "c3": 6000, "c4": 1500, "c5": 2000}
cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
prepare = int(sys.argv[2]) if len(sys.argv) > 2 else -1  # -1: like bench.py, until the bulk phase is over
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
K = int(sys.argv[4]) if len(sys.argv) > 4 else DEFAULT_K[cfg]
data, desc = corpus.config_input(cfg)
props = dict(pb=2, max_bucket_scan=4096) if cfg == "c5" else {}
sa = binding.SA(data, neighbours_per_step=K, seed=1673551, iters_per_epoch=len(data), timing=not os.environ.get("MGL_RUN_NOTIMING"), **props)
if os.environ.get("MGL_RUN_GREEDY"):  # the state bench.py measures above 32 MB: the greedy seed, no search before the timed steps
    sa.seed_greedy(int(os.environ["MGL_RUN_GREEDY"]))
    sa.set_accept_mode("single")
done = 0
while prepare != 0 and done < (prepare if prepare > 0 else PREPARE_CAP[cfg]):
    p = sa.run(prepare if prepare > 0 else (64 if len(data) <= (1 << 20) else 128))
    done += p["steps"]
    print("prepare", done, {k: p[k] for k in ("accepted", "bulk_steps", "best_cost", "packets")}, flush=True)
    if prepare < 0 and p["bulk_steps"] * 16 <= p["steps"]:
        break
if os.environ.get("MGL_RUN_ACCEPT"):  # the accept mode of the measured steps (default: the library's, auto)
    sa.set_accept_mode(os.environ["MGL_RUN_ACCEPT"])
if os.environ.get("MGL_RUN_WARM"):
    sa.run(int(os.environ["MGL_RUN_WARM"]))
st = sa.run(steps)
print("batch accepts / fallbacks", sa.batch_counters())
print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in st.items()})
sa.close()
