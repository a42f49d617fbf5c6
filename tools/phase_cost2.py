#!/usr/bin/env python3
"""Diagnostic: cost of each stage of the neighbour evaluation on an evolved slab, by stopping the kernels early
(mgl_debug_set key 0) on a fixed base: nothing is accepted while a stop is set.  GPU only.
  python tools/phase_cost2.py c3 [prepare=-1]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from megalania_amd import binding, corpus
cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
prepare = int(sys.argv[2]) if len(sys.argv) > 2 else -1
K = {"c1": 1024, "c2": 4096, "c3": 16384, "c4": 16384, "c5": 4096}[cfg]
data, desc = corpus.config_input(cfg)
props = dict(pb=2, max_bucket_scan=4096) if cfg == "c5" else {}
sa = binding.SA(data, neighbours_per_step=K, timing=True, iters_per_epoch=len(data), **props)
done = 0
while prepare != 0 and done < (prepare if prepare > 0 else 6000):
    p = sa.run(prepare if prepare > 0 else 128)
    done += p["steps"]
    if (prepare < 0 and p["bulk_steps"] == 0) or prepare > 0:
        break
print(f"{cfg}: after {done} steps, {p['packets'] if prepare else '?'} packets")
sa.set_accept_mode("single")
names = {1: "target + state_at", 2: "+ model_at (checkpoint + replay)", 31: "  top-K: price tables", 32: "  top-K: + literal/short-rep", 33: "  top-K: + bucket bounds",
         34: "  top-K: + runs of the deeper orders", 35: "  top-K: + rep pass", 36: "  top-K: + 16-byte run", 37: "  top-K: + 8-byte run", 38: "  top-K: + 4-byte run", 39: "  top-K: all sources, short ones filter only",
         3: "+ top-K complete / mutate", 4: "+ window walk", 41: "  re-simulation: distinct contexts listed", 42: "  re-simulation: + chain search, first chunk", 43: "  re-simulation: + merge part", 0: "+ chain_sim (everything)"}
prev = 0.0
stops = (4, 41, 42, 43, 0) if os.environ.get("MGL_SIM_ONLY") else (1, 2, 31, 32, 33, 34, 35, 36, 37, 39, 38, 3, 4, 41, 42, 43, 0)
for stop in stops:
    sa.L.mgl_debug_set(sa.h, 0, stop)
    sa.run(2)
    st = sa.run(10)
    ms = st["gpu_ms_neighbours"] / 10
    sim = st["gpu_ms_sim"] / max(1, st["sim_launches"]) * 1000
    print(f"stop={stop:2d} {names[stop]:40s} neighbour kernels {ms*1000:8.1f} us   (+{(ms-prev)*1000:8.1f})   k_sim avg launch {sim:7.1f} us   accepted={st['accepted']}", flush=True)
    prev = ms
