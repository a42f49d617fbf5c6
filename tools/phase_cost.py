#!/usr/bin/env python3
"""Diagnostic: true cost of each phase of the neighbour kernel, by stopping it early
(mgl_debug_set key 0) on a fixed base.  GPU only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from megalania_amd import binding, corpus
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 300
data, desc = corpus.config_input(cfg)
sa = binding.SA(data, neighbours_per_step=4096, timing=True)
sa.run(warm)
names = {41: "  chain_sim: context listing", 42: "  chain_sim: + cursors, lower_bound", 43: "  chain_sim: + part 1 (merge)", 31: "  top-K: price tables", 32: "  top-K: + literal/short-rep", 33: "  top-K: + bucket bounds", 37: "  top-K: + bucket entry loads only", 38: "  top-K: + prices, no match extension", 34: "  top-K: + per-hit extension/prices", 35: "  top-K: + candidates, no offers", 36: "  top-K: + offers that never qualify", 1: "target + state_at", 2: "+ model_at (checkpoint + replay)", 3: "+ top-K / mutate", 4: "+ window walk", 0: "+ chain_sim (full kernel)"}
prev = 0.0
for stop in (1, 2, 31, 32, 33, 37, 38, 34, 35, 36, 3, 4, 41, 42, 43, 0):
    sa.L.mgl_debug_set(sa.h, 0, stop)
    sa.run(3)
    st = sa.run(20)
    ms = st["gpu_ms_neighbours"] / 20
    print(f"stop={stop} {names[stop]:36s} kernel {ms*1000:8.1f} us   (+{(ms-prev)*1000:7.1f})   accepted={st['accepted']}")
    prev = ms
