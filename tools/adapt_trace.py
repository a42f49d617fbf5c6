#!/usr/bin/env python3
"""Diagnostic: which form of the neighbour launch the device picks over a run, with its measurements."""
import os, sys, struct
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from megalania_amd import binding, corpus
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
data, _ = corpus.config_input(cfg)
K = {"c2": 4096, "c3": 16384}[cfg]
sa = binding.SA(data, neighbours_per_step=K, timing=True)
for r in range(48):
    st = sa.run(25)
    raw = sa.debug_dump(16, np.uint8).tobytes()
    nbr_single, mode_steps, probing, e_clean, e_dirty, e_single, p_dirty = struct.unpack_from("<7I", raw, 116)
    print(f"step {(r+1)*25:5d} single={nbr_single} probing={probing} mode_steps={mode_steps} clean={e_clean/100:.0f}us dirty={e_dirty/100:.0f}us single={e_single/100:.0f}us p={p_dirty/65536:.2f} nbr_ms={st['gpu_ms_neighbours']/25:.3f} second={st['second_pass_neighbours']}")
