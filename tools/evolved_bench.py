#!/usr/bin/env python3
"""Diagnostic: step time on an SA-evolved slab (many matches / reps on the walk), split launch vs
the one-kernel form.  GPU only.   python tools/evolved_bench.py c2 20000"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from megalania_amd import binding, corpus
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
data, _ = corpus.config_input(cfg)
K = {"c1": 1024, "c2": 4096, "c3": 16384, "c5": 4096}[cfg]
sa = binding.SA(data, neighbours_per_step=K, iters_per_epoch=max(len(data), warm), timing=True)
sa.run(warm)
cur, cost = sa.current()
sa.close()
for mode in ("split", "single"):
    if mode == "single":
        os.environ["MGL_NO_SPLIT"] = "1"
    s = binding.SA(data, neighbours_per_step=K, iters_per_epoch=max(len(data), warm), timing=True)
    s.set_slab(cur)
    s.run(50)
    t0 = time.perf_counter()
    st = s.run(400)
    el = time.perf_counter() - t0
    print(f"{cfg} after {warm} steps ({st['packets']} packets) {mode}: {el / 400 * 1e3:.3f} ms/step, neighbours {st['gpu_ms_neighbours'] / 400:.3f} ms, "
          f"apply {st['gpu_ms_rebuild'] / 400:.3f} ms, second pass {st['second_pass_neighbours'] / 400:.1f}/step, accepted {st['accepted']}, evals {st['evaluations'] / 400:.0f}/step", flush=True)
    s.close()
